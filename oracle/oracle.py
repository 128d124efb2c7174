"""ctypes wrapper of the CPU oracle (oracle/liboracle.so; liboracle_fast.so for timing) -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
shipped product (dojo.jl_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "..", "dojo.jl_amd", "host"))
from dojo_amd.topology import CTopology, CSolverOptions, SolverOptions  # noqa: E402

_libs = {}


def build(force=False, fast=False):
    """liboracle.so: the checker (host-independent build, oracle/Makefile).  fast=True: liboracle_fast.so, the same source compiled
    -O3 -march=native -- for bench.py's cpu_baseline leg only (timed, never compared with)."""
    name = "liboracle_fast.so" if fast else "liboracle.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("capi.cpp", "dojo_oracle.hpp", "oracle_math.hpp", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        if all(os.path.exists(s) for s in srcs):
            subprocess.check_call(["make", "-C", _HERE, name] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return so


def lib(fast=False):
    if fast not in _libs:
        _lib = C.CDLL(build(fast=fast))
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_create.argtypes = [C.POINTER(CTopology), C.c_int]
        for name in ("orc_destroy", "orc_set_options", "orc_dims", "orc_get_solution", "orc_set_solution", "orc_gradients",
                     "orc_get_data", "orc_set_data", "orc_evaluate_residual", "orc_full_matrix", "orc_data_matrix",
                     "orc_data_attjac", "orc_set_state", "orc_get_state", "orc_set_external_force",
                     "orc_body_velocity_solution", "orc_save_to_storage", "orc_energy_of_storage_row", "orc_step_batch", "orc_debug_assemble", "orc_check_solution"):
            getattr(_lib, name).restype = None
        _lib.orc_time_batch.restype = C.c_double
        _lib.orc_physical_cores.restype = C.c_int
        _lib.orc_set_refine_steps.restype = None
        _lib.orc_set_sparse_solver.restype = None; _lib.orc_sparse_flops.restype = C.c_longlong; _lib.orc_sparse_solve_flops.restype = C.c_longlong
        _lib.orc_op_count.restype = C.c_longlong; _lib.orc_op_count.argtypes = [C.c_int]
        _lib.orc_ls_stats.restype = None
        _lib.orc_unit.restype = C.c_int; _lib.orc_joint_unit.restype = C.c_int; _lib.orc_contact_unit.restype = C.c_int
        _lib.orc_input_impulses.restype = None
        _lib.orc_maximal_to_minimal.restype = None; _lib.orc_minimal_to_maximal.restype = None
        _lib.orc_step.restype = C.c_int
        _lib.orc_simulate_step.restype = C.c_int
        _lib.orc_simulate_step_record.restype = C.c_int
        _libs[fast] = _lib
    return _libs[fast]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Single-environment CPU oracle of a MechanismSpec."""

    def __init__(self, spec, dtype="f64", opts=None, fast=False):
        """fast=True: the -march=native build (bench.py's cpu_baseline leg: timed, never the checker)"""
        self.spec = spec
        self._L = lib(fast)
        self._topo, self._keep = spec.to_ctypes()
        # dtype "count": the operation-counting scalar (oracle/counted.hpp; op_count() reads the thread's counter)
        self.h = C.c_void_p(self._L.orc_create(C.byref(self._topo), 0 if dtype == "f64" else 99 if dtype == "count" else 1))
        d = (C.c_int * 7)()
        self._L.orc_dims(self.h, d)
        self.n, self.nu, self.nd_full, self.nd, self.Nb, self.Ne, self.Nc = list(d)
        assert self.n == spec.n_solution and self.nu == spec.nu
        self.set_options(opts or SolverOptions())

    def __del__(self):
        try:
            self._L.orc_destroy(self.h)
        except Exception:
            pass

    def set_options(self, opts):
        self.opts = opts
        o = opts.to_c()
        self._L.orc_set_options(self.h, C.byref(o))

    # step!(mechanism, z, u)
    def step(self, z, u=None):
        z = np.ascontiguousarray(z, dtype=np.float64)
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        zs = np.zeros(13 * self.Nb); zr = np.zeros(13 * self.Nb); it = C.c_int(0)
        st = self._L.orc_step(self.h, _p(z), _p(u), _p(zs), _p(zr), C.byref(it))
        return zs, dict(status=st, iters=it.value, z_return=zr)

    def get_solution(self):
        s = np.zeros(self.n); self._L.orc_get_solution(self.h, _p(s)); return s

    def set_solution(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64); self._L.orc_set_solution(self.h, _p(s))

    def gradients(self, mode=0):
        nx = 12 * self.Nb
        dz = np.zeros((nx, nx)); du = np.zeros((nx, max(self.nu, 1)))
        self._L.orc_gradients(self.h, mode, _p(dz), _p(du))
        return dz, du.reshape(-1)[:nx * self.nu].reshape(nx, self.nu)

    def contact_gradients(self, mode=0):
        """get_contact_gradients (src/gradients/contact.jl): [12Nb, 5Nc], theta per contact = [friction, radius, origin(3)]"""
        nx = 12 * self.Nb
        dc = np.zeros((nx, max(5 * self.Nc, 1)))
        self._L.orc_contact_gradients(self.h, mode, _p(dc))
        return dc.reshape(-1)[:nx * 5 * self.Nc].reshape(nx, 5 * self.Nc)

    def get_data(self):
        d = np.zeros(self.nd_full); self._L.orc_get_data(self.h, _p(d)); return d

    def set_data(self, d):
        d = np.ascontiguousarray(d, dtype=np.float64); self._L.orc_set_data(self.h, _p(d))

    def evaluate_residual(self, data, sol):
        out = np.zeros(self.n)
        self._L.orc_evaluate_residual(self.h, _p(np.ascontiguousarray(data, dtype=np.float64)), _p(np.ascontiguousarray(sol, dtype=np.float64)), _p(out))
        return out

    def full_matrix(self):
        A = np.zeros((self.n, self.n)); self._L.orc_full_matrix(self.h, _p(A)); return A

    def data_matrix(self):
        D = np.zeros((self.n, self.nd)); self._L.orc_data_matrix(self.h, _p(D)); return D

    def data_attjac(self):
        G = np.zeros((self.nd_full, self.nd)); self._L.orc_data_attjac(self.h, _p(G)); return G

    def debug_assemble(self, z, u=None):
        A = np.zeros((self.n, self.n)); b = np.zeros(self.n)
        z = np.ascontiguousarray(z, dtype=np.float64); u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        self._L.orc_debug_assemble(self.h, _p(z), _p(u), _p(A), _p(b)); return A, b

    def check_solution(self, z, u, sol):
        """(rvio, bvio) of a candidate solution [joint impulses; body velocities; contact s,γ] of step!(z, u)"""
        v = np.zeros(2)
        z = np.ascontiguousarray(z, dtype=np.float64); u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        self._L.orc_check_solution(self.h, _p(z), _p(u), _p(np.ascontiguousarray(sol, dtype=np.float64)), _p(v)); return v[0], v[1]

    # simulate! pieces
    def set_state(self, z):
        z = np.ascontiguousarray(z, dtype=np.float64); self._L.orc_set_state(self.h, _p(z))

    def get_state(self):
        z = np.zeros(13 * self.Nb); self._L.orc_get_state(self.h, _p(z)); return z

    def set_external_force(self, body, force=(0, 0, 0), torque=(0, 0, 0), vertex=(0, 0, 0)):
        f, t, v = (np.array(a, dtype=np.float64) for a in (force, torque, vertex))
        self._L.orc_set_external_force(self.h, int(body), _p(f), _p(t), _p(v))

    def simulate_step(self, u=None, last=False):
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        return self._L.orc_simulate_step(self.h, _p(u), int(last))

    def velocity_solution(self):
        v = np.zeros(6 * self.Nb); self._L.orc_body_velocity_solution(self.h, _p(v)); return v

    def storage_row(self):
        """save_to_storage!(mechanism, storage, k) (storage.jl:50-67) for the body states as they are now:
        [Nb, 25] = x2(3) q2(4) v15(3) w15(3) px(3) pq(3) vl(3) wl(3)."""
        out = np.zeros((self.Nb, 25)); self._L.orc_save_to_storage(self.h, _p(out)); return out

    def energy(self, rows):
        """kinetic_energy, potential_energy (src/mechanics/energy.jl:24-92) of Storage rows [H, Nb, 25] (or one row [Nb, 25]) -> (ke [H], pe [H])"""
        rows = np.ascontiguousarray(rows, dtype=np.float64); one = rows.ndim == 2
        R = rows[None] if one else rows
        out = np.zeros((len(R), 2))
        for k in range(len(R)):
            self._L.orc_energy_of_storage_row(self.h, _p(np.ascontiguousarray(R[k])), _p(out[k]))
        return (out[0, 0], out[0, 1]) if one else (out[:, 0], out[:, 1])

    def simulate(self, z0, steps, control=None):
        """simulate!(mechanism, steps, storage, control!)  -> list of maximal states after each step.
        The last step is not followed by update_state! (simulate.jl:32), as in the reference."""
        self.set_state(z0)
        traj = []; status = []
        for k in range(1, steps + 1):
            u = control(self, k) if control else None
            status.append(self.simulate_step(u, last=(k == steps)))
            traj.append(self.get_state())
        return traj, status

    def simulate_storage(self, z0, U):
        """simulate!(mechanism, 1:H, storage, control!; record=true) with pre-sampled controls U [H, nu]:
        the solve, then save_to_storage! BEFORE update_state! (simulate.jl:28-32).  Returns ([H, Nb, 25], status)."""
        self.set_state(z0)
        H = len(U); rows = []; status = []
        for k in range(H):
            status.append(self._L.orc_simulate_step_record(self.h, _p(np.ascontiguousarray(U[k], dtype=np.float64)), int(k == H - 1), _p(row := np.zeros((self.Nb, 25)))))
            rows.append(row)
        return np.stack(rows), status

    def input_impulses(self, z, u):
        """set_maximal_state! + set_input!: the body impulses [JF2 (world); Jtau2 (body)] per body that mehrotra! finds, [Nb, 6]"""
        z = np.ascontiguousarray(z, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        jf = np.zeros(6 * self.Nb)
        self._L.orc_input_impulses(self.h, _p(z), _p(u), _p(jf))
        return jf.reshape(self.Nb, 6)

    def maximal_to_minimal(self, z):
        """maximal_to_minimal(mechanism, z)  src/mechanism/state.jl:44-66"""
        z = np.ascontiguousarray(z, dtype=np.float64); x = np.zeros(2 * self.nu)
        self._L.orc_maximal_to_minimal(self.h, _p(z), _p(x)); return x

    def minimal_to_maximal(self, x):
        """minimal_to_maximal(mechanism, x)  src/mechanism/state.jl:9-22"""
        x = np.ascontiguousarray(x, dtype=np.float64); z = np.zeros(13 * self.Nb)
        self._L.orc_minimal_to_maximal(self.h, _p(x), _p(z)); return z

    def set_refine_steps(self, n):
        """rounds of iterative refinement of every linear solve: 2 (default) = the checker, 0 = plain LU like the reference's direct solve"""
        self._L.orc_set_refine_steps(self.h, int(n))

    def set_sparse_solver(self, on=True):
        """timing variant (bench.py cpu_baseline): sparse LU without pivoting in the elimination order of the mechanism graph"""
        self._L.orc_set_sparse_solver(self.h, int(bool(on)))

    def op_count(self, reset=True):
        """floating-point operations (+ - * / sqrt sin cos atan pow, one each: a multiply-add = 2) the counting instances (dtype="count") of the calling
        thread have executed OUTSIDE their linear solves since the last reset"""
        return int(self._L.orc_op_count(1 if reset else 0))

    def ls_stats(self):
        """(line searches, residual evaluations) of this instance since creation (single-environment calls only)"""
        out = (C.c_longlong * 2)()
        self._L.orc_ls_stats(self.h, out)
        return int(out[0]), int(out[1])

    def joint_unit(self, joint, half, what, xa, qa, xb, qb, p=None, vel=None):
        """displacement (what 0) / displacement_jacobian_configuration(:parent 1 | :child 2; attjac) / impulse_transform * p (3 | 4) /
        impulse_transform_jacobian (5..8: pp, pc, cp, cc) / damper impulses (9 | 10) and their configuration (11..14) and velocity
        (15..18) Jacobians at the velocities vel = [va ωa vb ωb], of one joint half (0 translational, 1 rotational); impulse_map * λ (19 | 20)
        and impulse_map_jacobian (21..24), λ = p (impulses_length entries); see oracle/capi.cpp"""
        if what >= 19:
            p = np.concatenate([np.asarray(p, float), np.zeros(15 - len(p))]); vel = np.zeros(0)
        inp = np.concatenate([xa, qa, xb, qb, np.zeros(3) if p is None else p, np.zeros(12) if vel is None else vel]).astype(np.float64); out = np.zeros(36)
        n = self._L.orc_joint_unit(self.h, int(joint), int(half), int(what), _p(inp), _p(out))
        return out[:n].copy()

    def contact_unit(self, contact, what, xp, qp, xc, qc):
        """collision functions of a body-body contact at given configurations (see oracle/capi.cpp: contact_unit)"""
        inp = np.concatenate([xp, qp, xc, qc]).astype(np.float64); out = np.zeros(16)
        n = self._L.orc_contact_unit(self.h, int(contact), int(what), _p(inp), _p(out))
        if n < 0:
            raise ValueError("contact_unit: contact %d / what %d" % (contact, what))
        return out[:n].copy()

    def sparse_flops(self):
        return int(self._L.orc_sparse_flops(self.h))

    def sparse_solve_flops(self):
        """flops of one forward + backward substitution with the block-sparse factors (per right-hand side)"""
        return int(self._L.orc_sparse_solve_flops(self.h))

    def time_batch(self, Z, U=None, with_grad=False, grad_mode=0, nthreads=1, rounds=1):
        """wall-clock seconds for `rounds` passes over the batch on `nthreads` persistent threads (results discarded)"""
        Z = np.ascontiguousarray(Z, dtype=np.float64); U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        return float(self._L.orc_time_batch(self.h, Z.shape[0], _p(Z), _p(U), int(with_grad), int(grad_mode), int(nthreads), int(rounds)))

    def step_batch(self, Z, U=None, with_grad=False, grad_mode=0, nthreads=1):
        Z = np.ascontiguousarray(Z, dtype=np.float64); B = Z.shape[0]
        U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        Zn = np.zeros_like(Z); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
        nx = 12 * self.Nb
        dz = np.zeros((B, nx, nx)) if with_grad else None
        du = np.zeros((B, nx, self.nu)) if with_grad else None
        self._L.orc_step_batch(self.h, B, _p(Z), _p(U), _p(Zn), _p(st), _p(it), int(with_grad), grad_mode, _p(dz), _p(du), nthreads)
        return Zn, st, it, dz, du


def physical_cores():
    """physical cores this process may run on (one hardware thread per core; orc_time_batch pins one thread to each)"""
    return int(lib().orc_physical_cores())


def unit(what, q, w=(0.0, 0.0, 0.0), dt=0.01):
    """unit functions of the restatement (oracle/capi.cpp: orc_unit): flat fp64 result"""
    inp = np.array(list(q) + list(w) + [dt], dtype=np.float64)
    out = np.zeros(16)
    n = lib().orc_unit(int(what), _p(inp), _p(out))
    assert n > 0
    return out[:n].copy()
