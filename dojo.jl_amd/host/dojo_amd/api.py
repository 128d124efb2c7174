"""ctypes binding of libdojo_hip.so + the host-side mirror of Dojo's API for the hot path.

Reference surface mirrored here (argument meaning and error behaviour follow the reference):
  step!(mechanism, z, u; opts)                 src/simulation/step.jl:11-30        -> step(mech, z, u, opts=...)
  simulate!(mechanism, steps, storage, ctrl!)  src/simulation/simulate.jl:16-36    -> simulate(mech, z0, U)
  get_maximal_gradients!(mechanism, z, u)      src/gradients/state.jl:69-76        -> get_maximal_gradients(mech, z, u)
  get_solution(mechanism)                      src/gradients/finite_difference.jl  -> mech.get_solution()
The batch axis is the leading axis of every array.  There is NO CPU fallback: if the HIP
library or a GPU is missing, construction raises.
"""
import ctypes as C
import os
# one hardware queue per environment group of dojo_rollout (must be set before the HIP runtime starts; ROCm default is 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
from .topology import CTopology, CSolverOptions, CDims, SolverOptions

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "csrc")
_LIB_PATH = os.environ.get("DOJO_HIP_LIB") or os.path.join(_CSRC, "libdojo_hip.so")   # override: instrumented builds (tools/build_variant.sh prof -DDJ_PROF)
_lib = None

STATUS_SUCCESS, STATUS_FAILED, STATUS_EXCESSIVE_W = 0, 1, 2
GRAD_REFERENCE, GRAD_CONSISTENT = 0, 1


class DojoError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise DojoError("libdojo_hip.so not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(_LIB_PATH)
        L.dojo_last_error.restype = C.c_char_p
        L.dojo_handle_error.restype = C.c_char_p; L.dojo_handle_error.argtypes = [C.c_void_p]
        for f in ("dojo_device_count", "dojo_create", "dojo_get_dims", "dojo_set_options", "dojo_set_gradient_mode", "dojo_set_refinement", "dojo_set_async", "dojo_set_groups", "dojo_set_iteration_cap", "dojo_set_dispatch_order", "dojo_join", "dojo_comm_unique_id", "dojo_comm_init", "dojo_allgather_dev", "dojo_comm_info", "dojo_step", "dojo_step_impulses", "dojo_get_mu", "dojo_get_diagnostics", "dojo_next_state", "dojo_next_state_dev",
                  "dojo_get_solution", "dojo_gradients", "dojo_rollout", "dojo_get_state", "dojo_step_dev", "dojo_rollout_dev",
                  "dojo_last_kernel_ms", "dojo_last_kernel_times", "dojo_kernel_time_totals",
                  "dojo_minimal_to_maximal", "dojo_maximal_to_minimal", "dojo_step_minimal",
                  "dojo_minimal_to_maximal_dev", "dojo_maximal_to_minimal_dev", "dojo_step_minimal_dev",
                  "dojo_contact_gradients", "dojo_contact_gradients_dev", "dojo_minimal_gradients", "dojo_minimal_gradients_dev"):
            getattr(L, f).restype = C.c_int
        L.dojo_destroy.restype = None
        _lib = L
    return _lib


EXPORTED_SYMBOLS = ["dojo_device_count", "dojo_last_error", "dojo_handle_error", "dojo_step_impulses", "dojo_get_mu", "dojo_get_diagnostics", "dojo_next_state", "dojo_next_state_dev", "dojo_create", "dojo_destroy", "dojo_get_dims", "dojo_set_options",
                    "dojo_set_gradient_mode", "dojo_set_refinement", "dojo_set_async", "dojo_set_groups", "dojo_set_iteration_cap", "dojo_set_dispatch_order", "dojo_join", "dojo_comm_unique_id", "dojo_comm_init", "dojo_allgather_dev", "dojo_comm_info", "dojo_step", "dojo_get_solution", "dojo_gradients", "dojo_rollout", "dojo_get_state",
                    "dojo_step_dev", "dojo_rollout_dev", "dojo_last_kernel_ms", "dojo_last_kernel_times", "dojo_kernel_time_totals",
                    "dojo_minimal_to_maximal", "dojo_maximal_to_minimal", "dojo_step_minimal",
                    "dojo_minimal_to_maximal_dev", "dojo_maximal_to_minimal_dev", "dojo_step_minimal_dev",
                    "dojo_contact_gradients", "dojo_contact_gradients_dev", "dojo_minimal_gradients", "dojo_minimal_gradients_dev",
                    "dojo_simulate", "dojo_simulate_dev", "dojo_observe", "dojo_observe_dev",
                    "dojo_set_external_force", "dojo_set_external_force_dev"]


# columns of a Storage row (src/simulation/storage.jl:15-24)
STORAGE_FIELDS = {"x": slice(0, 3), "q": slice(3, 7), "v": slice(7, 10), "w": slice(10, 13), "px": slice(13, 16), "pq": slice(16, 19),
                  "vl": slice(19, 22), "wl": slice(22, 25)}


def device_count():
    return lib().dojo_device_count()


def _chk(rc):
    if rc != 0:
        raise DojoError("libdojo_hip error %d: %s" % (rc, lib().dojo_last_error().decode()))


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class BatchedMechanism:
    """B independent copies of one Dojo `Mechanism` resident on one GPU."""

    def __init__(self, spec, batch, dtype="f32", device=0, opts=None):
        self.spec, self.batch = spec, int(batch)
        self.np_dtype = np.float32 if dtype in ("f32", np.float32) else np.float64
        self.dtype_code = 1 if self.np_dtype == np.float32 else 0
        self._topo, self._keep = spec.to_ctypes()
        self.csg_per = 12 if any(c.model == 2 for c in spec.contacts) else 8     # exported [s; γ] scalars per contact (LinearContact: 6 + 6)
        self.h = C.c_void_p()
        _chk(lib().dojo_create(C.byref(self._topo), self.batch, self.dtype_code, int(device), C.byref(self.h)))
        d = CDims()
        _chk(lib().dojo_get_dims(self.h, C.byref(d)))
        self.dims = d
        # (the device exports [s(4); γ(4)] per contact for every contact model: an ImpactContact's entries are [s, 1, 0, 0, γ, 1, 0, 0])
        assert d.nu == spec.nu and d.n_joint_impulses == spec.n_joint_impulses and d.n_solution == spec.n_joint_impulses + 6 * spec.Nb + self.csg_per * len(spec.contacts)
        self.set_options(opts or SolverOptions())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            lib().dojo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, opts):
        self.opts = opts
        o = opts.to_c()
        _chk(lib().dojo_set_options(self.h, C.byref(o)))

    def set_gradient_mode(self, mode):
        _chk(lib().dojo_set_gradient_mode(self.h, int(mode)))

    def set_async(self, on=True):
        """dojo_step_dev no longer joins its environment groups into the caller's stream; join() does it once.  on = 2: pipelined groups as well
        (the IFT kernel of a group's step runs next to the group's next step kernel, include/dojo_hip.h)"""
        _chk(lib().dojo_set_async(self.h, 2 if on == 2 else int(bool(on))))

    def set_groups(self, n):
        _chk(lib().dojo_set_groups(self.h, int(n)))

    def set_iteration_cap(self, cap):
        """dojo_set_iteration_cap: solves unfinished after `cap` Newton iterations go on in the continuation kernel (line-search trials
        side by side) in steps that are joined into the caller's stream; 0 / < 0: off (the default).  Include/dojo_hip.h has the measurements."""
        _chk(lib().dojo_set_iteration_cap(self.h, int(cap)))

    def set_dispatch_order(self, mode):
        """dojo_set_dispatch_order: 0 batch order, 1 (default) the previous step's longest solves first where a joined step has more workgroups
        than the GPU holds at once, 2 always.  Results do not depend on it."""
        _chk(lib().dojo_set_dispatch_order(self.h, int(mode)))

    def join(self, stream=None):
        _chk(lib().dojo_join(self.h, C.c_void_p(stream or 0)))

    # ---- multi-GPU: RCCL communicator of the handle (one process per GPU) ----
    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * 128)()
        _chk(lib().dojo_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        _chk(lib().dojo_comm_init(self.h, int(rank), int(world), C.c_char_p(unique_id)))

    def allgather_dev(self, send_ptr, recv_ptr, count, as_int32=False, stream=None):
        _chk(lib().dojo_allgather_dev(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), C.c_int64(int(count)), int(bool(as_int32)), C.c_void_p(stream or 0)))

    def set_refinement(self, stiffness):
        """Refine the linear solves of environments whose cones reach max gamma/s > stiffness (inf: never, 0: always)."""
        _chk(lib().dojo_set_refinement(self.h, C.c_double(float(stiffness))))

    def _arr(self, a, shape):
        a = np.ascontiguousarray(a, dtype=self.np_dtype)
        if a.shape != shape:
            raise ValueError("expected shape %s, got %s" % (shape, a.shape))
        return a

    def step(self, z, u=None, with_gradient=False):
        B, s = self.batch, self.spec
        z = self._arr(z, (B, s.nz))
        u = None if (u is None or s.nu == 0) else self._arr(u, (B, s.nu))
        zn = np.empty_like(z); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        _chk(lib().dojo_step(self.h, _p(z), _p(u), _p(zn), _p(st), _p(it), int(with_gradient)))
        return zn, st, it

    def step_impulses(self, z, jf):
        """The mehrotra!(mechanism) seam: one step with the controls already folded into the bodies' impulses,
        jf [B, Nb, 6] = [state.JF2 (world); state.Jtau2 (body frame)] (src/integrators/constraint.jl:20-21)."""
        B, s = self.batch, self.spec
        z = self._arr(z, (B, s.nz))
        jf = self._arr(np.asarray(jf).reshape(B, 6 * s.Nb), (B, 6 * s.Nb))
        zn = np.empty_like(z); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        _chk(lib().dojo_step_impulses(self.h, _p(z), _p(jf), _p(zn), _p(st), _p(it)))
        return zn, st, it

    def next_state(self, z):
        """get_next_state of a state whose velocities are its solution: maps dojo_step's return (the internal state after
        update_state!) to the vector the reference's step! literally returns."""
        z = self._arr(z, (self.batch, self.spec.nz)); zo = np.empty_like(z)
        _chk(lib().dojo_next_state(self.h, _p(z), _p(zo)))
        return zo

    def diagnostics(self, read=True):
        """[B, 2]: max gamma/s of the cones and the largest Gauss-Jordan multiplier at the last step's final linearization
        (the first call switches the recording on)"""
        dg = np.zeros((self.batch, 2)) if read else None
        _chk(lib().dojo_get_diagnostics(self.h, _p(dg)))
        return dg

    def get_mu(self):
        mu = np.empty(self.batch, np.float64)
        _chk(lib().dojo_get_mu(self.h, _p(mu)))
        return mu

    def last_error(self):
        return lib().dojo_handle_error(self.h).decode()

    def get_solution(self):
        B, s = self.batch, self.spec
        vel = np.empty((B, 6 * s.Nb), self.np_dtype)
        ji = np.empty((B, max(s.n_joint_impulses, 1)), self.np_dtype)
        cs = np.empty((B, max(self.csg_per * len(s.contacts), 1)), self.np_dtype)
        _chk(lib().dojo_get_solution(self.h, _p(vel), _p(ji), _p(cs)))
        return vel, ji[:, :s.n_joint_impulses], cs[:, :self.csg_per * len(s.contacts)]

    def gradients(self):
        B, s = self.batch, self.spec
        dz = np.empty((B, s.nx, s.nx), self.np_dtype)
        du = np.empty((B, s.nx, max(s.nu, 1)), self.np_dtype)
        _chk(lib().dojo_gradients(self.h, _p(dz), _p(du)))
        return dz, du[:, :, :s.nu]

    def rollout(self, z0, U=None, steps=None, record=True):
        B, s = self.batch, self.spec
        z0 = self._arr(z0, (B, s.nz))
        if U is not None and s.nu:
            U = np.ascontiguousarray(U, dtype=self.np_dtype); H = U.shape[0]
            assert U.shape == (H, B, s.nu)
        else:
            U = None; H = int(steps)
        Z = np.empty((H, B, s.nz), self.np_dtype) if record else None
        st = np.empty((H, B), np.int32)
        _chk(lib().dojo_rollout(self.h, _p(z0), _p(U), H, _p(Z), _p(st)))
        return Z, st

    def set_external_force(self, fext):
        """set_external_force!(body; force, torque) for all bodies: fext [B, Nb, 6] = [Fext (world); τext (body frame)],
        None removes them.  In effect for every following step (bodies/set.jl:110-115, constraint.jl:15-18)."""
        if fext is None:
            _chk(lib().dojo_set_external_force(self.h, None)); return
        f = self._arr(np.asarray(fext).reshape(self.batch, 6 * self.spec.Nb), (self.batch, 6 * self.spec.Nb))
        _chk(lib().dojo_set_external_force(self.h, _p(f)))

    def simulate(self, z0, U=None, steps=None):
        """simulate!(mechanism, 1:H, storage, control!; record=true) with pre-sampled controls (simulate.jl:16-37).
        Returns (Z [H,B,13Nb], storage [H,B,Nb,25], status [H,B]); a Storage row is
        x2(3) q2(4) v15(3) w15(3) px(3) pq(3) vl(3) wl(3) of the solved step (storage.jl:50-67), see STORAGE_FIELDS."""
        B, s = self.batch, self.spec
        z0 = self._arr(z0, (B, s.nz))
        if U is not None and s.nu:
            U = np.ascontiguousarray(U, dtype=self.np_dtype); H = U.shape[0]
            assert U.shape == (H, B, s.nu)
        else:
            U = None; H = int(steps)
        Z = np.empty((H, B, s.nz), self.np_dtype)
        S = np.empty((H, B, s.Nb, 25), self.np_dtype)
        st = np.empty((H, B), np.int32)
        _chk(lib().dojo_simulate(self.h, _p(z0), _p(U), H, _p(Z), _p(S), _p(st)))
        return Z, S, st

    def observe(self, contact_forces=False):
        """get_state(environment): minimal state of the handle's current state [B, 2nu], followed (contact_forces=True,
        get_state(::AntARS), ant_ars.jl:72-80) by the clamped normal impulses of the last step [B, 2nu + Nc]."""
        B, s = self.batch, self.spec
        obs = np.empty((B, 2 * s.nu + (len(s.contacts) if contact_forces else 0)), self.np_dtype)
        _chk(lib().dojo_observe(self.h, _p(obs), int(bool(contact_forces))))
        return obs

    def contact_gradients(self):
        """get_contact_gradients (src/gradients/contact.jl) at the solution of the last step(..., with_gradient=True):
        [B, 12Nb, 5Nc], contact data = [friction_coefficient, contact_radius, contact_origin(3)] per contact."""
        B, s = self.batch, self.spec
        dc = np.zeros((B, s.nx, 5 * len(s.contacts)), self.np_dtype)
        _chk(lib().dojo_contact_gradients(self.h, _p(dc)))
        return dc

    # ---- minimal <-> maximal coordinates (src/mechanism/state.jl:9-66, src/simulation/step.jl:42-60) ----
    def minimal_to_maximal(self, x):
        B, s = self.batch, self.spec
        x = self._arr(x, (B, 2 * s.nu))
        z = np.empty((B, s.nz), self.np_dtype)
        _chk(lib().dojo_minimal_to_maximal(self.h, _p(x), _p(z)))
        return z

    def maximal_to_minimal(self, z):
        B, s = self.batch, self.spec
        z = self._arr(z, (B, s.nz))
        x = np.empty((B, 2 * s.nu), self.np_dtype)
        _chk(lib().dojo_maximal_to_minimal(self.h, _p(z), _p(x)))
        return x

    def step_minimal(self, x, u=None):
        """step_minimal_coordinates!: x [B, 2 nu] -> x_next, status, iters"""
        B, s = self.batch, self.spec
        x = self._arr(x, (B, 2 * s.nu))
        u = self._arr(u, (B, s.nu)) if (u is not None and s.nu) else None
        xn = np.empty((B, 2 * s.nu), self.np_dtype)
        st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        _chk(lib().dojo_step_minimal(self.h, _p(x), _p(u), _p(xn), _p(st), _p(it)))
        return xn, st, it

    def minimal_gradients(self, x, u=None):
        """get_minimal_gradients!: one step in minimal coordinates -> (x_next, status, iters, jx [B,2nu,2nu], ju [B,2nu,nu])"""
        B, s = self.batch, self.spec
        nm = 2 * s.nu
        x = self._arr(x, (B, nm))
        u = self._arr(u, (B, s.nu)) if (u is not None and s.nu) else None
        xn = np.empty((B, nm), self.np_dtype); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        jx = np.empty((B, nm, nm), self.np_dtype); ju = np.empty((B, nm, max(s.nu, 1)), self.np_dtype)
        _chk(lib().dojo_minimal_gradients(self.h, _p(x), _p(u), _p(xn), _p(st), _p(it), _p(jx), _p(ju)))
        return xn, st, it, jx, ju.reshape(-1)[:B * nm * s.nu].reshape(B, nm, s.nu)

    def last_kernel_ms(self):
        ms = C.c_double(0)
        _chk(lib().dojo_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def kernel_time_totals(self, reset=False):
        """(sum of step-kernel ms, sum of IFT-kernel ms, launches) over the timed launches since the last reset."""
        a = C.c_double(0); b = C.c_double(0); n = C.c_int64(0)
        _chk(lib().dojo_kernel_time_totals(self.h, C.byref(a), C.byref(b), C.byref(n), 1 if reset else 0))
        return a.value, b.value, n.value

    def last_kernel_times(self):
        """(step kernel ms, IFT kernel ms) of the last launch, from hipEvents on the launch stream."""
        a = C.c_double(0); b = C.c_double(0)
        _chk(lib().dojo_last_kernel_times(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


# ---------------------------------------------------------------------------------------
# Dojo-style free functions
# ---------------------------------------------------------------------------------------
def step(mechanism, z, u=None, opts=None):
    """step!(mechanism, z, u; opts): returns the mechanism's state after the step
    ((x3, v25, q3, ω25) per body -- SURVEY.md §8a Q1), plus per-environment status."""
    if opts is not None:
        mechanism.set_options(opts)
    zn, status, iters = mechanism.step(z, u, with_gradient=False)
    return zn, status


def get_maximal_gradients(mechanism, z, u=None, opts=None):
    """get_maximal_gradients!(mechanism, z, u; opts) -> (jacobian_state [B,12Nb,12Nb], jacobian_control [B,12Nb,nu])"""
    if opts is not None:
        mechanism.set_options(opts)
    mechanism.step(z, u, with_gradient=True)
    return mechanism.gradients()


def simulate(mechanism, z0, U=None, steps=None, opts=None, abort_upon_failure=False):
    """simulate!(mechanism, steps, storage, control!) with pre-sampled controls U[k]."""
    if opts is not None:
        mechanism.set_options(opts)
    return mechanism.rollout(z0, U, steps=steps, record=True)
