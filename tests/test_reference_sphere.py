"""Parity against numbers the REFERENCE ITSELF computed and ships.

/root/reference/examples/system_identification/data/datasets/synthetic_sphere.jld2 holds ten `Storage` trajectories of 100 steps
written by the reference's own `simulate!(mech, 2; record=true)` (examples/system_identification/synthetic_sphere.jl:16-42,
src/simulation/simulate.jl:16-50, src/simulation/storage.jl:50-67) on `get_mechanism(:sphere; timestep=0.02, gravity=-9.81,
friction_coefficient=0.2, radius=0.5)`: one body, one NonlinearContact + SphereHalfSpaceCollision, `mehrotra!` at the default
SolverOptions.  `tools/jld2_reader.py` extracted them into tests/golden/reference_sphere.npz (committed; /root/reference is not
needed at test time).  A Storage row k is (x2, q2, v15, ω15 | px, pq, vl, ωl) of the state the k-th solve started from, so

    step!(row k) == row k+1        for all 10 x 99 pairs,

which pins the converged point of `mehrotra!` -- and, since the solver stops at rtol 1e-6 / btol 1e-4 wherever its iterate path
happens to be, the iterate path itself: two paths that differ end ~1e-6..1e-4 apart, these agree to 1e-13.

CPU tier: the oracle and the device program under the SIMT emulator.  GPU tier: the HIP library through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

import dojo_amd as d
from dojo_amd import api, mechanisms
from oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "reference_sphere.npz")
REF_FILE = "/root/reference/examples/system_identification/data/datasets/synthetic_sphere.jld2"

# the bound of the contract (BASELINE.json north_star) and what is asserted for the fp64 paths: the achieved figures are
# 4e-14 (oracle), 1e-13 (device program), so a change of the iterate path (>= 1e-8 at these tolerances) cannot hide below it
CONTRACT = 1e-6
CONTRACT_F32 = 1e-3
PINNED = 1e-10


def sphere():
    # examples/system_identification/synthetic_sphere.jl:16-20
    return mechanisms.get_sphere(timestep=0.02, gravity=-9.81, friction_coefficient=0.2, radius=0.5)


def rows():
    f = np.load(FIXTURE)
    # maximal state layout per body: x2 v15 q2 ω15 (src/mechanism/state.jl:68-87)
    Z = np.concatenate([f["x"], f["v"], f["q"], f["ω"]], axis=-1)[:, 0]                     # [10, 100, 13]
    S = np.concatenate([f["x"], f["q"], f["v"], f["ω"], f["px"], f["pq"], f["vl"], f["ωl"]], axis=-1)[:, 0]   # [10, 100, 25] Storage row
    return Z, S


def pairs():
    Z, S = rows()
    return Z[:, :-1].reshape(-1, 13), Z[:, 1:].reshape(-1, 13), S[:, :-1].reshape(-1, 25)


def test_fixture_is_what_the_reader_extracts():
    """the committed fixture == a fresh extraction (runs where the reference tree is present: the build container)"""
    if not os.path.exists(REF_FILE):
        pytest.skip("reference tree not present (GPU box): the committed fixture is used as it is")
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    from jld2_reader import read_storages
    data, header = read_storages(REF_FILE)
    assert header.startswith("HDF5-based Julia Data Format")
    f = np.load(FIXTURE)
    assert sorted(f.files) == sorted(data)
    for k in data:
        assert data[k].shape == (10, 1, 100, 4 if k == "q" else 3)
        assert np.array_equal(f[k], data[k]), k
    assert np.abs(np.linalg.norm(data["q"], axis=-1) - 1).max() < 1e-13


def test_fixture_is_a_contact_trajectory():
    """what the rows cover: free flight, impact, sliding and rolling contact -- not only a ball at rest"""
    Z, _ = rows()
    height = Z[:, :, 2] - 0.5
    assert (height > 0.05).any() and (height < 1e-3).sum() > 500
    vt = np.linalg.norm(Z[:, :, 3:5], axis=-1)
    in_contact = height < 1e-3
    assert (vt[in_contact] > 0.1).sum() > 100                                        # sliding / rolling rows
    assert np.all(height > 0)                                                         # interior point: never below the floor


def test_oracle_step_equals_the_reference_rows():
    z0, z1, _ = pairs()
    o = Oracle(sphere())
    zn, st, it, _, _ = o.step_batch(z0)
    e = np.abs(zn - z1).max(axis=1)
    print("oracle vs reference Storage rows: max %.3g over %d pairs; iterations %s" % (e.max(), len(e), np.bincount(it)))
    assert (st == 0).all()
    assert e.max() <= PINNED, (e.max(), int(e.argmax()))


def test_oracle_plain_lu_equals_the_reference_rows():
    """the cpu_baseline leg times the oracle with plain (unrefined) LU solves -- the arithmetic the reference's direct LDU does"""
    z0, z1, _ = pairs()
    o = Oracle(sphere())
    o.set_refine_steps(0)
    zn, st, _, _, _ = o.step_batch(z0)
    assert (st == 0).all() and np.abs(zn - z1).max() <= PINNED


def test_oracle_storage_rows_equal_the_reference():
    """save_to_storage! (storage.jl:50-67): momenta px, pq and the derived velocities vl, ωl of the solved step"""
    Z, S = rows()
    o = Oracle(sphere())
    worst = 0.0
    for i in range(len(Z)):
        got, status = o.simulate_storage(Z[i, 0], np.zeros((100, 6)))
        assert all(s == 0 for s in status)
        worst = max(worst, np.abs(got[:, 0] - S[i]).max())
    print("oracle simulate! Storage vs reference, 100-step rollouts: max %.3g" % worst)
    assert worst <= 1e-9, worst           # 100 chained steps; one-step figure is 4e-14


def test_device_program_equals_the_reference_rows():
    """the shipped device program (dojo_device.hpp) under the CPU SIMT emulator, quad mapping, both ABI types"""
    from emu_wrap import emu_step
    z0, z1, s0 = pairs()
    sel = np.arange(0, len(z0), 3)                      # a third of the pairs keeps the CPU tier short; the GPU tier runs all
    out = emu_step(sphere(), z0[sel], None, dtype="f64", quad=True, envs_per_wave=16)
    e = np.abs(out["z_next"] - z1[sel]).max(axis=1)
    print("device program (emulator) vs reference rows: max %.3g over %d pairs" % (e.max(), len(e)))
    assert (out["status"] == 0).all()
    assert e.max() <= PINNED, e.max()
    assert np.abs(out["storage"][:, 0] - s0[sel]).max() <= PINNED
    out32 = emu_step(sphere(), z0[sel], None, dtype="f32", quad=True, envs_per_wave=16)
    assert np.abs(out32["z_next"] - z1[sel]).max() <= 1e-5      # fp32 buffers at the ABI: input + output rounding (bound: 1e-3)


# ---------------------------------------------------------------------------------------------------------------- GPU tier

@pytest.mark.gpu
@pytest.mark.parametrize("dtype,bound,contract", [("f64", PINNED, CONTRACT), ("f32", 1e-5, CONTRACT_F32)])
def test_hip_step_equals_the_reference_rows(dtype, bound, contract):
    """990 environments = the 990 reference pairs, one launch through dojo_step"""
    z0, z1, _ = pairs()
    gm = api.BatchedMechanism(sphere(), len(z0), dtype=dtype)
    zn, st, it = gm.step(z0.astype(gm.np_dtype))
    gm.close()
    e = np.abs(zn.astype(np.float64) - z1).max(axis=1)
    print("HIP %s vs reference Storage rows: max %.3g over %d pairs; iterations %s" % (dtype, e.max(), len(e), np.bincount(it)))
    assert (st == 0).all()
    assert e.max() <= bound <= contract, e.max()


@pytest.mark.gpu
def test_hip_simulate_storage_equals_the_reference():
    """dojo_simulate (simulate!(...; record=true)) from the ten initial rows: the whole 100-step Storage, momenta included"""
    Z, S = rows()
    gm = api.BatchedMechanism(sphere(), len(Z), dtype="f64")
    Zt, St, st = gm.simulate(Z[:, 0], steps=100)
    gm.close()
    assert (st == 0).all()
    e = np.abs(St[:, :, 0].transpose(1, 0, 2) - S)
    print("HIP dojo_simulate Storage vs reference: max %.3g (x q v ω %.3g | px pq vl ωl %.3g)" % (e.max(), e[..., :13].max(), e[..., 13:].max()))
    assert e.max() <= 1e-9 <= CONTRACT


@pytest.mark.gpu
def test_hip_iterate_path_equals_the_oracles_on_the_reference_rows():
    """equal Newton iteration counts on every pair, and the exported cone variables agree: the path, not only the end point"""
    z0, _, _ = pairs()
    gm = api.BatchedMechanism(sphere(), len(z0), dtype="f64")
    _, st, it = gm.step(z0)
    vel, ji, cs = gm.get_solution()
    gm.close()
    o = Oracle(sphere())
    _, st_o, it_o, _, _ = o.step_batch(z0)
    assert np.array_equal(st, st_o) and np.array_equal(it, it_o)
    for b in range(0, len(z0), 45):
        o.step(z0[b])
        sol = o.get_solution()               # [joint impulses (0); v25 ω25; s γ]
        assert np.abs(sol[:6] - vel[b]).max() <= PINNED
        assert np.abs(sol[6:] - cs[b]).max() <= 1e-9


@pytest.mark.gpu
def test_system_identification_on_the_reference_dataset():
    """The reference's own use of these trajectories (examples/system_identification/synthetic_sphere.jl:45-100, utilities.jl) on the device:
    three-step prediction cost over the ten trajectories as ONE batch, gradient and Gauss-Newton Hessian through dojo_contact_gradients
    (get_contact_gradients, src/gradients/contact.jl:1-55) chained over the steps.  The cost vanishes at the parameters the reference generated
    the data with; the chained analytic gradient is the derivative of the cost (central differences, tight tolerances); and the example's quasi-Newton loop recovers
    friction_coefficient = 0.2, contact_radius = 0.5 from its guess [0, 1]."""
    sys.path.insert(0, os.path.join(HERE, "..", "examples"))
    import sphere_system_identification_device as ex
    Z = ex.dataset()
    f0 = lambda th: ex.loss(np.concatenate([th, np.zeros(3)]), Z)
    fgH0 = lambda th: ex.loss(np.concatenate([th, np.zeros(3)]), Z, derivatives=True)
    assert f0(np.array([0.2, 0.5])) < 1e-20                      # (the device's step IS the reference's on these rows)
    th = np.array([0.12, 0.47]); th5 = np.concatenate([th, np.zeros(3)])
    c, g, H = fgH0(th)
    assert abs(c - f0(th)) < 1e-12 * max(1.0, c)
    # Is the chained gradient the derivative of the cost?  At tight solver tolerances, with the consistent IFT: to 1e-6.  As
    # get_contact_gradients evaluates it after step! (data blocks on the post-update state, the default mode): 1e-4 off.  At the reference's
    # DEFAULT tolerances (what the example runs with) the IFT differentiates the relaxed problem at the central-path parameter the solve ended
    # on (btol = 1e-4), central differences the terminated iteration as a whole: a few percent apart, both good descent directions.
    tight = d.SolverOptions(rtol=1e-10, btol=1e-10)
    h = 1e-5
    fd_t = np.array([(ex.loss(th5 + h * e, Z, opts=tight) - ex.loss(th5 - h * e, Z, opts=tight)) / (2 * h) for e in np.eye(5)[:2]])
    g_cons = ex.loss(th5, Z, derivatives=True, grad_mode=1, opts=tight)[1]
    g_ref = ex.loss(th5, Z, derivatives=True, grad_mode=0, opts=tight)[1]
    fd = np.array([(f0(th + h * e) - f0(th - h * e)) / (2 * h) for e in np.eye(2)])
    print("tight tolerances: consistent IFT %s  reference evaluation %s  central differences %s | default tolerances: %s vs %s" % (g_cons, g_ref, fd_t, g, fd))
    assert np.abs(g_cons - fd_t).max() <= 1e-5 * np.abs(fd_t).max(), (g_cons, fd_t)
    assert np.abs(g_ref - fd_t).max() <= 1e-3 * np.abs(fd_t).max(), (g_ref, fd_t)
    assert np.abs(g - fd).max() <= 0.1 * np.abs(fd).max(), (g, fd)
    sol = ex.quasi_newton_solve(f0, fgH0, np.array([0.0, 1.0]), verbose=False)
    print("recovered friction_coefficient %.6f, contact_radius %.6f, cost %.3e" % (sol[0], sol[1], f0(sol)))
    assert abs(sol[0] - 0.2) < 5e-3 and abs(sol[1] - 0.5) < 5e-4 and f0(sol) < 1e-6


class _OracleBatch:
    """the slice of api.BatchedMechanism the example uses, served by the CPU oracle (one environment at a time)"""

    def __init__(self, spec, batch, dtype="f64", opts=None):
        self.o = Oracle(spec, opts=opts); self.B = batch; self.mode = 0

    def set_gradient_mode(self, mode):
        self.mode = mode

    def step(self, z, with_gradient=False):
        zn = np.zeros_like(z); st = np.zeros(self.B, np.int32); it = np.zeros(self.B, np.int32)
        self.dz = np.zeros((self.B, 12, 12)); self.dc = np.zeros((self.B, 12, 5))
        for b in range(self.B):
            zn[b], info = self.o.step(z[b]); st[b] = info["status"]; it[b] = info["iters"]
            if with_gradient:
                self.dz[b] = self.o.gradients(self.mode)[0]; self.dc[b] = self.o.contact_gradients(self.mode)
        return zn, st, it

    def gradients(self):
        return self.dz, None

    def contact_gradients(self):
        return self.dc

    def close(self):
        pass


def test_system_identification_on_the_reference_dataset_oracle(monkeypatch):
    """the same example through the ORACLE (CPU tier): its get_contact_gradients restatement recovers the reference's parameters from the
    reference's data, and the cost vanishes at them"""
    sys.path.insert(0, os.path.join(HERE, "..", "examples"))
    import sphere_system_identification_device as ex
    monkeypatch.setattr(ex.api, "BatchedMechanism", _OracleBatch)
    Z = ex.dataset()
    f0 = lambda th: ex.loss(np.concatenate([th, np.zeros(3)]), Z)
    fgH0 = lambda th: ex.loss(np.concatenate([th, np.zeros(3)]), Z, derivatives=True)
    assert f0(np.array([0.2, 0.5])) < 1e-20
    sol = ex.quasi_newton_solve(f0, fgH0, np.array([0.0, 1.0]), verbose=False)
    print("oracle: recovered friction_coefficient %.6f, contact_radius %.6f" % (sol[0], sol[1]))
    assert abs(sol[0] - 0.2) < 5e-3 and abs(sol[1] - 0.5) < 5e-4
