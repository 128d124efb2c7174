"""The iteration cap + continuation kernel (Globals::iter_cap, dojo_set_iteration_cap) must not change a single bit.

A solve that is unfinished after `cap` Newton iterations leaves the step kernel with its loop scalars (KernelArgs::resume) and goes on in
the continuation kernel, where R wavefronts carry the same environment and evaluate the line-search trials alpha / 2^k side by side before
replaying line_search!'s accept / halve decisions (src/solver/line_search.jl:1-34) over the results.  Every evaluation is the same function
of the same values as in the sequential search, so iterates, iteration counts, status and gradients are those of the uncapped loop exactly.

CPU tier: the device program under the SIMT emulator (EMU_ITER_CAP=<cap>:<replicas>) against itself without the cap, on
tests/golden/long_solves_ant.npz (Ant solves of 21..50 iterations, seven of them running into max_iter with exhausted line searches --
tools/long_solves.py) and on the other mapping cases (several contacts per body split over the quad, several environments per wavefront,
joint limits).  The GPU tier compares capped and uncapped steps through the C ABI (tests/test_iteration_cap_gpu.py).
"""
import os
import numpy as np
import pytest

import dojo_amd as d
from emu_wrap import emu_step

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("z_next", "status", "iters", "vel", "joint_imp", "contact_sg", "storage")


def _run(spec, Z, U, cap, grad=False, **kw):
    if cap is None:
        os.environ.pop("EMU_ITER_CAP", None)
    else:
        os.environ["EMU_ITER_CAP"] = cap
    try:
        return emu_step(spec, Z, U, quad=True, grad=grad, **kw)
    finally:
        os.environ.pop("EMU_ITER_CAP", None)


def _same(r0, r1, grad=False):
    for k in KEYS + (("dz", "du") if grad else ()):
        assert np.array_equal(r0[k], r1[k]), (k, np.abs(np.asarray(r0[k], float) - np.asarray(r1[k], float)).max())


def long_solves():
    f = np.load(os.path.join(HERE, "golden", "long_solves_ant.npz"))
    return f["z"], f["u"], f["iters"], f["status"]


@pytest.mark.parametrize("cap", ["16:4", "5:3", "2:2"])
def test_long_ant_solves_are_unchanged_by_the_cap(cap):
    """three solves that run into max_iter (status 1, exhausted line searches all the way) and three long converged ones"""
    Z, U, it, st = long_solves()
    sel = [0, 3, 6, 7, 12, 20]
    spec = d.baseline_config(3)
    r0 = _run(spec, Z[sel], U[sel], None)
    assert np.array_equal(r0["iters"], it[sel]) and np.array_equal(r0["status"], st[sel])          # (the oracle's counts, tools/long_solves.py)
    _same(r0, _run(spec, Z[sel], U[sel], cap))


def test_gradients_after_a_continued_solve_are_unchanged():
    Z, U, it, st = long_solves()
    sel = [9, 15, 23]                                     # 44, 31 and 21 iterations, converged
    spec = d.baseline_config(3)
    r0 = _run(spec, Z[sel], U[sel], None, grad=True)
    assert np.all(r0["status"] == 0)
    _same(r0, _run(spec, Z[sel], U[sel], "7:4", grad=True), grad=True)


@pytest.mark.parametrize("cfg,cap,epw", [(2, "3:3", 4), (4, "4:4", 1)])
def test_other_mappings_are_unchanged_by_the_cap(cfg, cap, epw):
    """Block-on-plane (four contacts split over the quad, four environments per wavefront: a capped wavefront carries finished and
    unfinished environments, three replicas), Quadruped"""
    spec = d.baseline_config(cfg)
    Z, U = d.synthetic_inputs(spec, 2 * epw if epw > 1 else 2)
    r0 = _run(spec, Z, U, None, envs_per_wave=epw)
    assert r0["iters"].max() > int(cap.split(":")[0])
    _same(r0, _run(spec, Z, U, cap, envs_per_wave=epw))


def test_sixteen_environments_per_wavefront():
    """the :sphere mechanism (one body, one contact: four lanes per environment, sixteen environments per wavefront) dropped, rolling and
    resting: the environments of a wavefront need 5 .. 50 iterations, so a capped wavefront carries finished and unfinished ones and the
    continuation's line-search verdicts are exchanged per environment"""
    spec = d.get_sphere()
    rng = np.random.default_rng(3)
    B = 32
    Z = np.zeros((B, 13)); Z[:, 6] = 1.0
    Z[:, 2] = 0.5 + np.where(np.arange(B) % 3 == 0, 0.3, 1e-3) * rng.random(B)        # a third of them in the air
    Z[:, 3:6] = rng.normal(0, 1.0, (B, 3)); Z[:, 10:13] = rng.normal(0, 2.0, (B, 3))
    r0 = _run(spec, Z, None, None, envs_per_wave=16)
    assert r0["iters"].min() <= 6 and r0["iters"].max() >= 40, r0["iters"]           # (5 .. 50: fast spins make this contact hard)
    for cap in ("7:4", "12:2"):
        _same(r0, _run(spec, Z, None, cap, envs_per_wave=16))


def test_joint_limits_and_a_cap_of_one():
    """cartpole with limits on the cart (a limit row in the line search's cone step and centering) and the smallest cap there is"""
    spec = d.get_cartpole(joint_limits={"cart_joint": [-0.3, 0.3]}, dampers=0.1)
    from dojo_amd import coords
    Z = np.stack([coords.initialize(spec) for _ in range(4)])
    Z[:, 1] = [0.29, -0.295, 0.2999, 0.1]               # cart positions next to its limits
    Z[:, 4] = [2.0, -3.0, 5.0, 0.0]                     # ... moving into them
    U = np.zeros((4, spec.nu)); U[:, 0] = [5.0, -5.0, 20.0, 0.0]
    r0 = _run(spec, Z, U, None, envs_per_wave=4)
    _same(r0, _run(spec, Z, U, "1:4", envs_per_wave=4))
