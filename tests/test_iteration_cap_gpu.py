"""GPU tier: the iteration cap + continuation kernel (dojo_set_iteration_cap, include/dojo_hip.h) changes the results of a step by rounding only.

The step kernel hands solves that are unfinished after `cap` Newton iterations to dojo_stepc_kernel, whose wavefront replicas evaluate the
line-search trials of src/solver/line_search.jl:1-34 side by side (dojo_device.hpp, mehrotra()).  The algorithm is the uncapped one decision
for decision -- under the SIMT emulator, where both halves are the same compiled code, every output is equal BIT FOR BIT
(tests/test_iteration_cap_emu.py).  On the GPU the continuation kernel is a second instantiation of the lane program and the compiler
contracts a*b+c into fused multiply-adds differently in it, so its iterates differ from the step kernel's in the last bits (1e-14 relative
per evaluation) -- the relation the step kernel has to the CPU oracle.  Asserted here, through the C ABI, with the cap off
(dojo_set_iteration_cap(h, 0)) against a cap of 16 and caps that push almost every solve through the continuation kernel:
equal status and iteration counts (long solves excepted, counted), states / solutions / Jacobians to 1e-6 where the counts agree (the contract's bound for two fp64 implementations: at the default tolerances a solve of ~20 iterations ends at mu ~ 1e-9 where the conditioning of the KKT system turns last-bit differences into 1e-7; measured: 5e-12 typical, 2.9e-7 worst of 65 536 environment-steps); and the
iteration counts of the long solves against the CPU oracle's (tests/golden/long_solves_ant.npz, tools/long_solves.py).
"""
import os
import numpy as np
import pytest

import dojo_amd as d
from dojo_amd import api

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _step_all(spec, Z, U, cap, dtype="f64", grad=True, opts=None):
    gm = api.BatchedMechanism(spec, len(Z), dtype=dtype, opts=opts)
    gm.set_iteration_cap(cap)
    zn, st, it = gm.step(Z, U, with_gradient=grad)
    out = dict(zn=zn, st=st, it=it)
    out["vel"], out["ji"], out["cs"] = gm.get_solution()
    out["mu"] = gm.get_mu()
    if grad:
        out["dz"], out["du"] = gm.gradients()
    gm.close()
    return out


LONG = 20                 # iterations beyond which a solve may end apart from its twin (tests/test_gpu_parity.py::_split_by_state)
APART_CEILING = 1e-2      # ... and never further than this


def _close(a, b, what, tol=1e-9, max_path=0):
    """Equal status / iteration counts but for `max_path` environments (long solves whose iterate paths part after a rounding difference).
    Where they agree: every output within tol (relative to max(1, |.|) per environment) for solves of at most LONG iterations; a longer one
    wanders at mu ~ 1e-12 on a near-singular system and may end apart (both ends within the solver's tolerances) -- its state within
    APART_CEILING, counted."""
    same = (a["st"] == b["st"]) & (a["it"] == b["it"])
    npath = int((~same).sum())
    short = same & (a["it"] <= LONG); lng = same & (a["it"] > LONG)
    worst = 0.0
    worst_j = 0.0
    for k in a:
        if k in ("st", "it") or not short.any():
            continue
        x = np.asarray(a[k], float)[short].reshape(int(short.sum()), -1); y = np.asarray(b[k], float)[short].reshape(int(short.sum()), -1)
        if x.size == 0:
            continue
        e_ = float((np.abs(x - y).max(axis=1) / np.maximum(1.0, np.abs(x).max(axis=1))).max())
        if k in ("dz", "du"): worst_j = max(worst_j, e_)         # Jacobians: the conditioning of the converged system amplifies the states' last bits
        else: worst = max(worst, e_)
    assert worst_j <= max(1e-6, 10 * tol), (what, "Jacobians", worst_j)   # (the north-star's gradient bound)
    el = np.abs(np.asarray(a["zn"], float)[lng] - np.asarray(b["zn"], float)[lng]).max(axis=1) if lng.any() else np.zeros(0)
    print("%s: %d of %d on another iterate path (iters %s vs %s); %d short solves, worst difference %.2e; %d long solves, %d of them within %g, worst %.2e"
          % (what, npath, len(same), a["it"][~same][:8], b["it"][~same][:8], int(short.sum()), worst, int(lng.sum()), int((el <= tol).sum()), tol, el.max() if len(el) else 0.0))
    assert npath <= max_path, (what, npath, a["it"][~same], b["it"][~same])
    assert worst <= tol, (what, worst)
    assert len(el) == 0 or el.max() <= APART_CEILING, (what, el.max())
    return npath, worst


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_long_ant_solves_bit_identical_and_counts_are_the_oracles(dtype):
    f = np.load(os.path.join(HERE, "golden", "long_solves_ant.npz"))
    Z, U = f["z"], f["u"]
    spec = d.baseline_config(3)
    ref = _step_all(spec, Z, U, 0, dtype)
    # the oracle's counts on these inputs (7 of the 24 run into max_iter = 50): solves of 20..50 iterations at mu ~ 1e-12 amplify rounding
    # differences, so a few take another path on any two implementations -- counted
    assert (ref["it"] != f["iters"]).sum() <= (4 if dtype == "f64" else 8), (ref["it"], f["iters"])
    tol = 1e-6 if dtype == "f64" else 1e-5              # (fp32 ABI: the outputs are rounded to 2^-24)
    for cap in (16, 1, 5, 30):
        _close(ref, _step_all(spec, Z, U, cap, dtype), "long solves %s cap %d" % (dtype, cap), tol, max_path=6)
        assert (_step_all(spec, Z, U, cap, dtype)["it"] != f["iters"]).sum() <= (6 if dtype == "f64" else 10)


def test_full_batch_closed_loop_rollout_bit_identical():
    """BASELINE configs[2] at its batch: 4096 distinct Ant environments, 12 closed-loop steps with fresh random controls, the rollout
    entry point (environment groups on internal streams, every group with its own continuation list)"""
    spec = d.baseline_config(3)
    B, H = 4096, 12
    Z0, U0 = d.synthetic_inputs(spec, B)
    rng = np.random.default_rng(7)
    U = (0.5 * rng.standard_normal((H, B, spec.nu)) * (np.abs(U0) > 0)).astype(np.float32)
    outs = []
    for cap in (0, 16, 6):
        gm = api.BatchedMechanism(spec, B, dtype="f32")
        gm.set_iteration_cap(cap)
        Z, st = gm.rollout(Z0.astype(np.float32), U)
        outs.append((Z, st)); gm.close()
    for Z, st in outs[1:]:                                  # (the rollout entry point chains its groups' steps without a barrier: no cap in force there)
        assert np.array_equal(outs[0][0], Z) and np.array_equal(outs[0][1], st)
    assert set(np.unique(outs[0][1])) <= {0, 1, 2}          # the internal DJ_STATUS_CONTINUE never reaches the caller


def test_full_batch_joined_steps():
    """BASELINE configs[2] at its batch, stepped with a join per step (dojo_step: where the cap is in force): 4096 distinct Ant environments,
    8 closed-loop steps from the same states with the cap off / at 16 / at 6 -- every step compared from the SAME input states"""
    spec = d.baseline_config(3)
    B, H = 4096, 8
    Z, U0 = d.synthetic_inputs(spec, B)
    rng = np.random.default_rng(7)
    g0 = api.BatchedMechanism(spec, B, dtype="f64"); g0.set_iteration_cap(0)
    g1 = api.BatchedMechanism(spec, B, dtype="f64"); g1.set_iteration_cap(16)
    g2 = api.BatchedMechanism(spec, B, dtype="f64"); g2.set_iteration_cap(6)
    npath = 0; nlong = 0
    for k in range(H):
        U = 0.5 * rng.standard_normal((B, spec.nu)) * (np.abs(U0) > 0)
        outs = []
        for gm in (g0, g1, g2):
            zn, st, it = gm.step(Z, U, with_gradient=True)
            dz, du = gm.gradients()
            outs.append(dict(zn=zn, st=st, it=it, dz=dz, du=du))
        assert set(np.unique(outs[0]["st"])) <= {0, 1, 2}
        nlong += int((outs[0]["it"] > 16).sum())
        for o, nm in ((outs[1], "cap 16"), (outs[2], "cap 6")):
            n_, w_ = _close(outs[0], o, "Ant B 4096 step %d %s" % (k, nm), 1e-6, max_path=4)
            npath += n_
        Z = outs[0]["zn"]
    for gm in (g0, g1, g2):
        gm.close()
    assert nlong > 10                                       # the batch does contain solves a cap of 16 hands over
    assert npath <= 8


@pytest.mark.parametrize("cfg,B", [(2, 1024), (4, 512), ("sphere", 256)])
def test_other_configurations(cfg, B):
    """Block-on-plane (contacts split over the quad, four environments per wavefront, three replicas), Quadruped, and the :sphere mechanism
    spinning on the ground (sixteen environments per wavefront with 5 .. 50 iterations: a listed workgroup carries finished and unfinished ones)"""
    if cfg == "sphere":
        spec = d.get_sphere()
        rng = np.random.default_rng(3)
        Z = np.zeros((B, 13)); Z[:, 6] = 1.0
        Z[:, 2] = 0.5 + np.where(np.arange(B) % 3 == 0, 0.3, 1e-3) * rng.random(B)
        Z[:, 3:6] = rng.normal(0, 1.0, (B, 3)); Z[:, 10:13] = rng.normal(0, 2.0, (B, 3))
        U = None
    else:
        spec = d.baseline_config(cfg)
        Z, U = d.synthetic_inputs(spec, B)
    ref = _step_all(spec, Z, U, 0, "f64", True)
    assert ref["it"].max() >= 5, ref["it"].max()
    for cap in (int(ref["it"].max()) // 2, 2):            # half of the solves / nearly all of them go through the continuation kernel
        _close(ref, _step_all(spec, Z, U, cap, "f64", True), "%s cap %d" % (cfg, cap), 1e-6, max_path=max(2, B // 128))


def test_forward_only_and_tight_tolerances():
    """forward-only steps take the same path; with tolerances that switch the refining kernels on the cap is out of force (no continuation
    kernel exists for them) and the call still gives the uncapped result"""
    spec = d.baseline_config(3)
    Z, U = d.synthetic_inputs(spec, 256)
    _close(_step_all(spec, Z, U, 0, "f64", False), _step_all(spec, Z, U, 3, "f64", False), "forward only", 1e-6)
    tight = d.SolverOptions(rtol=1e-8, btol=1e-8)
    a, b = _step_all(spec, Z, U, 0, "f64", True, tight), _step_all(spec, Z, U, 3, "f64", True, tight)
    for k in a:
        assert np.array_equal(a[k], b[k]), k                # (no cap in force: the same kernels)
