"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of libdojo_hip.so, against
the CPU oracle on the same seeded inputs.  Tolerances: state/solution inf-norm <= 1e-6 in fp64
(tight solver tolerances) and <= 1e-3 in fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle

pytestmark = pytest.mark.gpu
TIGHT = d.SolverOptions(rtol=1e-10, btol=1e-10)


def _rollout_compare(cfg, batch, steps, dtype, tol, opts):
    spec = d.baseline_config(cfg)
    Z, U = d.synthetic_inputs(spec, batch)
    gm = api.BatchedMechanism(spec, batch, dtype=dtype, opts=opts)
    o = Oracle(spec, opts=opts)
    z_o = Z.copy(); z_g = Z.copy()
    worst = 0.0
    for k in range(steps):
        zn_g, st, it = gm.step(z_g, U)
        zn_o, st_o, it_o, _, _ = o.step_batch(z_o, U, nthreads=8)
        ok = (st == 0) & (st_o == 0)
        assert ok.mean() > 0.9
        err = np.abs(zn_g[ok].astype(np.float64) - zn_o[ok]).max()
        worst = max(worst, err)
        z_o = zn_o; z_g = zn_o.astype(gm.np_dtype)       # re-synchronise so errors do not compound over the rollout
    gm.close()
    return worst


@pytest.mark.parametrize("cfg,batch,steps", [(1, 64, 5), (2, 128, 40), (3, 64, 12), (4, 32, 12), (5, 8, 6)])
def test_forward_parity_fp64(cfg, batch, steps):
    worst = _rollout_compare(cfg, batch, steps, "f64", 1e-6, TIGHT)
    assert worst <= 1e-6, worst


@pytest.mark.parametrize("cfg,batch,steps", [(2, 128, 40), (3, 64, 12), (4, 32, 12)])
def test_forward_parity_fp32(cfg, batch, steps):
    worst = _rollout_compare(cfg, batch, steps, "f32", 1e-3, d.SolverOptions())
    assert worst <= 1e-3, worst


def test_solution_export_matches_oracle():
    spec = d.baseline_config(3)
    Z, U = d.synthetic_inputs(spec, 16)
    gm = api.BatchedMechanism(spec, 16, dtype="f64", opts=TIGHT)
    o = Oracle(spec, opts=TIGHT)
    zn, st, it = gm.step(Z, U)
    vel, ji, cs = gm.get_solution()
    for b in range(16):
        o.step(Z[b], U[b])
        sol = o.get_solution()
        nj = spec.n_joint_impulses
        assert np.abs(vel[b] - sol[nj:nj + 6 * spec.Nb]).max() < 1e-6
    gm.close()
