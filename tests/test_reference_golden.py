"""Reference-held vectors.  tools/reference_golden.jl runs dojo-sim/Dojo.jl itself (step!, get_maximal_gradients!) on the
seeded inputs of tests/golden/reference_inputs/ and writes tests/golden/reference_outputs/config<N>.txt; when those files are present
the oracle (CPU tier) and the HIP path (GPU tier) are compared with them.  The build container has no Julia, so the files are absent
there and these tests skip -- the reader and the exchange format are still exercised (round trip through a file written in the
Julia script's format)."""
import os
import sys
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import reference_exchange as rx          # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "oracle_steps.npz"))
OPTS = d.SolverOptions(rtol=rx.TOL, btol=rx.TOL)
SKIP = "no reference outputs: run `julia --project=<Dojo.jl> tools/reference_golden.jl` on a machine with Julia (tests/golden/reference_outputs/config%d.txt)"


def _write_like_julia(cfg, path, zn, st, dz, du):
    """what tools/reference_golden.jl prints, from arrays in this repository's order (bodies listed in REVERSED order, as a stand-in for
    the reference's own ordering: only names may carry the correspondence)"""
    spec = d.baseline_config(cfg)
    with open(path, "w") as f:
        f.write("config %d x\n" % cfg)
        for c in range(len(zn)):
            f.write("status %d %s\n" % (c, "success" if st[c] == 0 else "failed"))
            for i in reversed(range(spec.Nb)):
                f.write("zn %d %s %s\n" % (c, spec.bodies[i].name, rx.fmt(zn[c, 13 * i:13 * i + 13])))
            for i in reversed(range(spec.Nb)):
                for k in range(spec.Nb):
                    blk = dz[c, 12 * i:12 * i + 12, 12 * k:12 * k + 12]
                    if np.any(blk):
                        f.write("dz %d %s %s %s\n" % (c, spec.bodies[i].name, spec.bodies[k].name, rx.fmt(blk)))
                for j in spec.joints:
                    sl = spec.input_slice(j.name)
                    if sl.stop > sl.start and np.any(du[c, 12 * i:12 * i + 12, sl]):
                        f.write("du %d %s %s %s\n" % (c, spec.bodies[i].name, j.name, rx.fmt(du[c, 12 * i:12 * i + 12, sl])))


@pytest.mark.parametrize("cfg", [1, 3])
def test_exchange_format_round_trip(cfg, tmp_path):
    spec = d.baseline_config(cfg)
    Z, U = G["c%d_z" % cfg], G["c%d_u" % cfg]
    o = Oracle(spec, opts=OPTS)
    Zn, st, it, dz, du = o.step_batch(Z, U, with_grad=True, grad_mode=0, nthreads=4)
    _write_like_julia(cfg, str(tmp_path / ("config%d.txt" % cfg)), Zn, st, dz, du)
    R = rx.load_outputs(cfg, str(tmp_path))
    assert np.array_equal(R["status"], st) and np.array_equal(R["zn"], Zn) and np.array_equal(R["dz"], dz) and np.array_equal(R["du"], du)


def _two_sphere_compare(step):
    """config 6: the two-sphere mechanism with its body-body contact, forward only.  The reference's Newton matrix holds FiniteDiff values
    where the oracle and the device use the analytic expressions (sphere_sphere.jl:57-63): the iterates may differ, the solutions agree to the
    solver tolerance"""
    R = rx.load_outputs(rx.TWO_SPHERES)
    if R is None:
        pytest.skip(SKIP % rx.TWO_SPHERES)
    Z, U = rx.two_spheres_inputs()
    zn, st = step(rx.spec_of(rx.TWO_SPHERES), Z, U)
    ok = (st == 0) & (R["status"] == 0)
    assert ok.sum() >= len(Z) - 1
    assert np.abs(zn[ok] - R["zn"][ok]).max() <= 1e-5


def test_oracle_matches_reference_outputs_two_spheres():
    def step(spec, Z, U):
        Zn, st = Oracle(spec).step_batch(Z, U, nthreads=4)[:2]
        return Zn, st
    _two_sphere_compare(step)


@pytest.mark.gpu
def test_gpu_matches_reference_outputs_two_spheres():
    def step(spec, Z, U):
        from dojo_amd import api
        gm = api.BatchedMechanism(spec, len(Z), dtype="f64")
        zn, st, it = gm.step(Z, U)
        gm.close()
        return zn, st
    _two_sphere_compare(step)


def test_exported_inputs_are_current():
    """tests/golden/reference_inputs/ is what `tools/reference_exchange.py export` writes from the golden inputs"""
    Z6, _ = rx.two_spheres_inputs()
    recs6 = [ln.split() for ln in open(os.path.join(ROOT, "tests", "golden", "reference_inputs", "config%d.txt" % rx.TWO_SPHERES))]
    z6 = [r for r in recs6 if r[0] == "z"]
    assert len(z6) == 2 * len(Z6) and np.array_equal(np.array(z6[3][3:16], dtype=float), Z6[1, 13:26]) and z6[3][2] == "sphere2"
    for cfg in rx.BUILDERS:
        spec = d.baseline_config(cfg)
        Z = G["c%d_z" % cfg]
        recs = [ln.split() for ln in open(os.path.join(ROOT, "tests", "golden", "reference_inputs", "config%d.txt" % cfg))]
        zs = [r for r in recs if r[0] == "z"]
        assert len(zs) == len(Z) * spec.Nb
        r = zs[spec.Nb + 1] if len(zs) > spec.Nb + 1 else zs[0]
        c, i = int(r[1]), spec.body_index(r[2])
        assert np.array_equal(np.array(r[3:16], dtype=float), Z[c, 13 * i:13 * i + 13])


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_oracle_matches_reference_outputs(cfg):
    R = rx.load_outputs(cfg)
    if R is None:
        pytest.skip(SKIP % cfg)
    spec = d.baseline_config(cfg)
    Z, U = G["c%d_z" % cfg], G["c%d_u" % cfg]
    o = Oracle(spec, opts=OPTS)
    Zn, st, it, dz, du = o.step_batch(Z, U, with_grad=True, grad_mode=0, nthreads=4)
    ok = (st == 0) & (R["status"] == 0)
    assert ok.any() and np.array_equal(st == 0, R["status"] == 0)
    assert np.abs(Zn[ok] - R["zn"][ok]).max() <= 1e-6                         # north-star bound, fp64
    for c in np.nonzero(ok)[0]:
        assert np.abs(dz[c] - R["dz"][c]).max() <= 1e-6 * max(1.0, np.abs(R["dz"][c]).max())
        assert np.abs(du[c] - R["du"][c]).max() <= 1e-6 * max(1.0, np.abs(R["du"][c]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_gpu_matches_reference_outputs(cfg):
    R = rx.load_outputs(cfg)
    if R is None:
        pytest.skip(SKIP % cfg)
    from dojo_amd import api
    spec = d.baseline_config(cfg)
    Z, U = G["c%d_z" % cfg], G["c%d_u" % cfg]
    gm = api.BatchedMechanism(spec, len(Z), dtype="f64", opts=OPTS)
    zn, st, it = gm.step(Z, U, with_gradient=True)
    dz, du = gm.gradients()
    gm.close()
    ok = (st == 0) & (R["status"] == 0)
    assert ok.any() and np.array_equal(st == 0, R["status"] == 0)
    assert np.abs(zn[ok] - R["zn"][ok]).max() <= 1e-6
    for c in np.nonzero(ok)[0]:
        assert np.abs(dz[c] - R["dz"][c]).max() <= 1e-6 * max(1.0, np.abs(R["dz"][c]).max())
        assert np.abs(du[c] - R["du"][c]).max() <= 1e-6 * max(1.0, np.abs(R["du"][c]).max())
