"""Golden fixtures (tests/golden/oracle_steps.npz, made by tools/make_golden.py): frozen oracle outputs on seeded inputs.
They are oracle outputs, not reference outputs -- the reference needs Julia.  CPU tier: the oracle still
reproduces them; GPU tier: the HIP path, through the C ABI, matches them."""
import os
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "oracle_steps.npz"))
OPTS = d.SolverOptions(rtol=1e-8, btol=1e-8)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_oracle_reproduces_golden(cfg):
    spec = d.baseline_config(cfg)
    Z, U = G["c%d_z" % cfg], G["c%d_u" % cfg]
    o = Oracle(spec, opts=OPTS)
    Zn, st, it, dz, du = o.step_batch(Z, U, with_grad=True, grad_mode=0, nthreads=4)
    assert np.array_equal(st, G["c%d_status" % cfg]) and np.array_equal(it, G["c%d_iters" % cfg])
    assert np.abs(Zn - G["c%d_zn" % cfg]).max() < 1e-12
    ref = G["c%d_dz0" % cfg].astype(np.float64)
    assert np.abs(dz[0] - ref).max() <= 2e-7 * max(1.0, np.abs(ref).max())      # the larger fixtures are stored in fp32


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_gpu_matches_golden(cfg):
    from dojo_amd import api
    spec = d.baseline_config(cfg)
    Z, U = G["c%d_z" % cfg], G["c%d_u" % cfg]
    gm = api.BatchedMechanism(spec, len(Z), dtype="f64", opts=OPTS)
    zn, st, it = gm.step(Z, U, with_gradient=True)
    dz, du = gm.gradients()
    gm.close()
    ok = (st == 0) & (G["c%d_status" % cfg] == 0)
    assert ok.any()
    assert np.array_equal(it[ok], G["c%d_iters" % cfg][ok])                # the same Newton iterate path ...
    assert np.abs(zn[ok] - G["c%d_zn" % cfg][ok]).max() < 1e-6            # ... hence the north-star bound as a maximum
    if ok[0]:
        ref = G["c%d_dz0" % cfg].astype(np.float64)
        assert np.abs(dz[0] - ref).max() <= 1.2e-6 * max(1.0, np.abs(ref).max())      # (the larger fixtures are stored in fp32: 1e-6 + their rounding)


# ---- mechanisms beyond the five BASELINE configurations (translational springs / dampers, other joint prototypes) ----
GM = np.load(os.path.join(ROOT, "tests", "golden", "oracle_steps_mechanisms.npz"))
MECHS = {"raiberthopper": dict(), "nslider": dict(num_bodies=4, springs=1.0, dampers=0.2),
         "snake_planaraxis": dict(num_bodies=3, joint_type="PlanarAxis", springs=1.0, dampers=0.3),
         "twister": dict(num_bodies=4, springs=0.5, dampers=0.2),
         "npendulum_orbital": dict(num_bodies=3, rest_joint_type="Orbital", springs=0.5, dampers=0.3)}


@pytest.mark.parametrize("key", sorted(MECHS))
def test_oracle_reproduces_golden_mechanisms(key):
    spec = d.get_mechanism(key.split("_")[0], **MECHS[key])
    o = Oracle(spec, opts=OPTS)
    Zn, st, it, dz, du = o.step_batch(GM[key + "_z"], GM[key + "_u"], with_grad=True, grad_mode=0, nthreads=4)
    assert np.array_equal(st, GM[key + "_status"]) and np.array_equal(it, GM[key + "_iters"])
    assert np.abs(Zn - GM[key + "_zn"]).max() < 1e-12
    assert np.abs(dz[0] - GM[key + "_dz0"]).max() <= 1e-9 * max(1.0, np.abs(GM[key + "_dz0"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(MECHS))
def test_gpu_matches_golden_mechanisms(key):
    from dojo_amd import api
    spec = d.get_mechanism(key.split("_")[0], **MECHS[key])
    Z, U = GM[key + "_z"], GM[key + "_u"]
    gm = api.BatchedMechanism(spec, len(Z), dtype="f64", opts=OPTS)
    zn, st, it = gm.step(Z, U, with_gradient=True)
    dz, du = gm.gradients()
    gm.close()
    ok = (st == 0) & (GM[key + "_status"] == 0)
    assert ok.any()
    assert np.array_equal(it[ok], GM[key + "_iters"][ok])
    assert np.abs(zn[ok] - GM[key + "_zn"][ok]).max() < 1e-6
    if ok[0]:
        ref = GM[key + "_dz0"]
        assert np.abs(dz[0] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
        if spec.nu:
            assert np.abs(du[0] - GM[key + "_du0"]).max() <= 1e-6 * max(1.0, np.abs(GM[key + "_du0"]).max())


# ---- contact models that are forward only in the reference (LinearContact, ImpactContact) ----
GC = np.load(os.path.join(ROOT, "tests", "golden", "oracle_steps_contacts.npz"))
CONTACTS = {"block_linear": ("block", dict(contact_type="linear", contact_corners=4, friction_coefficient=0.3)),
            "sphere_linear": ("sphere", dict(contact_type="linear")),
            "block_impact": ("block", dict(contact_type="impact", contact_corners=4))}


@pytest.mark.parametrize("key", sorted(CONTACTS))
def test_oracle_reproduces_golden_contacts(key):
    name, kw = CONTACTS[key]
    spec = d.get_mechanism(name, **kw)
    o = Oracle(spec, opts=OPTS)
    Z = GC[key + "_z"]; U = np.zeros((len(Z), spec.nu))
    Zn, st, it, _, _ = o.step_batch(Z, U, nthreads=4)
    assert np.array_equal(st, GC[key + "_status"]) and np.array_equal(it, GC[key + "_iters"])
    assert np.abs(Zn - GC[key + "_zn"]).max() < 1e-12
    o.step(Z[0], U[0])
    assert np.abs(o.get_solution()[6 * spec.Nb + spec.n_joint_impulses:] - GC[key + "_sg"][0]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(CONTACTS))
def test_gpu_matches_golden_contacts(key):
    from dojo_amd import api
    name, kw = CONTACTS[key]
    spec = d.get_mechanism(name, **kw)
    Z = GC[key + "_z"]; U = np.zeros((len(Z), spec.nu))
    gm = api.BatchedMechanism(spec, len(Z), dtype="f64", opts=OPTS)
    zn, st, it = gm.step(Z, U)
    _, _, sg = gm.get_solution()
    gm.close()
    assert np.array_equal(st, GC[key + "_status"]) and np.array_equal(it, GC[key + "_iters"])
    assert np.abs(zn - GC[key + "_zn"]).max() < 1e-8
    ref = GC[key + "_sg"]
    if key == "block_impact":                                   # the device exports [s(4); γ(4)] per contact for an ImpactContact: entries 0 and 4
        sg = sg.reshape(len(Z), -1, 8)[:, :, [0, 4]].reshape(len(Z), -1)
    assert np.abs(sg - ref).max() < 1e-7 * max(1.0, np.abs(ref).max())
