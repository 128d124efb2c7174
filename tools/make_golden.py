#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_steps.npz: seeded inputs and the oracle's outputs for the five BASELINE configurations.

The reference itself cannot be run here (no Julia), so these are NOT reference outputs: they freeze the
oracle (which is pinned by the reference's own finite-difference property tests, tests/test_oracle_*.py) so that a change
of the oracle or of the host-side mechanism builders shows up as a diff, and they give the GPU tests one fixed set of
vectors that does not depend on the oracle being rebuilt the same way.  Usage: python tools/make_golden.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import dojo_amd as d
from oracle import Oracle

out = {}
BASE = "--mechanisms-only" not in sys.argv        # python tools/make_golden.py --mechanisms-only leaves oracle_steps.npz as it is
for cfg, B, pre in (((1, 4, 2), (2, 4, 60), (3, 4, 12), (4, 2, 8), (5, 2, 2)) if BASE else ()):
    spec = d.baseline_config(cfg)
    opts = d.SolverOptions(rtol=1e-8, btol=1e-8)
    o = Oracle(spec, opts=opts)
    Z, U = d.synthetic_inputs(spec, B)
    for _ in range(pre):
        Z, st, it, _, _ = o.step_batch(Z, U, nthreads=4)
    Zn, st, it, dz, du = o.step_batch(Z, U, with_grad=True, grad_mode=0, nthreads=4)
    out["c%d_z" % cfg] = Z; out["c%d_u" % cfg] = U; out["c%d_zn" % cfg] = Zn; out["c%d_status" % cfg] = st; out["c%d_iters" % cfg] = it
    # Jacobians: keep them small -- the first environment, fp32 is plenty for a 1e-6 relative comparison
    out["c%d_dz0" % cfg] = dz[0].astype(np.float64) if cfg <= 2 else dz[0].astype(np.float32)
    out["c%d_du0" % cfg] = du[0].astype(np.float64) if cfg <= 2 else du[0].astype(np.float32)
if BASE:
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_steps.npz"), **out)
    print("wrote tests/golden/oracle_steps.npz", {k: v.shape for k, v in out.items() if k.endswith("_z")})


# Mechanisms beyond the five BASELINE configurations (translational springs / dampers, other joint prototypes):
# tests/golden/oracle_steps_mechanisms.npz, same conventions.
MECHS = {"raiberthopper": (dict(), 4, 3), "nslider": (dict(num_bodies=4, springs=1.0, dampers=0.2), 4, 2),
         "snake_planaraxis": (dict(num_bodies=3, joint_type="PlanarAxis", springs=1.0, dampers=0.3), 4, 4),
         "twister": (dict(num_bodies=4, springs=0.5, dampers=0.2), 4, 4),
         "npendulum_orbital": (dict(num_bodies=3, rest_joint_type="Orbital", springs=0.5, dampers=0.3), 4, 3)}
out = {}
for key, (kw, B, pre) in MECHS.items():
    spec = d.get_mechanism(key.split("_")[0], **kw)
    o = Oracle(spec, opts=d.SolverOptions(rtol=1e-8, btol=1e-8))
    Z, U = d.synthetic_inputs(spec, B)
    for _ in range(pre):
        Z, st, it, _, _ = o.step_batch(Z, U, nthreads=4)
    Zn, st, it, dz, du = o.step_batch(Z, U, with_grad=True, grad_mode=0, nthreads=4)
    out[key + "_z"] = Z; out[key + "_u"] = U; out[key + "_zn"] = Zn; out[key + "_status"] = st; out[key + "_iters"] = it
    out[key + "_dz0"] = dz[0]; out[key + "_du0"] = du[0]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_steps_mechanisms.npz"), **out)
print("wrote tests/golden/oracle_steps_mechanisms.npz", {k: v.shape for k, v in out.items() if k.endswith("_z")}, {k: v for k, v in out.items() if k.endswith("_status")})


# Contact models that are forward only in the reference (no data Jacobians): LinearContact, ImpactContact.
# tests/golden/oracle_steps_contacts.npz: thrown blocks / spheres a few steps after touching the floor.
CONTACTS = {"block_linear": ("block", dict(contact_type="linear", contact_corners=4, friction_coefficient=0.3)),
            "sphere_linear": ("sphere", dict(contact_type="linear")),
            "block_impact": ("block", dict(contact_type="impact", contact_corners=4))}
out = {}
rng = np.random.default_rng(11)
for key, (name, kw) in CONTACTS.items():
    spec = d.get_mechanism(name, **kw)
    o = Oracle(spec, opts=d.SolverOptions(rtol=1e-8, btol=1e-8))
    B = 4
    if name == "block":
        Z = np.stack([d.initialize(spec, position=[0, 0, rng.uniform(0.0, 0.05)], velocity=rng.normal(size=3), angular_velocity=rng.normal(size=3) * 0.5) for _ in range(B)])
    else:
        Z = np.tile(d.initialize(spec), (B, 1)); Z[:, 2] = 0.5 + rng.uniform(0.0, 0.05, B); Z[:, 3:6] = rng.normal(size=(B, 3)); Z[:, 10:13] = rng.normal(size=(B, 3))
    U = np.zeros((B, spec.nu))
    for _ in range(14):
        Z, st, it, _, _ = o.step_batch(Z, U, nthreads=4)
    Zn, st, it, _, _ = o.step_batch(Z, U, nthreads=4)
    sg = []
    for b in range(B):
        o.step(Z[b], U[b]); sg.append(o.get_solution()[6 * spec.Nb + spec.n_joint_impulses:])
    out[key + "_z"] = Z; out[key + "_zn"] = Zn; out[key + "_status"] = st; out[key + "_iters"] = it; out[key + "_sg"] = np.stack(sg)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_steps_contacts.npz"), **out)
print("wrote tests/golden/oracle_steps_contacts.npz", {k: v.shape for k, v in out.items() if k.endswith("_sg")}, {k: v for k, v in out.items() if k.endswith("_status") or k.endswith("_iters")})
