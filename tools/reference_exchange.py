#!/usr/bin/env python3
"""Exchange files between this repository and a machine that can run the reference (Julia + Dojo.jl), SURVEY.md §8c.

  python tools/reference_exchange.py export     writes tests/golden/reference_inputs/config<N>.txt: the seeded inputs of
                                                tests/golden/oracle_steps.npz keyed by BODY and JOINT NAME (the reference orders its
                                                bodies by Julia Dict iteration: only names are portable)
  julia --project=<Dojo.jl checkout> tools/reference_golden.jl     (there)  -> tests/golden/reference_outputs/config<N>.txt
  python -m pytest tests/test_reference_golden.py                           compares the oracle (CPU tier) and the HIP path (GPU tier)
                                                                            with those files when they exist, skips otherwise

Plain text, one record per line, so that the Julia side needs nothing beyond its standard library:
  config <N> <builder> <kwargs as key=value ...>
  options rtol btol
  case <c>
  z <c> <body name> x(3) v15(3) q(4: s v1 v2 v3) w15(3)
  u <c> <joint name> <input values of that joint ...>
Outputs (written by the Julia script, parsed by load_outputs below):
  status <c> <success|failed>
  zn <c> <body name> 13 values                      the mechanism's state after step! (x2 v15 q2 w15 after update_state!)
  dz <c> <row body> <col body> 144 values           12 x 12 block of jacobian_state, row-major; rows (x3 v25 phi3 w25), cols (x2 v15 phi2 w15)
  du <c> <row body> <joint name> 12 * nu_j values   12 x nu_j block of jacobian_control, row-major"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import dojo_amd as d

# the reference-side constructor calls that build the same five mechanisms (DojoEnvironments/src/mechanisms/*/mechanism.jl)
BUILDERS = {1: ("pendulum", ""), 2: ("block", "contact_corners=4"), 3: ("ant", "contact_body=false"), 4: ("quadruped", "contact_body=false"), 5: ("atlas", "contact_body=false")}
TOL = 1e-8
# config 6 (forward only): the two-sphere mechanism of test/collisions.jl:2-58 with a body-body (SphereSphereCollision) contact, built inline
# by tools/reference_golden.jl; reference-default tolerances (its Newton matrix is inexact for such a contact: DESIGN.md section 9)
TWO_SPHERES = 6
TWO_SPHERES_KW = dict(friction_type="nonlinear", joint_world_body1="Floating", gravity=-9.81)


def spec_of(cfg):
    return d.get_two_spheres(**TWO_SPHERES_KW) if cfg == TWO_SPHERES else d.baseline_config(cfg)


def two_spheres_inputs():
    """seeded states of the two-sphere mechanism: the second sphere approaching / touching / resting on the first from random directions"""
    rng = np.random.default_rng(606)
    C = 12
    Z = np.zeros((C, 2, 13)); Z[:, :, 6] = 1.0
    dirs = rng.normal(size=(C, 3)); dirs[:, 2] = np.abs(dirs[:, 2]) + 0.3; dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    Z[:, 1, 0:3] = dirs * rng.uniform(1.0005, 1.3, size=(C, 1))
    Z[:, 1, 3:6] = -dirs * rng.uniform(0.0, 3.0, size=(C, 1)) + 0.3 * rng.normal(size=(C, 3))
    Z[:, 1, 10:13] = rng.normal(size=(C, 3)); Z[:, 0, 3:6] = 0.2 * rng.normal(size=(C, 3))
    return Z.reshape(C, 26), np.zeros((C, 12))


def fmt(v):
    return " ".join(repr(float(x)) for x in np.asarray(v).ravel())


def export():
    G = np.load(os.path.join(ROOT, "tests", "golden", "oracle_steps.npz"))
    out = os.path.join(ROOT, "tests", "golden", "reference_inputs"); os.makedirs(out, exist_ok=True)
    for cfg, (name, kw) in BUILDERS.items():
        spec = d.baseline_config(cfg)
        Z, U = G["c%d_z" % cfg], G["c%d_u" % cfg]
        with open(os.path.join(out, "config%d.txt" % cfg), "w") as f:
            f.write("config %d %s %s\n" % (cfg, name, kw)); f.write("options %r %r\n" % (TOL, TOL))
            for c in range(len(Z)):
                f.write("case %d\n" % c)
                for i, b in enumerate(spec.bodies):
                    f.write("z %d %s %s\n" % (c, b.name, fmt(Z[c, 13 * i:13 * i + 13])))
                for j in spec.joints:
                    sl = spec.input_slice(j.name)
                    if sl.stop > sl.start:
                        f.write("u %d %s %s\n" % (c, j.name, fmt(U[c, sl])))
        print("wrote", os.path.join(out, "config%d.txt" % cfg))
    spec = spec_of(TWO_SPHERES); Z, _ = two_spheres_inputs()
    with open(os.path.join(out, "config%d.txt" % TWO_SPHERES), "w") as f:
        f.write("config %d two_spheres %s\n" % (TWO_SPHERES, " ".join("%s=%s" % kv for kv in TWO_SPHERES_KW.items())))
        f.write("options %r %r\n" % (1e-6, 1e-4))
        for c in range(len(Z)):
            f.write("case %d\n" % c)
            for i, b in enumerate(spec.bodies):
                f.write("z %d %s %s\n" % (c, b.name, fmt(Z[c, 13 * i:13 * i + 13])))
    print("wrote", os.path.join(out, "config%d.txt" % TWO_SPHERES))


def load_outputs(cfg, directory=None):
    """-> dict(status [C], zn [C, 13 Nb], dz [C, 12 Nb, 12 Nb], du [C, 12 Nb, nu]) in THIS repository's body / joint order, or None"""
    path = os.path.join(directory or os.path.join(ROOT, "tests", "golden", "reference_outputs"), "config%d.txt" % cfg)
    if not os.path.exists(path):
        return None
    spec = spec_of(cfg)
    bi = {b.name: i for i, b in enumerate(spec.bodies)}
    recs = [ln.split() for ln in open(path) if ln.strip()]
    C = 1 + max(int(r[1]) for r in recs if r[0] in ("zn", "status"))
    nb, nu = spec.Nb, spec.nu
    out = dict(status=np.ones(C, int), zn=np.full((C, 13 * nb), np.nan), dz=np.zeros((C, 12 * nb, 12 * nb)), du=np.zeros((C, 12 * nb, nu)))
    for r in recs:
        c = int(r[1]) if r[0] != "config" else 0
        if r[0] == "status": out["status"][c] = 0 if r[2] == "success" else 1
        elif r[0] == "zn": out["zn"][c, 13 * bi[r[2]]:13 * bi[r[2]] + 13] = [float(x) for x in r[3:16]]
        elif r[0] == "dz": out["dz"][c, 12 * bi[r[2]]:12 * bi[r[2]] + 12, 12 * bi[r[3]]:12 * bi[r[3]] + 12] = np.array(r[4:148], dtype=float).reshape(12, 12)
        elif r[0] == "du":
            sl = spec.input_slice(r[3]); n = sl.stop - sl.start
            out["du"][c, 12 * bi[r[2]]:12 * bi[r[2]] + 12, sl] = np.array(r[4:4 + 12 * n], dtype=float).reshape(12, n)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "export": export()
    else: print(__doc__)
