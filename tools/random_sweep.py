"""One-off hunt: many random tree mechanisms (tests/random_mechanisms.py) of all sizes, GPU against the oracle and against itself
(determinism).  Prints one line per failure.  usage: random_sweep.py first_seed count"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
from random_mechanisms import random_mechanism
s0, cnt = int(sys.argv[1]), int(sys.argv[2])
TOL = float(os.environ.get("SWEEP_TOL", "1e-9"))
opts = d.SolverOptions(rtol=TOL, btol=TOL)
nfail = 0; nchk = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(1000 + seed)
    TRA = os.environ.get("SWEEP_TRA", "0") == "1"          # joints with free translations, their springs / dampers / limits (<= 16 bodies)
    nb = int(rng.choice([1, 2, 3, 5, 8, 12, 15, 16] if TRA else [1, 2, 3, 5, 8, 12, 15, 16, 17, 20, 28, 31, 32, 33, 36, 48]))
    impact = seed % 5 == 4
    try:
        if os.environ.get("SWEEP_CUT", "0") == "1":          # cut elements (tools/random_cut_sweep.py: a loop-closing joint, a free ball on a body, a contact between
            sys.path.insert(0, os.path.join(ROOT, "tools"))   # two bodies of the tree that are no neighbours); gradients for the loops only (body-body contacts: forward only)
            from random_cut_sweep import with_cut
            got = with_cut(seed)
            if got is None: continue
            kind, spec, z0, u0 = got; nb = spec.Nb; impact = kind != "loop"
        else:
            spec, z0, u0 = random_mechanism(seed, nb=nb, contact_type="impact" if impact else "nonlinear", translational=TRA, tra_limits=TRA)
        B = 3
        Z = np.tile(z0, (B, 1)); U = np.tile(u0, (B, 1)) + rng.normal(size=(B, spec.nu)) * 0.2
        gm = api.BatchedMechanism(spec, B, dtype="f64", opts=opts)
        o = Oracle(spec, opts=opts)
        for k in range(2):
            gm.set_gradient_mode(k % 2)
            zg, st, it = gm.step(Z, U, with_gradient=not impact)
            g = None if impact else gm.gradients()
            zg2, st2, it2 = gm.step(Z, U, with_gradient=not impact)
            if not (np.array_equal(zg, zg2) and np.array_equal(it, it2)):
                print("NONDETERMINISTIC seed %d nb %d" % (seed, nb)); nfail += 1
            Zo = Z.copy()
            for b in range(B):
                zo, info = o.step(Z[b], U[b]); Zo[b] = zo
                if info["status"] != 0 or st[b] != 0:
                    if info["status"] != st[b]:
                        print("status differs seed %d nb %d b %d: gpu %d oracle %d" % (seed, nb, b, st[b], info["status"]))
                    continue
                nchk += 1
                e = np.abs(zg[b] - zo).max()
                msg = []
                if abs(int(it[b]) - info["iters"]) > (0 if TOL < 1e-8 else 2): msg.append("iters %d/%d" % (it[b], info["iters"]))
                if e > 20 * TOL: msg.append("state %.2e" % e)
                if g is not None:
                    dz, du = o.gradients(mode=k % 2)
                    ez = np.abs(g[0][b] - dz).max() / max(1, np.abs(dz).max()); eu = np.abs(g[1][b] - du).max() / max(1, np.abs(du).max()) if spec.nu else 0
                    if ez > 1e3 * TOL or eu > 1e3 * TOL: msg.append("grad %.2e %.2e" % (ez, eu))
                if msg:
                    print("MISMATCH seed %d nb %d contacts %d impact %d step %d env %d: %s" % (seed, nb, len(spec.contacts), impact, k, b, ", ".join(msg))); nfail += 1
            Z = Zo
        gm.close()
    except Exception as ex:
        print("EXCEPTION seed %d nb %d: %r" % (seed, nb, ex)); nfail += 1
print("swept %d mechanisms, %d converged env-steps compared, %d failures" % (cnt, nchk, nfail))
