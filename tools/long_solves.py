"""tests/golden/long_solves_ant.npz: Ant environment-steps whose Mehrotra solve is long (> 20 Newton iterations, many with exhausted line
searches, some running into max_iter) -- the inputs the iteration cap / continuation kernel is for.  Found with the
CPU oracle in a closed-loop rollout with random controls (seeded); expected values come from the oracle / the uncapped device program at
test time.  Usage: python tools/long_solves.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host"), os.path.join(ROOT, "oracle"), ROOT]
import dojo_amd as d          # noqa: E402
from oracle import Oracle     # noqa: E402

spec = d.baseline_config(3)
B, H = 1024, 16
Z, U0 = d.synthetic_inputs(spec, B)
rng = np.random.default_rng(20241008)
o = Oracle(spec)
zs, us, its, sts = [], [], [], []
for k in range(H):
    U = rng.normal(0, 0.5, U0.shape)
    Zn, st, it, _, _ = o.step_batch(Z, U, nthreads=8)
    idx = np.nonzero((it > 20) | (st != 0))[0]
    for i in idx:
        zs.append(Z[i].copy()); us.append(U[i].copy()); its.append(it[i]); sts.append(st[i])
    print(k, len(idx), it.mean(), file=sys.stderr)
    Z = Zn
its = np.array(its); sts = np.array(sts)
order = np.argsort(-its)
# keep a spread: every solve that failed, and the longest converged ones, 48 in all
keep = list(order[:48])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "long_solves_ant.npz"), z=np.array(zs)[keep], u=np.array(us)[keep], iters=its[keep], status=sts[keep])
print(len(zs), "found;", "kept", len(keep), "iters", its[keep], "status", sts[keep])
