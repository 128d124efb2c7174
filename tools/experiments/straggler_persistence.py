"""How predictable are the long solves?  Closed-loop Ant rollout as in bench.py (fresh random controls every step): for every step, which of the
environments that take > 20 Newton iterations also did so in the previous step (a hardness-sorted dispatch can only start early what it can predict)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import dojo_amd as d
from dojo_amd import api
spec = d.baseline_config(3); B = 4096
Z, U0 = d.synthetic_inputs(spec, B)
rng = np.random.Generator(np.random.Philox(key=[20241008, 1000]))
gm = api.BatchedMechanism(spec, B, dtype="f32")
z = Z.astype(np.float32); prev = None
for k in range(26):
    U = (0.5 * rng.standard_normal((B, spec.nu)) * (np.abs(U0) > 0)).astype(np.float32)
    z, st, it = gm.step(z, U)
    hard = it > 20
    if prev is not None:
        order = np.argsort(-prev_it, kind="stable")
        rank = np.empty(B, int); rank[order] = np.arange(B)
        print("step %2d: > 20 iterations %3d (failed %2d) | of those, > 20 in the previous step %3d, > 14 %3d | in the first 256 of the previous step's order: %3d, first 1024: %3d"
              % (k, hard.sum(), (st != 0).sum(), (hard & prev).sum(), (hard & (prev_it > 14)).sum(), (rank[hard] < 256).sum(), (rank[hard] < 1024).sum()))
    prev = hard; prev_it = it.copy()
gm.close()
