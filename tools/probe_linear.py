"""GPU probe: forward env-steps/s of LinearContact / ImpactContact / NonlinearContact blocks (4 corner contacts) at batch 1024."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import dojo_amd as d
from dojo_amd import api
B, H = 1024, 200
rng = np.random.default_rng(0)
for ct in ("nonlinear", "linear", "impact"):
    spec = d.get_block(contact_type=ct, contact_corners=4)
    Z = np.stack([d.initialize(spec, position=[0, 0, rng.uniform(0.0, 0.3)], velocity=rng.normal(size=3), angular_velocity=rng.normal(size=3) * 0.5) for _ in range(B)])
    gm = api.BatchedMechanism(spec, B, dtype="f64")
    gm.rollout(Z, None, steps=20, record=False)
    t0 = time.time(); Zt, st = gm.rollout(Z, None, steps=H, record=False); el = time.time() - t0
    print("%-9s B %d: %.2f M env-steps/s (forward, fp64, rollout of %d steps incl. host copies), converged %.4f" % (ct, B, B * H / el / 1e6, H, float((st == 0).mean())))
    gm.close()
