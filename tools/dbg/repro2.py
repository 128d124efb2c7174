import sys, os
ROOT = os.getcwd()
for p in ("dojo.jl_amd/host", "oracle", "", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import torch; torch.cuda.init()
import faulthandler; faulthandler.enable()
import test_gpu_parity as t
print("imported", flush=True)
mode = sys.argv[1]
if mode == "test":
    t.test_parity_at_the_baseline_batch_distinct_seeds()
else:
    import numpy as np, dojo_amd as d
    from dojo_amd import api
    spec = d.baseline_config(3); B = 4096
    Z, U = d.synthetic_inputs(spec, B)
    gm = api.BatchedMechanism(spec, B, dtype="f64")
    for _ in range(8):
        Z, st, it = gm.step(Z, U)
    gm.close()
    print("pre-steps done", flush=True)
    gm = api.BatchedMechanism(spec, B, dtype="f64", opts=d.SolverOptions())
    gm.set_gradient_mode(0)
    zn, st, it = gm.step(Z.astype(gm.np_dtype), U.astype(gm.np_dtype), with_gradient=True)
    print("stepped", flush=True)
    dz, du = gm.gradients()
    gm.close()
print("done", flush=True)
