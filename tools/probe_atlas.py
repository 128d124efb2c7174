import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
name = sys.argv[1] if len(sys.argv) > 1 else "atlas"
spec = d.get_mechanism(name)
TIGHT = d.SolverOptions(rtol=1e-8, btol=1e-8)
B = 8
Z, U = d.synthetic_inputs(spec, B)
o = Oracle(spec, opts=TIGHT)
z = Z.copy()
for k in range(11):
    z, st_o, it_o, _, _ = o.step_batch(z, U, nthreads=8)
zo, st_o, it_o, _, _ = o.step_batch(z, U, nthreads=8)
for dt in ("f64", "f32"):
    for rw in (None, float("inf")):
        gm = api.BatchedMechanism(spec, B, dtype=dt, opts=TIGHT)
        if rw is not None: gm.set_refinement(rw)
        zz = z.astype(gm.np_dtype)
        zn, st, it = gm.step(zz, U.astype(gm.np_dtype))
        print(dt, "refine", rw, "status", st, "iters", it, "orc", st_o, it_o, "err", np.abs(zn.astype(np.float64) - zo).max(axis=1))
        gm.close()
