"""GPU probe: roll a mechanism out like test_translational_springs_dampers_gpu and dump the environment-steps whose
iteration count / state differs from the oracle's -> gpurun_out/mismatch_<name>.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
name, batch, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spec = d.get_mechanism(name)
TIGHT = d.SolverOptions(rtol=1e-8, btol=1e-8)
Z, U = d.synthetic_inputs(spec, batch)
gm = api.BatchedMechanism(spec, batch, dtype="f64", opts=TIGHT)
o = Oracle(spec, opts=TIGHT)
z = Z.copy(); bad = []
for k in range(steps):
    zg, st, it = gm.step(z, U)
    zo, st_o, it_o, _, _ = o.step_batch(z, U, nthreads=16)
    ok = (st == 0) & (st_o == 0)
    e = np.abs(zg - zo).max(axis=1)
    for b in np.nonzero(ok & ((it != it_o) | (e > 1e-7)))[0]:
        print("step", k, "env", b, "it", it[b], it_o[b], "err %.2e" % e[b]); bad.append((z[b].copy(), U[b].copy(), k, b))
    z = zo
if bad:
    np.savez(os.path.join(ROOT, "gpurun_out", "mismatch_%s.npz" % name), z=np.array([x[0] for x in bad]), u=np.array([x[1] for x in bad]), meta=np.array([[x[2], x[3]] for x in bad]))
print("n bad", len(bad))
