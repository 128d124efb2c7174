"""Default tolerances, every environment through the refining kernels (dojo_set_refinement(h, 0)) at the BASELINE batch:
gradient error against the oracle over ALL converged environments, and what the refining kernels cost."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = d.baseline_config(3)
Z, U = d.synthetic_inputs(spec, B)
g = api.BatchedMechanism(spec, B, dtype="f64")
for _ in range(8):
    Z, st, it = g.step(Z, U)
o = Oracle(spec)
Zo, st_o, it_o, dz_o, du_o = o.step_batch(Z, U, with_grad=True, nthreads=os.cpu_count())
for thr in (float("inf"), 1e8, 1e6, 1e4, 0.0):
    g.set_refinement(thr)
    t0 = time.time(); zn, st, it = g.step(Z, U, with_gradient=True); dz, du = g.gradients(); t1 = time.time()
    ok = np.nonzero((st == 0) & (st_o == 0))[0]
    eg = np.array([max(np.abs(dz[b] - dz_o[b]).max() / max(1.0, np.abs(dz_o[b]).max()), np.abs(du[b] - du_o[b]).max() / max(1.0, np.abs(du_o[b]).max())) for b in ok])
    es = np.abs(zn[ok] - Zo[ok]).max()
    print("threshold %-8g converged %d iters_mismatch %d state %.2e grad max %.2e q99 %.2e above1e-6 %d  kernels %s" % (
        thr, len(ok), int((it[ok] != it_o[ok]).sum()), es, eg.max(), np.quantile(eg, 0.99), int((eg > 1e-6).sum()), g.last_kernel_times()), flush=True)
g.close()
