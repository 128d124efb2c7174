"""Which environments carry the largest gradient errors of the plain fp64-ABI path, against their final cone stiffness (max gamma/s) -- the
selection criterion of a gradient-only refinement.  GPU + oracle; usage: python tools/probe_grad_refine.py [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host"), os.path.join(ROOT, "oracle"), ROOT]
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = d.baseline_config(3)
Z, U = d.synthetic_inputs(spec, B)
o = Oracle(spec)
gm = api.BatchedMechanism(spec, B, dtype="f64")
gm.set_refinement(1e30)          # tracked, never exceeded
gm.diagnostics(read=False)       # (the first call switches the recording on)
for _ in range(8):
    Z, st, it = gm.step(Z, U)
zn, st, it = gm.step(Z, U, with_gradient=True)
dz, du = gm.gradients()
diag = gm.diagnostics()
w = np.asarray(diag)[:, 0]
Zo, st_o, it_o, dz_o, du_o = o.step_batch(Z, U, with_grad=True, nthreads=os.cpu_count())
ok = (st == 0) & (st_o == 0)
ea = np.array([max(np.abs(dz[b] - dz_o[b]).max(), np.abs(du[b] - du_o[b]).max()) if ok[b] else 0.0 for b in range(B)])
print("stiffness quantiles", np.quantile(w[ok], [0.1, 0.5, 0.9, 0.99, 1.0]))
print("abs grad err quantiles", np.quantile(ea[ok], [0.5, 0.9, 0.99, 0.999, 1.0]))
order = np.argsort(-ea)[:12]
for b in order: print("env %4d abs err %.2e stiffness %.2e iters %d" % (b, ea[b], w[b], it[b]))
for th in (1e6, 1e7, 1e8, 1e9, 1e10):
    sel = ok & (w > th)
    rest = ok & ~sel
    print("threshold %.0e: %4d environments above (%.2f %%), max abs err of the others %.2e" % (th, sel.sum(), 100.0 * sel.mean(), ea[rest].max() if rest.any() else 0.0))
gm.close()
