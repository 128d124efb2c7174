"""GPU probe: f32-ABI handle vs f64-ABI handle vs oracle on the same fp32-representable inputs (Ant, default options)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spec = d.baseline_config(3)
Z, U = d.synthetic_inputs(spec, B)
g64 = api.BatchedMechanism(spec, B, dtype="f64"); g32 = api.BatchedMechanism(spec, B, dtype="f32")
for _ in range(8):
    Z, st, it = g64.step(Z, U)
Z32 = Z.astype(np.float32); U32 = U.astype(np.float32)
Zr = d.fp32_abi_state(Z32); Ur = U32.astype(np.float64)
za, sa, ia = g64.step(Zr, Ur, with_gradient=True); dza, dua = g64.gradients()
zb, sb, ib = g32.step(Z32, U32, with_gradient=True); dzb, dub = g32.gradients()
o = Oracle(spec)
zo, so, io, dzo, duo = o.step_batch(Zr, Ur, with_grad=True, nthreads=os.cpu_count())
ok = (sa == 0) & (sb == 0) & (so == 0)
rel = lambda a, b: np.array([np.abs(a[i] - b[i]).max() / max(1.0, np.abs(b[i]).max()) for i in np.nonzero(ok)[0]])
e1 = rel(dzb.astype(np.float64), dza); e2 = rel(dza, dzo); e3 = rel(dzb.astype(np.float64), dzo)
print("iters differ f32/f64:", int((ia[ok] != ib[ok]).sum()), " f64/orc:", int((ia[ok] != io[ok]).sum()))
print("state  |f32-f64| max %.2e   |f64-orc| max %.2e   max|z| %.2e" % (np.abs(zb.astype(np.float64) - za)[ok].max(), np.abs(za - zo)[ok].max(), np.abs(za[ok]).max()))
for n, e in (("dz f32 vs f64", e1), ("dz f64 vs orc", e2), ("dz f32 vs orc", e3)):
    print("%s: q50 %.1e q99 %.1e max %.1e  n>1e-3 %d" % (n, np.quantile(e, .5), np.quantile(e, .99), e.max(), int((e > 1e-3).sum())))
w = np.argmax(e3); idx = np.nonzero(ok)[0][w]
print("worst env", idx, "max|dz| orc %.2e f64 %.2e f32 %.2e  iters %d" % (np.abs(dzo[idx]).max(), np.abs(dza[idx]).max(), np.abs(dzb[idx]).max(), ia[idx]))
