"""One-off hunt on the GPU: the fp32-ABI differentiable step at the BASELINE batch against the oracle on the state the fp32 buffers stand for;
dumps the environments with the largest gradient error (gpurun_out/hunt_f32.npz) for analysis under the emulator."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
spec = d.baseline_config(3); B = 4096
Z, U = d.synthetic_inputs(spec, B)
gm = api.BatchedMechanism(spec, B, dtype="f64")
for _ in range(8):
    Z, st, it = gm.step(Z, U)
gm.close()
Zf = Z.astype(np.float32); Uf = U.astype(np.float32)
g32 = api.BatchedMechanism(spec, B, dtype="f32")
zn, st, it = g32.step(Zf, Uf, with_gradient=True); dz, du = g32.gradients(); g32.close()
o = Oracle(spec)
Zo, st_o, it_o, dz_o, du_o = o.step_batch(d.fp32_abi_state(Zf.astype(np.float64)), Uf.astype(np.float64), with_grad=True, nthreads=os.cpu_count() or 8)
ok = (st == 0) & (st_o == 0)
eg = np.array([max(np.abs(dz[b] - dz_o[b]).max() / max(1.0, np.abs(dz_o[b]).max()), np.abs(du[b] - du_o[b]).max() / max(1.0, np.abs(du_o[b]).max())) if ok[b] else 0.0 for b in range(B)])
es = np.abs(zn.astype(np.float64) - Zo).max(axis=1)
top = np.argsort(-eg)[:12]
for b in top: print("env %4d grad err %.2e state err %.2e iters %d/%d |J| %.1e" % (b, eg[b], es[b], it[b], it_o[b], np.abs(dz_o[b]).max()))
np.savez(os.path.join(ROOT, "gpurun_out", "hunt_f32.npz"), z=Zf[top].astype(np.float64), u=Uf[top].astype(np.float64), eg=eg[top], env=top, dz=dz[top[:2]], dz_o=dz_o[top[:2]])
