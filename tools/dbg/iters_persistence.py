"""does a solve that was long in one step stay long in the next?  (what dojo_set_dispatch_order's prediction rests on)"""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host"), ROOT]
import torch, numpy as np
import dojo_amd as d
from dojo_amd import api
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = 40
spec = d.baseline_config(cfg)
Z, U = d.synthetic_inputs(spec, B)
dev = torch.device("cuda:0"); torch.cuda.init()
gm = api.BatchedMechanism(spec, B, dtype="f32", device=0); gm.set_async(True)
z = torch.tensor(Z, dtype=torch.float32, device=dev); zn = torch.empty_like(z); u = torch.tensor(U, dtype=torch.float32, device=dev)
it = torch.zeros((K, B), dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
for k in range(K):
    api._chk(api.lib().dojo_step_dev(gm.h, p(z), p(u), p(zn), p(st), p(it[k]), C.c_void_p(0), C.c_void_p(0), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    z, zn = zn, z
gm.join(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
I = it.cpu().numpy().astype(float)
a, b = I[5:-1].ravel(), I[6:].ravel()
print("config %d B %d, steps 5..%d: mean iterations %.2f, max %d" % (cfg, B, K - 1, a.mean(), int(I.max())))
print("correlation of consecutive steps' iteration counts: %.3f" % np.corrcoef(a, b)[0, 1])
for t in (14, 20, 30, 45):
    n = (a >= t).sum()
    print("  P(next >= %2d | this >= %2d) = %.3f   (P(next >= %2d) = %.4f; %d such solves)" % (t, t, ((a >= t) & (b >= t)).sum() / max(n, 1), t, (b >= t).mean(), n))
# how much of the next step's top-1024 does this step's top-1024 hold?
hit = []
for k in range(5, K - 1):
    top_now = set(np.argsort(-I[k], kind="stable")[:B // 4]); top_next = np.argsort(-I[k + 1], kind="stable")[:B // 4]
    hit.append(np.mean([e in top_now for e in top_next]))
print("share of the next step's longest quarter that is in this step's longest quarter: %.3f (0.25 = chance)" % np.mean(hit))
worst = [int(np.argmax(I[k + 1])) for k in range(5, K - 1)]
rank = [int((I[k] > I[k][w]).sum()) for k, w in zip(range(5, K - 1), worst)]
print("rank (by this step's count) of the solve that is the longest of the next step: median %d of %d, quartiles %s" % (np.median(rank), B, np.percentile(rank, [25, 75]).astype(int).tolist()))
