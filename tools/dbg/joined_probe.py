"""joined-per-step time of one handle under different group settings (after an asynchronous phase, like bench.py's sync leg)"""
import os, sys, time, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host"), ROOT]
import torch, numpy as np
import dojo_amd as d
from dojo_amd import api
spec = d.baseline_config(3); B = 4096
Z, U = d.synthetic_inputs(spec, B)
dev = torch.device("cuda:0"); torch.cuda.init()
gm = api.BatchedMechanism(spec, B, dtype="f32", device=0)
lib = api.lib()
z = torch.tensor(Z, dtype=torch.float32, device=dev); zn = torch.empty_like(z); u = torch.tensor(U, dtype=torch.float32, device=dev)
nx = 12 * spec.Nb
dz = torch.empty((B, nx, nx), dtype=torch.float32, device=dev); du = torch.empty((B, spec.nu, nx), dtype=torch.float32, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
def step():
    global z, zn
    api._chk(lib.dojo_step_dev(gm.h, p(z), p(u), p(zn), p(st), p(it), p(dz), p(du), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    z, zn = zn, z
zsave = None
def timed(n, label):
    global z, zsave
    if zsave is not None: z.copy_(zsave)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    gm.join(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (label, 1e3 * (time.perf_counter() - t) / n), flush=True)
gm.set_async(True); timed(10, "warmup async"); zsave = z.clone(); timed(20, "async, default groups")
gm.set_async(False); timed(20, "joined, default groups")
for g in (4, 16, 0, 4, 3, 2, 0):
    gm.set_groups(g); timed(20, "joined, set_groups(%d)" % g)
