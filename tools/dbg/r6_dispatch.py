"""joined-per-step time of one handle under dojo_set_dispatch_order 0 / 1 / 2, per group setting; asynchronous too"""
import os, sys, time, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host"), ROOT]
import torch, numpy as np
import dojo_amd as d
from dojo_amd import api
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kw = {"distribution": sys.argv[3]} if len(sys.argv) > 3 else {}
spec = d.baseline_config(cfg)
Z, U = d.synthetic_inputs(spec, B, **kw)
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
def timed(n, label, sync=False):
    global z, zsave
    if zsave is not None: z.copy_(zsave)
    step(); step()                                   # (the first sorted step has no permutation yet)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        step()
        if sync: torch.cuda.synchronize()
    gm.join(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t) / n
    try: kt = gm.last_kernel_times()
    except Exception: kt = (float("nan"), float("nan"))
    print("%-58s %.3f ms/step  %8.0f env-steps/s   iters mean %.2f max %d   last launch: step %.3f ift %.3f ms" % (label, ms, B / ms * 1e3, it.float().mean().item(), it.max().item(), kt[0], kt[1]), flush=True)
print("config %d, B = %d %s" % (cfg, B, kw))
gm.set_async(True); timed(10, "warmup async"); zsave = z.clone()
if os.environ.get('CAP'): gm.set_iteration_cap(int(os.environ['CAP'])); print('iteration cap', os.environ['CAP'])
for rep in range(2):
    for mode in ((2,) if os.environ.get('ONLY_SORTED') else (0, 2)):
        gm.set_dispatch_order(mode)
        gm.set_async(True); gm.set_groups(0); timed(20, "async, default groups, order %d" % mode)
        gm.set_async(False)
        for g in (0, 1, 2, 3, 4, 8, 16):
            gm.set_groups(g); timed(20, "joined, set_groups(%d), order %d" % (g, mode))
        gm.set_groups(1); timed(20, "joined + host sync per step, groups 1, order %d" % mode, sync=True)
        gm.set_groups(4); timed(20, "joined + host sync per step, groups 4, order %d" % mode, sync=True)
