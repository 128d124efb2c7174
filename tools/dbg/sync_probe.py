"""joined steps (a barrier per step) of the Ant batch with / without the iteration cap: wall time per step, status / iteration statistics"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dojo.jl_amd", "host")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
import dojo_amd as d
from dojo_amd import api
cap = int(sys.argv[1]) if len(sys.argv) > 1 else -1
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
spec = d.baseline_config(cfg)
dev = torch.device("cuda", 0); torch.cuda.init()
Z0, U0 = d.synthetic_inputs(spec, B)
z = torch.tensor(Z0, dtype=torch.float32, device=dev); zn = torch.empty_like(z)
rng = np.random.default_rng(5)
U = torch.tensor(0.5 * rng.standard_normal((K + 3, B, spec.nu)) * (np.abs(U0) > 0), dtype=torch.float32, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
dz = torch.empty((B, spec.nx, spec.nx), dtype=torch.float32, device=dev); du = torch.empty((B, max(spec.nu, 1), spec.nx), dtype=torch.float32, device=dev)
gm = api.BatchedMechanism(spec, B, dtype="f32")
gm.set_iteration_cap(cap)
if groups > 0: gm.set_groups(groups)
lib = api.lib(); p = lambda t: C.c_void_p(t.data_ptr())
ts = []
for k in range(K + 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    api._chk(lib.dojo_step_dev(gm.h, p(z), p(U[k]), p(zn), p(st), p(it), p(dz), p(du), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    z, zn = zn, z
    if k >= 3: print("step %2d: %.3f ms  max iters %d  >16: %d  failed %d  mean %.2f" % (k, ts[-1], int(it.max()), int((it > 16).sum()), int((st != 0).sum()), float(it.float().mean())))
print("cap %d groups %d: mean %.3f ms per joined step -> %.0f env-steps/s" % (cap, groups, np.mean(ts[3:]), B / np.mean(ts[3:]) * 1e3))
