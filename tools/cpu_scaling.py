"""thread scaling of the C++ oracle's timed loop (bench.py cpu_baseline) on this host: env-steps/s against threads"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
import oracle as orc
from oracle import Oracle
spec = d.baseline_config(3); o = Oracle(spec); o.set_refine_steps(0)
cores = orc.physical_cores()
print("physical cores", cores, "logical", os.cpu_count())
try: print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup cpu.max", e)
for sparse in (True, False):
    o.set_sparse_solver(sparse); base = None
    nt = 1
    while nt <= cores:
        Z, U = d.synthetic_inputs(spec, 32 * nt)
        el = o.time_batch(Z, U, with_grad=True, nthreads=nt)
        v = 32 * nt / el; base = base or v
        print("%s %4d threads: %9.1f env-steps/s  efficiency %.2f" % ("sparse" if sparse else "dense ", nt, v, v / (nt * base)), flush=True)
        nt *= 2
