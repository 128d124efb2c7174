import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dojo_amd as d
from oracle import Oracle
from emu_wrap import emu_step
spec = d.baseline_config(3); o = Oracle(spec)
B = 64; Z, U = d.synthetic_inputs(spec, B)
tot = 0; mism = 0; stm = 0; ez = []
for k in range(6):
    Zo, st_o, it_o, _, _ = o.step_batch(Z, U, nthreads=8)
    r = emu_step(spec, Z, U, quad=True)
    ok = (st_o == 0) & (r["status"] == 0)
    tot += B; mism += int((r["iters"] != it_o).sum()); stm += int((r["status"] != st_o).sum())
    ez.append(np.abs(r["z_next"] - Zo).max(axis=1)[ok])
    Z = Zo
ez = np.concatenate(ez)
print("env-steps %d  iteration-count mismatches %d (%.1f %%)  status mismatches %d  state err q50 %.1e max %.1e" % (tot, mism, 100.0 * mism / tot, stm, np.median(ez), ez.max()))
