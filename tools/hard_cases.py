"""CPU loop for the IFT accuracy work: the hard Ant environment-steps the GPU hunts dumped (tools/hunt_parity.py ->
tests/golden/hard_cases_ant.npz) through the SIMT emulator (the shipped device source on CPU threads) against the oracle.
usage: hard_cases.py [npz] [n]      prints per case the relative and the absolute gradient inf-norm error"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dojo_amd as d
from oracle import Oracle
from emu_wrap import emu_step

f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "hard_cases_ant.npz")
D = np.load(f)
Z, U = D["z"], D["u"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(Z)
order = np.argsort(-D["meta"][:, 3]) if "meta" in D else np.arange(len(Z))
Z, U = Z[order][:n], U[order][:n]
spec = d.baseline_config(3)
o = Oracle(spec)
Zo, st_o, it_o, dz_o, du_o = o.step_batch(Z, U, with_grad=True, nthreads=os.cpu_count() or 8)
r = emu_step(spec, Z, U, grad=True, quad=True)
worst = 0.0
for b in range(len(Z)):
    ez = np.abs(r["z_next"][b] - Zo[b]).max()
    ea = max(np.abs(r["dz"][b] - dz_o[b]).max(), np.abs(r["du"][b] - du_o[b]).max())
    er = max(np.abs(r["dz"][b] - dz_o[b]).max() / max(1.0, np.abs(dz_o[b]).max()), np.abs(r["du"][b] - du_o[b]).max() / max(1.0, np.abs(du_o[b]).max()))
    worst = max(worst, er)
    print("case %2d  st %d/%d it %2d/%2d  state %.1e  grad rel %.2e abs %.2e  |J| %.1e" % (b, r["status"][b], st_o[b], r["iters"][b], it_o[b], ez, er, ea, np.abs(dz_o[b]).max()))
print("worst rel %.3e" % worst)
