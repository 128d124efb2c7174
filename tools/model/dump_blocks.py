import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dojo_amd as d
from emu_wrap import emu_step
D_ = np.load(sys.argv[1]); ci = int(sys.argv[2])
order = np.argsort(-D_["meta"][:, 3])
spec = d.baseline_config(3)
os.environ["EMU_IFT_LU_W"] = "0"; os.environ["DJ_DUMP_BLOCKS"] = "1"
r = emu_step(spec, D_["z"][order[ci]][None], D_["u"][order[ci]][None], grad=True, quad=True)
np.savez(sys.argv[3], dz=r["dz"][0], du=r["du"][0])
