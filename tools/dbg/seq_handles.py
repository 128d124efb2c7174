import os, sys, time
sys.path[:0] = ["/root/repo/dojo.jl_amd/host"]
import numpy as np, dojo_amd as d
from dojo_amd import api
B = 4096
rng = np.random.default_rng(0)
fb = d.get_fourbar(timestep=0.01)
zf = np.stack([d.initialize(fb, inner_angle=0.15 + 0.3 * rng.random(), base_angle=np.pi / 4) for _ in range(B)])
Uf = rng.standard_normal((B, fb.nu))
ts = d.get_two_spheres(friction_type="nonlinear", gravity=-9.81, joint_world_body1="Floating", free_on="world")
te = d.get_two_spheres(friction_type="nonlinear", gravity=-9.81, joint_world_body1="Floating", free_on="body1")
Z = np.zeros((B, 2, 13)); Z[:, :, 6] = 1.0; Z[:, 1, 2] = 1.1; Z[:, 1, 5] = -1.0
BMAX = B
for tag in sys.argv[1].split(","):
    B = BMAX
    if "@" in tag: tag, b_ = tag.split("@"); B = int(b_)
    if tag == "sleep": time.sleep(5); continue
    if tag in ("loop", "loopgrad"): spec, z, U, g = fb, zf[:B], Uf[:B], tag == "loopgrad"
    elif tag == "cc": spec, z, U, g = ts, Z.reshape(BMAX, -1)[:B], np.zeros((B, ts.nu)), False
    elif tag.startswith("edge"): spec, z, U, g = te, Z.reshape(BMAX, -1)[:B], np.zeros((B, ts.nu)), False
    elif tag.startswith("ant"): spec = d.baseline_config(3); z, U = d.synthetic_inputs(spec, B); g = True
    gm = api.BatchedMechanism(spec, B, dtype="f64")
    if tag[-1] in "12": gm.set_groups(int(tag[-1]))
    for k in range(3): zz, st, it = gm.step(z, U, with_gradient=g)
    gm.close()
    print(tag, "ok", flush=True)
