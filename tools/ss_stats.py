"""body-body contact on the GPU against the oracle: per-step statistics (status / iteration agreement, state error of the agreeing environments).
usage: ss_stats.py [centred|offcentre]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
mode = sys.argv[1] if len(sys.argv) > 1 else "centred"
if mode == "centred":
    cases = [("nonlinear", "f64", "Fixed"), ("nonlinear", "f64", "Floating"), ("impact", "f64", "Fixed"), ("nonlinear", "f32", "Fixed")]
else:
    cases = [(ft, "f64", j) for ft in ("nonlinear", "linear") for j in ("Floating", "Revolute")]
for ft, dtype, joint in cases:
    rng = np.random.default_rng(17 if mode == "centred" else 5)
    if mode == "centred":
        B = 256
        spec = d.get_two_spheres(friction_type=ft, gravity=-9.81, joint_world_body1=joint)
        Z = np.zeros((B, 2, 13)); Z[:, :, 6] = 1.0
        dirs = rng.normal(size=(B, 3)); dirs[:, 2] = np.abs(dirs[:, 2]) + 0.3; dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        Z[:, 1, 0:3] = dirs * rng.uniform(1.05, 1.6, size=(B, 1))
        Z[:, 1, 3:6] = -dirs * rng.uniform(0.0, 3.0, size=(B, 1)) + 0.3 * rng.normal(size=(B, 3))
        Z[:, 1, 10:13] = rng.normal(size=(B, 3))
        Z = Z.reshape(B, -1); steps = 25
    else:
        from test_device_program_emu import off_centre_pair
        B = 64
        spec, z0 = off_centre_pair(ft, joint)
        Z = np.tile(z0, (B, 1)); Z[:, 16:19] += 0.2 * rng.normal(size=(B, 3)); Z[:, 23:26] += 0.5 * rng.normal(size=(B, 3)); steps = 20
    gm = api.BatchedMechanism(spec, B, dtype=dtype); o = Oracle(spec)
    z = Z.astype(np.float32).astype(np.float64) if dtype == "f32" else Z.copy()
    for k in range(steps):
        zg, st, it = gm.step(z, np.zeros((B, spec.nu)))
        zin = d.fp32_abi_state(z) if dtype == "f32" else z
        Zo, st_o, it_o = o.step_batch(zin, np.zeros((B, spec.nu)), nthreads=8)[:3]
        same = (st == 0) & (st_o == 0) & (it == it_o)
        both = (st == 0) & (st_o == 0)
        e_same = np.abs(zg[same] - Zo[same]).max() if same.any() else 0.0
        e_both = np.abs(zg[both] - Zo[both]).max() if both.any() else 0.0
        print("%s %s %s step %2d: gpu ok %3d orc ok %3d both %3d same-iters %3d | err same %.1e both %.1e | iters gpu %.1f orc %.1f max %d/%d" % (ft, dtype, joint, k, (st == 0).sum(), (st_o == 0).sum(), both.sum(), same.sum(), e_same, e_both, it.mean(), it_o.mean(), it.max(), it_o.max()), flush=True)
        z = zg.astype(np.float64)
    gm.close()
