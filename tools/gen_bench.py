"""Throughput of the general lane-mapping builds (cut elements, several limits per joint) on the GPU -- capability builds, not tuned:
four-bar linkage (a kinematic loop, fwd and fwd + IFT), two free spheres (a body-body contact between bodies of different trees),
a Spherical joint with three limits.  Prints env-steps/s at B = 4096 (host arrays in and out: includes the copies)."""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dojo.jl_amd", "host")]
import numpy as np
import dojo_amd as d
from dojo_amd import api

B = int(os.environ.get("B", 4096))
rng = np.random.default_rng(0)


def run(name, spec, z, U, grad, steps=10):
    gm = api.BatchedMechanism(spec, B, dtype="f64")
    gm.set_groups(1)            # one launch on one queue for every case (INTEGRATION.md "Device memory": a queue that has run a general build keeps its scratch)
    for k in range(2):
        z1, st, it = gm.step(z, U, with_gradient=grad)
    t0 = time.perf_counter()
    zz = z
    for k in range(steps):
        zz, st, it = gm.step(zz, U, with_gradient=grad)
    dt = (time.perf_counter() - t0) / steps
    print("%-40s B %d  %s  %.3f ms/step  %.0f env-steps/s  converged %.4f  mean iters %.1f" % (name, B, "fwd+IFT" if grad else "fwd    ", 1e3 * dt, B / dt, (st == 0).mean(), it.mean()))
    gm.close()


spec = d.get_fourbar(timestep=0.01)
z = np.stack([d.initialize(spec, inner_angle=0.15 + 0.3 * rng.random(), base_angle=np.pi / 4 + 0.3 * rng.standard_normal()) for _ in range(B)])
U = rng.standard_normal((B, spec.nu)) * np.array([1.0, 0.3, 1.0, 0.3, 0.5])
ONLY = os.environ.get("ONLY", "")
if ONLY in ("", "loop"): run("fourbar (loop)", spec, z, U, False)
if ONLY in ("", "loopgrad"): run("fourbar (loop)", spec, z, U, True)
spec = d.get_two_spheres(friction_type="nonlinear", gravity=-9.81, joint_world_body1="Floating", free_on="world")
Z = np.zeros((B, 2, 13)); Z[:, :, 6] = 1.0
dirs = rng.normal(size=(B, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
Z[:, 1, 0:3] = dirs * rng.uniform(1.02, 1.2, size=(B, 1)); Z[:, 1, 3:6] = -dirs * rng.uniform(0.5, 3.0, size=(B, 1))
spec_e = d.get_two_spheres(friction_type="nonlinear", gravity=-9.81, joint_world_body1="Floating", free_on="body1")
if ONLY in ("", "edge"): run("two spheres, tree edge (quad build)", spec_e, Z.reshape(B, -1), np.zeros((B, spec.nu)), False)
if ONLY in ("", "cc"): run("two free spheres (cut contact)", spec, Z.reshape(B, -1), np.zeros((B, spec.nu)), False)
