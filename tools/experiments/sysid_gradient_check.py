"""examples/sphere_system_identification_device.py: the chained contact-data gradient against central differences of the cost, for the two gradient
modes (0: as get_contact_gradients evaluates after step!, 1: consistent IFT) and three solver tolerances.  Run from the repository root on a GPU box."""
import sys, os, numpy as np
sys.path.insert(0, "examples"); sys.path.insert(0, "dojo.jl_amd/host")
import dojo_amd as d
import sphere_system_identification_device as ex
Z = ex.dataset()
th = np.array([0.12, 0.47])
for tol in (None, 1e-8, 1e-10):
    opts = None if tol is None else d.SolverOptions(rtol=tol, btol=tol)
    for mode in (0, 1):
        c, g, H = ex.loss(np.concatenate([th, np.zeros(3)]), Z, derivatives=True, grad_mode=mode, opts=opts)
        for h in (1e-4, 1e-6):
            fd = np.array([(ex.loss(np.concatenate([th + h * e, np.zeros(3)]), Z, opts=opts) - ex.loss(np.concatenate([th - h * e, np.zeros(3)]), Z, opts=opts)) / (2 * h) for e in np.eye(2)])
            print("tol", tol, "mode", mode, "h", h, "g", g, "fd", fd)
