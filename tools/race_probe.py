"""Repeat one step of a random two-wavefront mechanism and count result variations (race hunting)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dojo_amd as d
from dojo_amd import api
from random_mechanisms import random_mechanism
nb, seed, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
grad = len(sys.argv) > 4 and sys.argv[4] == "grad"
opts = d.SolverOptions(rtol=1e-9, btol=1e-9)
spec, z0, u0 = random_mechanism(seed, nb=nb)
if os.environ.get('RACE_MUT'):
    exec(os.environ['RACE_MUT'])
B = 64
rng = np.random.default_rng(seed)
Z = np.tile(z0, (B, 1)); U = np.tile(u0, (B, 1)) + rng.normal(size=(B, spec.nu)) * 0.2
gm = api.BatchedMechanism(spec, B, dtype="f64", opts=opts)
ref = None; nvar = 0; nit = 0
for r in range(reps):
    zn, st, it = gm.step(Z, U, with_gradient=grad)
    g = gm.gradients()[0] if grad else None
    if ref is None:
        ref = (zn.copy(), it.copy(), None if g is None else g.copy())
    else:
        dv = np.abs(zn - ref[0]).max(axis=1)
        bad = (dv > 0) | (it != ref[1])
        if g is not None:
            bad |= (np.abs(g - ref[2]).reshape(B, -1).max(axis=1) > 0)
        nvar += int(bad.sum()); nit += int((it != ref[1]).sum())
print(os.environ.get("RACE_MUT", ""), "nb=%d seed=%d: %d of %d env-steps differ from the first run (%d with another iteration count)" % (nb, seed, nvar, (reps - 1) * B, nit))
