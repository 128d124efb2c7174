#!/usr/bin/env python3
"""The reference's system-identification example on the device, with the reference's OWN dataset.

/root/reference/examples/system_identification/synthetic_sphere.jl:45-100 + utilities.jl learns the friction coefficient and the contact
radius of the `:sphere` mechanism from ten stored trajectories of the reference's simulate! (tests/golden/reference_sphere.npz, extracted from
the .jld2 the reference ships by tools/jld2_reader.py): three-step predictions from rows 10..12 of every trajectory, cost

    f(θ) = Σ_traj Σ_i  ½ (z_pred − z_true)ᵀ Q (z_pred − z_true),      Q = diag(1 1 1, .1 .1 .1, 1 1 1 1, .1 .1 .1),

gradient and Gauss-Newton Hessian through `get_contact_gradients` (src/gradients/contact.jl:1-55) chained over the steps
(utilities.jl:42-90), minimised by the quasi-Newton loop of utilities.jl:1-38 from the guess [0, 1] inside the box [0, 0.8] x [0.05, 1].

Here: the ten trajectories are ONE batch (B = 10); `dojo_step_dev` + `dojo_gradients` + `dojo_contact_gradients` give z_pred, ∂z'/∂z and
∂z'/∂θ for all of them per step; the contact data θ = [friction_coefficient, contact_radius, contact_origin(3)] is mechanism data, so every
cost evaluation builds a handle for its θ (set_data!(mechanism.contacts, θ), utilities.jl:52).  The data were generated with θ* = [0.2, 0.5].

    python examples/sphere_system_identification_device.py          # needs a GPU: libdojo_hip has no CPU fallback
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import dojo_amd as d                                   # noqa: E402
from dojo_amd import api, mechanisms                  # noqa: E402

TIMESTEPS = (10, 11, 12)                                # synthetic_sphere.jl:14 (1-based Storage rows)
Q = np.diag([1, 1, 1, .1, .1, .1, 1, 1, 1, 1, .1, .1, .1])      # utilities.jl:61 over z = [x v q ω]


def dataset():
    f = np.load(os.path.join(ROOT, "tests", "golden", "reference_sphere.npz"))
    return np.concatenate([f["x"], f["v"], f["q"], f["ω"]], axis=-1)[:, 0]           # [10, 100, 13]


def sphere(theta):
    """get_mechanism(:sphere; timestep=0.02, gravity=-9.81, friction_coefficient=0.2, radius=0.5) with set_data!(contacts, [θ; 0 0 0])"""
    spec = mechanisms.get_sphere(timestep=0.02, gravity=-9.81, friction_coefficient=0.2, radius=0.5)
    spec.contacts[0].friction_coefficient = float(theta[0])
    spec.contacts[0].radius = float(theta[1])
    return spec


def attitude_jacobian(z):
    """attitude_jacobian(z, 1) (src/mechanism/methods.jl:44-60): d z(13) / d [x v φ ω](12), LVᵀmat(q) for the quaternion rows"""
    s, v1, v2, v3 = z[6:10]
    G = np.zeros((13, 12))
    G[0:6, 0:6] = np.eye(6)
    G[6:10, 6:9] = np.array([[-v1, -v2, -v3], [s, -v3, v2], [v3, s, -v1], [-v2, v1, s]])
    G[10:13, 9:12] = np.eye(3)
    return G


def loss(theta, Z, derivatives=False, grad_mode=0, opts=None):
    """loss(mechanism, θ, storage, timesteps) of utilities.jl:55-90, summed over the trajectories (synthetic_sphere.jl:52-74)"""
    B = len(Z)
    gm = api.BatchedMechanism(sphere(theta), B, dtype="f64", opts=opts)          # opts None: the reference's default SolverOptions, as the example
    gm.set_gradient_mode(grad_mode)                     # 0: as get_contact_gradients evaluates after step! (the example's); 1: the consistent IFT
    zp = Z[:, TIMESTEPS[0] - 1].copy()
    cost = 0.0
    g = np.zeros(5); H = np.zeros((5, 5))
    dc = np.zeros((B, 12, 5))
    for i in TIMESTEPS:
        ztrue = Z[:, i]                                   # row i+1 (1-based)
        zp, st, it = gm.step(zp, with_gradient=derivatives)
        assert (st == 0).all()
        e = zp - ztrue
        cost += 0.5 * np.einsum("bi,ij,bj->", e, Q, e)
        if derivatives:
            dz, _ = gm.gradients()                        # [B, 12, 12]  jacobian_state
            dth = gm.contact_gradients()                  # [B, 12, 5]   jacobian_contact
            dc = dth + dz @ dc
            for b in range(B):
                J = attitude_jacobian(zp[b]) @ dc[b]
                g += J.T @ Q @ e[b]
                H += J.T @ Q @ J
    gm.close()
    return (cost, g[:2], H[:2, :2]) if derivatives else cost


def quasi_newton_solve(f, fgH, x0, iters=20, gtol=1e-8, ftol=1e-6, lower=(0.0, 0.05), upper=(0.8, 1.0), reg=1e-9, verbose=True):
    """utilities.jl:1-38 (Nocedal & Wright, algorithm 6.1, with the clamped line search of the example)"""
    lo, hi = np.array(lower), np.array(upper)
    x = np.array(x0, float); ls_failure = False; reg_min, reg_max = 1e-9, 1e6
    for k in range(iters):
        fe, ge, He = fgH(x)
        He = He + reg * np.eye(len(x))
        reg = float(np.clip(reg * 2 if ls_failure else reg / 1.5, reg_min, reg_max))
        if np.abs(ge).max() < gtol or fe < ftol:
            break
        p = -np.linalg.solve(He, ge)
        alpha, ls_failure = 1.0, False
        for kk in range(4):
            if f(np.clip(x + alpha * p, lo, hi)) <= fe:
                break
            alpha /= 3
            if kk == 3:
                alpha = 0.001 / np.abs(p).max(); ls_failure = True
        x = np.clip(x + alpha * p, lo, hi)
        if verbose:
            print("k: %2d   f: %.3e   theta = [%.6f, %.6f]" % (k + 1, fe, x[0], x[1]))
    return x


def main():
    Z = dataset()
    f0 = lambda th: loss(np.concatenate([th, np.zeros(3)]), Z)
    fgH0 = lambda th: loss(np.concatenate([th, np.zeros(3)]), Z, derivatives=True)
    print("cost at the parameters the data were generated with, f([0.2, 0.5]) = %.3e" % f0(np.array([0.2, 0.5])))
    guess = np.array([0.0, 1.0])
    print("cost at the guess, f([0, 1]) = %.3e" % f0(guess))
    sol = quasi_newton_solve(f0, fgH0, guess)
    print("solution: friction_coefficient %.6f (0.2), contact_radius %.6f (0.5), cost %.3e" % (sol[0], sol[1], f0(sol)))


if __name__ == "__main__":
    main()
