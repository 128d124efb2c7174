#!/usr/bin/env python3
"""The reference's LQR example on the device.

/root/reference/examples/control/cartpole_lqr.jl:9-31 (and docs/src/creating_simulation/define_controller.md):

    mechanism = get_mechanism(:cartpole)
    A, B = get_minimal_gradients!(mechanism, zeros(4), zeros(2))
    K = lqr(Discrete, A, B[:, 1], I(4), I(1))                 # the docs print K = [-0.948838; -2.54837; 48.6627; 10.871]
    controller!(mechanism, k) = set_input!(cart_joint, -K' * get_minimal_state(mechanism))
    initialize!(mechanism, :cartpole; position=0, orientation=pi/4);  simulate!(mechanism, 10.0, controller!)

Here: `dojo_minimal_gradients` gives A and B (the IFT Jacobians of one step in minimal coordinates, computed on the GPU), scipy solves the
discrete Riccati equation, and the closed loop runs as ONE batch of start angles through `dojo_step_minimal` -- the policy needs the state of
every step, so the steps are joined (the "sync_per_step" regime of bench.py).

    python examples/cartpole_lqr_device.py [batch]          # needs a GPU: libdojo_hip has no CPU fallback
"""
import os
import sys
import time

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import dojo_amd as d                                   # noqa: E402
from dojo_amd import api                               # noqa: E402

K_DOCS = np.array([-0.948838, -2.54837, 48.6627, 10.871])          # define_controller.md:23


def lqr_discrete(A, B, Q, R):
    """lqr(Discrete, A, B, Q, R) of ControlSystemsBase"""
    P = scipy.linalg.solve_discrete_are(A, B, Q, R)
    return np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    spec = d.get_cartpole()
    gm = api.BatchedMechanism(spec, B, dtype="f64")
    # linearization at the upright equilibrium (one environment would do; the batch gives B copies of the same Jacobians)
    xn, st, it, jx, ju = gm.minimal_gradients(np.zeros((B, 4)), np.zeros((B, 2)))
    assert np.all(st == 0)
    K = lqr_discrete(jx[0], ju[0][:, :1], np.eye(4), np.eye(1))[0]
    print("K (device gradients):", K)
    print("K (reference docs):  ", K_DOCS, " relative difference %.1e" % np.abs(K / K_DOCS - 1).max())
    # the closed loop from a spread of start angles (the docs' pi/4 first), 10 s = 1000 steps
    rng = np.random.default_rng(0)
    X = np.zeros((B, 4)); X[:, 2] = rng.uniform(-np.pi / 4, np.pi / 4, size=B); X[0, 2] = np.pi / 4
    t0 = time.perf_counter()
    reach = np.zeros(B)
    for k in range(1000):
        U = np.zeros((B, 2)); U[:, 0] = -(X @ K)
        X, st, it = gm.step_minimal(X, U)
        assert np.all(st == 0), k
        reach = np.maximum(reach, np.abs(X[:, 0]))
    el = time.perf_counter() - t0
    print("after 10 s: max |x| over the batch %.3e (cart, cart velocity, angle, angular velocity of the docs' start: %s)" % (np.abs(X).max(), np.round(X[0], 4)))
    print("the cart of the docs' start swung out %.2f m; %d environments x 1000 joined steps in %.2f s = %.0f env-steps/s (host arrays in and out every step)" % (reach[0], B, el, B * 1000 / el))
    gm.close()


if __name__ == "__main__":
    main()
