"""Pins the oracle's data Jacobian (∂ residual / ∂ θ): restatement of test/data.jl:82-126.

  simulate tsim with u = 0.2 on the actuated inputs, rtol = btol = eps, then
  || FD(d full_vector / d data) · attjac  −  jacobian_data!(dense) ||_inf(entries) < 1e-6
"""
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle


def fd_data_jacobian(o, data, sol, delta=1.0e-5):
    n, nd = o.n, len(data)
    J = np.zeros((n, nd))
    for i in range(nd):
        dp, dm = data.copy(), data.copy()
        dp[i] += delta; dm[i] -= delta
        J[:, i] = (o.evaluate_residual(dp, sol) - o.evaluate_residual(dm, sol)) / (2 * delta)
    return J


def run_data(spec, tsim=0.1, eps=1.0e-6):
    o = Oracle(spec, opts=d.SolverOptions(rtol=eps, btol=eps))
    z0 = d.initialize(spec)
    steps = int(np.ceil(tsim / spec.timestep))
    u = 0.2 * np.ones(spec.nu)
    if spec.joints[0].nu == 6:          # ctrl!: no input on a floating base (test/data.jl:69-79)
        u[:6] = 0.0
    o.simulate(z0, steps, control=lambda o_, k: u)
    data0 = o.get_data()
    sol0 = o.get_solution()
    fd = fd_data_jacobian(o, data0, sol0)
    o.set_data(data0); o.set_solution(sol0)
    fd = fd @ o.data_attjac()
    an = o.data_matrix()
    return np.abs(fd - an).max()


CASES = [
    ("pendulum", dict(), 0.1),
    ("pendulum", dict(springs=2.0, dampers=0.3), 0.1),
    ("pendulum", dict(springs=2.0, dampers=0.3, joint_limits={"joint": [-0.3, 0.9]}), 0.3),
    ("block", dict(contact=False), 0.1),
    ("block", dict(), 0.1),
    ("block", dict(), 0.6),
    ("ant", dict(timestep=0.01), 0.1),
    ("ant", dict(timestep=0.01), 0.4),
    ("quadruped", dict(), 0.1),
    ("quadruped", dict(parse_springs=False, parse_dampers=False, springs=2.0, dampers=0.3), 0.3),
    ("atlas", dict(), 0.1),
    ("atlas", dict(parse_springs=False, parse_dampers=False, springs=2.0, dampers=0.3), 0.1),
    # translational springs and dampers (spring_jacobian_configuration / damper_jacobian_configuration of translational/*.jl)
    ("slider", dict(springs=2.0, dampers=0.3), 0.1),
    ("nslider", dict(num_bodies=3, springs=1.0, dampers=1.0), 0.1),
    ("raiberthopper", dict(timestep=0.01, springs=(0.0, 5.0), dampers=(0.0, 0.5)), 0.1),
    ("raiberthopper", dict(timestep=0.01), 0.5),
    # test/data.jl:28-31: snake, 3 bodies, springs = dampers = 1; joint prototypes with free translations and two free rotations
    ("snake", dict(num_bodies=3, springs=1.0, dampers=1.0), 0.1),
    ("snake", dict(num_bodies=3, springs=1.0, dampers=1.0, joint_type="PlanarAxis"), 0.1),
    ("twister", dict(num_bodies=4, springs=1.0, dampers=0.5, joint_type="PrismaticOrbital"), 0.1),
    ("npendulum", dict(num_bodies=3, springs=1.0, dampers=0.5, rest_joint_type="CylindricalFree"), 0.1),
]


@pytest.mark.parametrize("name,kw,tsim", CASES)
def test_data_jacobian(name, kw, tsim):
    err = run_data(d.get_mechanism(name, **kw), tsim)
    assert err < 1.0e-6, err


def test_data_jacobian_translational_limits():
    spec = d.get_slider(joint_limits={"joint": [-0.2, 0.3]}, dampers=0.1, springs=0.5)
    assert run_data(spec, 0.05) < 1.0e-6 and run_data(spec, 0.4) < 1.0e-6        # free, and resting on the lower stop
