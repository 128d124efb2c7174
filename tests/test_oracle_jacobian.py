"""Pins the oracle's system matrix: restatement of the reference's test/jacobian.jl:1-117.

  simulate tsim with u = 0.1 on every input, rtol = btol = eps, then
  || FD(d full_vector / d solution) + full_matrix(system) ||_inf(entries) < eps
(full_vector = -residual, hence the '+').  "Flying" = 0.1 s, "In contact" = 0.4 s.
"""
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle

EPS = 1.0e-7


def fd_solution_matrix(o, data, sol, delta=1.0e-5):
    n = len(sol)
    J = np.zeros((n, n))
    for i in range(n):
        sp, sm = sol.copy(), sol.copy()
        sp[i] += delta; sm[i] -= delta
        J[:, i] = (o.evaluate_residual(data, sp) - o.evaluate_residual(data, sm)) / (2 * delta)
    return J


def run_solmat(spec, tsim, eps=EPS):
    o = Oracle(spec, opts=d.SolverOptions(rtol=eps, btol=eps))
    z0 = d.initialize(spec)
    steps = int(np.ceil(tsim / spec.timestep))
    u = 0.1 * np.ones(spec.nu)
    traj, status = o.simulate(z0, steps, control=lambda o_, k: u)
    # the reference test does not assert solver success either (test/jacobian.jl:19-23)
    data = o.get_data()
    o.set_data(data)
    sol = o.get_solution()
    solmat = o.full_matrix()
    fd = fd_solution_matrix(o, data, sol)
    err = np.abs(fd + solmat).max()
    return err, traj


CASES = [
    ("pendulum", dict(springs=1.0, dampers=0.2)),
    ("pendulum", dict()),
    ("block", dict()),
    ("block", dict(contact_type="impact")),      # test/jacobian.jl:93,114: contact_type=:impact (ImpactContact, src/contacts/impact.jl)
    ("ant", dict(timestep=0.01)),
    ("quadruped", dict()),
    ("quadruped", dict(parse_springs=False, parse_dampers=False, springs=1.0, dampers=0.2)),
    ("atlas", dict(parse_dampers=False)),
    ("atlas", dict()),
    # test/jacobian.jl:86,89,106,109: translational springs and dampers (Prismatic joints)
    ("slider", dict(springs=1.0, dampers=0.2)),
    ("nslider", dict(springs=1.0, dampers=0.2)),
    ("raiberthopper", dict(timestep=0.01)),
    # test/jacobian.jl:85,88,90 (and 105,108,110): snake, npendulum, twister with springs = 1, dampers = 0.2
    ("snake", dict(springs=1.0, dampers=0.2)),
    ("npendulum", dict(springs=1.0, dampers=0.2)),
    ("twister", dict(springs=1.0, dampers=0.2)),
    ("sphere", dict()),
    ("sphere", dict(contact_type="linear")),     # test/jacobian.jl:92,113: contact_type=:linear (LinearContact, src/contacts/linear.jl)
    ("sphere", dict(contact_type="impact")),     # test/jacobian.jl:93,114
    ("block", dict(contact_type="linear")),
    ("cartpole", dict(dampers=0.1)),
    ("block2d", dict()),
    ("dzhanibekov", dict()),
    ("tippetop", dict()),
]


@pytest.mark.parametrize("name,kw", CASES)
def test_solmat_flying(name, kw):
    err, _ = run_solmat(d.get_mechanism(name, **kw), 0.1)
    assert err < EPS, err


@pytest.mark.parametrize("name,kw", [c for c in CASES if c[0] != "atlas"] + [("atlas", dict())])
def test_solmat_in_contact(name, kw):
    err, traj = run_solmat(d.get_mechanism(name, **kw), 0.4)
    assert err < EPS, err


def test_solmat_pendulum_with_limits():
    spec = d.get_pendulum(joint_limits={"joint": [-0.3, 0.25 * np.pi]}, dampers=0.1)
    err, _ = run_solmat(spec, 0.4)
    assert err < EPS, err


def test_solmat_translational_limits():
    """Limits on a translational coordinate (src/joints/limits.jl with the Prismatic coordinate): the slider resting on its
    lower limit, and the raiberthopper with its leg pushed against the stop."""
    spec = d.get_slider(joint_limits={"joint": [-0.2, 0.3]}, dampers=0.1, springs=0.5)
    err, traj = run_solmat(spec, 0.4)
    assert err < EPS and abs(traj[-1][2] + 0.7) < 1e-6, err          # z = −(0.5 + 0.2): on the lower stop
    from dojo_amd.mechanisms import set_limits
    spec = d.get_raiberthopper(timestep=0.01); set_limits(spec, {"leg": [-0.6, -0.4]})
    err, _ = run_solmat(spec, 0.3)
    assert err < EPS, err


@pytest.mark.parametrize("kind", ["spherical", "planar", "cylindrical", "mixed"])
def test_solmat_limits_on_several_coordinates(kind):
    """Limits on ALL free coordinates of a joint half and on both halves (src/joints/limits.jl:1-61 in general: three rotation-vector limits
    on a Spherical joint, two on a Planar joint's translation, one + one on a Cylindrical; dojo_amd.mechanisms.get_limited_chain) -- the
    identity of test/jacobian.jl on the oracle while the mechanism is driven into its stops, the limited coordinates stay inside their bounds,
    and with the drive switched off the Planar plate comes to rest ON its lower stop (test/joint_limits.jl's behaviour for a pendulum)."""
    spec = d.get_limited_chain(kind)
    o = Oracle(spec, opts=d.SolverOptions(rtol=1e-9, btol=1e-9))
    rng = np.random.default_rng(1)
    z = d.initialize(spec)
    u = 2.0 * rng.standard_normal(spec.nu)
    worst = 0.0; active = 0
    lo = np.concatenate([np.concatenate([h.limits[0] for h in (j.tra, j.rot) if h.limits is not None]) for j in spec.joints if j.tra.limits is not None or j.rot.limits is not None])
    hi = np.concatenate([np.concatenate([h.limits[1] for h in (j.tra, j.rot) if h.limits is not None]) for j in spec.joints if j.tra.limits is not None or j.rot.limits is not None])
    for k in range(120):
        z, info = o.step(z, u)
        assert info["status"] == 0
        x = o.maximal_to_minimal(z)
        th = []; off = 0
        for j in spec.joints:                      # minimal state per joint: [coordinates(nu); velocities(nu)], translational coordinates first
            if j.tra.limits is not None: th += list(x[off:off + j.tra.nu])
            if j.rot.limits is not None: th += list(x[off + j.tra.nu:off + j.nu])
            off += 2 * j.nu
        th = np.array(th)
        assert np.all(th >= lo - 1e-6) and np.all(th <= hi + 1e-6), (k, th)
        active += int(np.any((th < lo + 1e-5) | (th > hi - 1e-5)))
        if k % 30 == 29:
            data = o.get_data(); o.set_data(data); sol = o.get_solution()
            worst = max(worst, np.abs(fd_solution_matrix(o, data, sol) + o.full_matrix()).max())
    assert active > 20                             # the stops were reached and held
    assert worst < (1e-4 if kind == "mixed" else EPS), worst      # (mixed: a contact at the edge of its cone, where the central difference straddles the kink)
    if kind == "planar":
        for k in range(300):
            z, info = o.step(z, np.zeros(spec.nu))
        x = o.maximal_to_minimal(z)
        assert abs(x[1] - (-0.3)) < 1e-5 and abs(x[3]) < 1e-6, x       # the vertical coordinate rests on its lower stop (the horizontal one still creeps against its damper)

