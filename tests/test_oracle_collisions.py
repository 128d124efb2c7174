"""The reference's sphere-sphere collision tests restated on the oracle (test/collisions.jl:1-575, "Collision: Sphere-sphere";
src/contacts/collisions/{collision,sphere_sphere}.jl, src/contacts/{contact,velocity}.jl for a contact between two bodies).
The collision's Jacobians against finite differences of the functions they differentiate (`test_jacobians`, :59-170), the geometry of
the two-sphere mechanism, and the rollouts: the free sphere comes to rest on the fixed one, bounces off without gravity, pushes a
floating one away.  For all three friction types of the reference's loop (:nonlinear, :linear, :impact)."""
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle


def _fd(f, x, h=1e-6):
    x = np.asarray(x, float); f0 = np.atleast_1d(f(x))
    J = np.zeros((f0.size, x.size))
    for i in range(x.size):
        e = np.zeros_like(x); e[i] = h
        J[:, i] = (np.atleast_1d(f(x + e)) - np.atleast_1d(f(x - e))) / (2 * h)
    return J


def _state(spec, x1, x2, v2, q1=None, q2=None):
    z = np.zeros((2, 13)); z[:, 6] = 1.0
    z[0, 0:3] = x1; z[1, 0:3] = x2; z[1, 3:6] = v2
    if q1 is not None: z[0, 6:10] = q1
    if q2 is not None: z[1, 6:10] = q2
    return z.reshape(-1)


def check_jacobians(o, xp, qp, xc, qc):
    """test_jacobians(mechanism)  test/collisions.jl:59-170: every ∂ of the collision against the derivative of the function itself"""
    dis = o.contact_unit(0, 0, xp, qp, xc, qc)[0]
    sgn = 1.0 if dis >= 0 else -1.0
    for jp in (True, False):
        k = 0 if jp else 1
        def wrt_x(what):
            return (lambda x: o.contact_unit(0, what, x, qp, xc, qc)) if jp else (lambda x: o.contact_unit(0, what, xp, qp, x, qc))
        def wrt_q(what):
            return (lambda q: o.contact_unit(0, what, xp, q, xc, qc)) if jp else (lambda q: o.contact_unit(0, what, xp, qp, xc, q))
        x0, q0 = (xp, qp) if jp else (xc, qc)
        assert np.abs(sgn * 0 + o.contact_unit(0, 22 + k, xp, qp, xc, qc).reshape(3, 3) - _fd(wrt_x(3), x0)).max() < 1e-6           # ∂normal∂x  :66-75 (the sign is inside, collision.jl:62-67)
        assert np.abs(o.contact_unit(0, 24 + k, xp, qp, xc, qc).reshape(3, 4) - _fd(wrt_q(3), q0)).max() < 1e-6                    # ∂normal∂q  :77-84
        T1x = o.contact_unit(0, 26 + k, xp, qp, xc, qc).reshape(3, 3); T2x = o.contact_unit(0, 28 + k, xp, qp, xc, qc).reshape(3, 3)
        assert np.abs(T1x - _fd(lambda x: wrt_x(4)(x)[:3], x0)).max() < 1e-6 and np.abs(T2x - _fd(lambda x: wrt_x(4)(x)[3:], x0)).max() < 1e-6   # :86-104
        T2q = o.contact_unit(0, 32 + k, xp, qp, xc, qc).reshape(3, 4)
        assert np.abs(T2q - _fd(lambda q: wrt_q(4)(q)[3:], q0)).max() < 1e-6                                                          # :116-124
        # ∂tangent_one∂q: the reference multiplies by skew(t1) where skew(w) belongs (collision.jl:207); both vanish for spheres about the centres of mass
        assert np.abs(o.contact_unit(0, 30 + k, xp, qp, xc, qc)).max() < 1e-12 and np.abs(_fd(lambda q: wrt_q(4)(q)[:3], q0)).max() < 1e-6
        assert np.abs(o.contact_unit(0, 10 + k, xp, qp, xc, qc).reshape(1, 3) - _fd(wrt_x(0), x0)).max() < 1e-5                       # ∂distance∂x  :126-134
        assert np.abs(o.contact_unit(0, 12 + k, xp, qp, xc, qc).reshape(1, 4) - _fd(wrt_q(0), q0)).max() < 1e-5                       # ∂distance∂q  :136-144
        for rel_parent in (True, False):
            w = (14 if rel_parent else 16) + k
            assert np.abs(o.contact_unit(0, w, xp, qp, xc, qc).reshape(3, 3) - _fd(wrt_x(1 if rel_parent else 2), x0)).max() < 1e-6  # ∂contact_point∂x  :146-156
            assert np.abs(o.contact_unit(0, w + 4, xp, qp, xc, qc).reshape(3, 4) - _fd(wrt_q(1 if rel_parent else 2), q0)).max() < 1e-6


Q1 = np.array([1.0, 0, 0, 0])


@pytest.mark.parametrize("friction_type", ["nonlinear", "linear", "impact"])
def test_geometry_and_jacobians(friction_type):
    """:172-216: distance 1, contact points (0, 0, 0.5) and (0, 0, 1.5), normal (0, 0, -1) (child -> parent), tangents e_y and -e_x; the
    Jacobians there, at a generic pair of poses and in penetration"""
    spec = d.get_two_spheres(friction_type=friction_type)
    o = Oracle(spec)
    xp, xc = np.zeros(3), np.array([0.0, 0, 2.0])
    assert abs(o.contact_unit(0, 0, xp, Q1, xc, Q1)[0] - 1.0) < 1e-6
    assert np.abs(o.contact_unit(0, 1, xp, Q1, xc, Q1) - [0, 0, 0.5]).max() < 1e-6 and np.abs(o.contact_unit(0, 2, xp, Q1, xc, Q1) - [0, 0, 1.5]).max() < 1e-6
    assert np.abs(o.contact_unit(0, 3, xp, Q1, xc, Q1) - [0, 0, -1.0]).max() < 1e-6
    assert np.abs(o.contact_unit(0, 4, xp, Q1, xc, Q1) - [0, 1.0, 0, -1.0, 0, 0]).max() < 1e-6
    check_jacobians(o, xp, Q1, xc, Q1)
    rng = np.random.default_rng(5)
    qa, qb = rng.normal(size=4), rng.normal(size=4); qa /= np.linalg.norm(qa); qb /= np.linalg.norm(qb)
    check_jacobians(o, np.array([0.3, -0.2, 0.1]), qa, np.array([0.9, 0.7, 1.4]), qb)
    check_jacobians(o, np.array([0.0, 0.0, 0.0]), qa, np.array([0.3, 0.2, 0.7]), qb)              # penetrating: the normal keeps pointing child -> parent


def rollout(spec, z0, steps, opts=None):
    o = Oracle(spec, opts=opts or d.SolverOptions())
    z = z0.copy(); Z = [z]
    for _ in range(steps):
        z, info = o.step(z, np.zeros(spec.nu))
        assert info["status"] == 0
        Z.append(z)
    return np.array(Z).reshape(len(Z), 2, 13), o


@pytest.mark.parametrize("friction_type", ["nonlinear", "linear", "impact"])
def test_rollouts(friction_type):
    """:218-330: 2 s at timestep 0.1.  Under gravity the free sphere ends resting on the fixed one (z = 1 to 1e-4); without gravity, thrown
    at it with 5 m/s, it does not pass (z > 1: the contact is inelastic); a floating first sphere is pushed away (both move down,
    more than one diameter apart); the same along x."""
    spec = d.get_two_spheres(friction_type=friction_type, gravity=-9.81)
    Z, o = rollout(spec, _state(spec, [0, 0, 0], [0, 0, 2.0], [0, 0, 0]), 20)
    assert np.abs(Z[-1, 1, 0:3] - [0, 0, 1.0]).max() < 1e-4
    check_jacobians(o, Z[-1, 0, 0:3], Z[-1, 0, 6:10], Z[-1, 1, 0:3], Z[-1, 1, 6:10])
    spec = d.get_two_spheres(friction_type=friction_type, gravity=0.0)
    Z, o = rollout(spec, _state(spec, [0, 0, 0], [0, 0, 2.0], [0, 0, -5.0]), 20)
    assert Z[-1, 1, 2] > 1.0
    spec = d.get_two_spheres(friction_type=friction_type, gravity=0.0, joint_world_body1="Floating")
    Z, o = rollout(spec, _state(spec, [0, 0, 0], [0, 0, 2.0], [0, 0, -5.0]), 20)
    assert Z[-1, 1, 2] - Z[-1, 0, 2] > 1.0 and Z[-1, 1, 2] < 0.0
    # the momentum of the pair is what the free sphere brought (equal masses), and the spheres do not approach any more
    assert abs(Z[-1, :, 5].sum() - (-5.0)) < 1e-6 and Z[-1, 1, 5] - Z[-1, 0, 5] > -1e-6
    spec = d.get_two_spheres(friction_type=friction_type, gravity=0.0)
    z0 = _state(spec, [0, 0, 0], [2.0, 0, 0], [-5.0, 0, 0])
    o = Oracle(spec)
    assert abs(o.contact_unit(0, 0, z0[0:3], Q1, z0[13:16], Q1)[0] - 1.0) < 1e-6
    Z, o = rollout(spec, z0, 20)
    assert Z[-1, 1, 0] > 1.0


def test_friction_between_the_spheres():
    """beyond the reference's cases: the free sphere lands on the fixed one slightly off the pole with spin; NonlinearContact friction
    (mu = 0.5) acts through the contact point on both bodies, and the fixed sphere's joint takes the load: the solver converges on every step
    and the contact impulse stays inside its cone"""
    spec = d.get_two_spheres(friction_type="nonlinear", gravity=-9.81, timestep=0.02)
    z = _state(spec, [0, 0, 0], [0.05, 0.0, 1.2], [0.3, 0, 0])
    z[13 + 10:13 + 13] = [0.0, 2.0, 0.0]
    o = Oracle(spec)
    for _ in range(40):
        z, info = o.step(z, np.zeros(spec.nu)); assert info["status"] == 0
        sol = o.get_solution()
        g = sol[-4:]
        assert g[0] >= -1e-9 and np.hypot(g[2], g[3]) <= g[1] + 1e-6 and abs(g[1] - 0.5 * g[0]) < 1e-4 + 1e-3 * abs(g[0]) or g[0] < 1e-6
        assert np.linalg.norm(z[13:16] - z[0:3]) > 1.0 - 1e-4


# ---- the HIP path (GPU tier): the same mechanism through the C-ABI ----
@pytest.mark.gpu
@pytest.mark.parametrize("friction_type,dtype,free_on", [("nonlinear", "f64", "body1"), ("impact", "f64", "body1"), ("linear", "f64", "body1"), ("nonlinear", "f32", "body1"),
                                                         ("nonlinear", "f64", "world"), ("impact", "f64", "world"), ("nonlinear", "f32", "world")])
def test_body_body_contact_on_the_device(friction_type, dtype, free_on):
    """(free_on = "world": the second sphere is a free body as in the reference's get_two_body -- the contact is no tree edge but a cut element of the
    general lane-mapping builds, round 5; "body1": it hangs in the tree on the first sphere, the quad builds)
    the two-sphere mechanism on the GPU: a batch of 256 environments whose free sphere starts at random places above / beside the other
    one with random velocities and spins, stepped 25 times next to the oracle: every environment-step both sides solve along the same Newton path ends within 1e-6 of the oracle's state (a bound);
    solves that take different paths are counted and bounded (see below) (fp64 ABI; the fp32 ABI, whose states are rounded between the steps); then the reference's resting case through dojo_simulate"""
    from dojo_amd import api
    B = 256
    rng = np.random.default_rng(17)
    for joint in ("Fixed", "Floating"):
        spec = d.get_two_spheres(friction_type=friction_type, gravity=-9.81, joint_world_body1=joint, free_on=free_on)
        Z = np.zeros((B, 2, 13)); Z[:, :, 6] = 1.0
        dirs = rng.normal(size=(B, 3)); dirs[:, 2] = np.abs(dirs[:, 2]) + 0.3; dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        Z[:, 1, 0:3] = dirs * rng.uniform(1.05, 1.6, size=(B, 1))
        Z[:, 1, 3:6] = -dirs * rng.uniform(0.0, 3.0, size=(B, 1)) + 0.3 * rng.normal(size=(B, 3))
        Z[:, 1, 10:13] = rng.normal(size=(B, 3))
        Z = Z.reshape(B, -1)
        gm = api.BatchedMechanism(spec, B, dtype=dtype)
        o = Oracle(spec)
        z = Z.astype(np.float32).astype(np.float64) if dtype == "f32" else Z.copy()
        contact_seen = 0; n_apart = 0; n_stat = 0
        bound = 1e-6 if dtype == "f64" else 1e-4                  # (fp32 ABI: the states are rounded to fp32 between the steps)
        for k in range(25):
            zg, st, it = gm.step(z, np.zeros((B, spec.nu)))
            zin = d.fp32_abi_state(z) if dtype == "f32" else z
            Zo, st_o, it_o = o.step_batch(zin, np.zeros((B, spec.nu)), nthreads=8)[:3]
            # Every environment-step that both sides solve along the same Newton path (equal iteration counts) ends within `bound` of the
            # oracle's state -- a BOUND, on a host-independent oracle (oracle/Makefile: no FMA contraction) -- unless the solve is a long one
            # (> 20 iterations: DESIGN.md section 7), and then within 1e-3.  Where the two sides take DIFFERENT paths the end points differ by up
            # to ~1e-4: with a body-body contact the Newton matrix is inexact by construction (contact.jl:37-77 leaves dvt/dx out, the
            # reference returns FiniteDiff Jacobians), the iteration converges linearly, and the iterate at which rvio crosses rtol depends on
            # the last bits of the linear solves.  Those environment-steps are counted and bounded (<= 1 % of them, <= 1e-3), not compared
            # at 1e-6: body-body parity is a same-path statement (row f4 stays "partial").
            both = (st == 0) & (st_o == 0)
            e = np.abs(zg[both].astype(np.float64) - Zo[both]).max(axis=1)
            itb, itob = it[both], it_o[both]
            path = (itb == itob) if dtype == "f64" else (np.abs(itb - itob) <= 2)
            long_ = (itb > 20) | (itob > 20)
            reg = path & ~long_
            assert not reg.any() or e[reg].max() <= bound, (joint, k, e[reg].max())
            assert e.max() < 1e-3, (joint, k, e.max())
            n_apart += int((~path).sum()); n_stat += int((st != st_o).sum())
            contact_seen += int((np.linalg.norm(Zo[:, 13:16] - Zo[:, 0:3], axis=1) < 1.0 + 1e-3).sum())
            z = zg.astype(np.float64)
        print("%s %s %s: solves along different Newton paths %d, status mismatches %d of %d environment-steps" % (friction_type, dtype, joint, n_apart, n_stat, 25 * B))
        assert contact_seen > B and n_stat <= 0.01 * 25 * B and n_apart <= 0.01 * 25 * B, (n_apart, n_stat)
        with pytest.raises(Exception):
            gm.step(z, np.zeros((B, spec.nu)), with_gradient=True)          # forward only, like the reference's data Jacobians
        gm.close()
    spec = d.get_two_spheres(friction_type=friction_type, gravity=-9.81, free_on=free_on)
    gm = api.BatchedMechanism(spec, 4, dtype=dtype)
    z0 = _state(spec, [0, 0, 0], [0, 0, 2.0], [0, 0, 0])
    Zs, S, st = gm.simulate(np.tile(z0, (4, 1)), np.zeros((20, 4, spec.nu)), steps=20)
    gm.close()
    assert (st == 0).all()
    zend = Zs[-1, 0]
    rows, st_o = Oracle(spec).simulate_storage(z0, np.zeros((20, spec.nu)))            # the Storage rows (momenta with the contact impulse on both spheres)
    assert np.abs(S[:, 0].astype(np.float64) - rows).max() < (1e-6 if dtype == "f64" else 1e-4)
    assert np.abs(zend.astype(np.float64).reshape(2, 13)[1, 0:3] - [0, 0, 1.0]).max() < (1e-4 if dtype == "f64" else 2e-4)      # test/collisions.jl:226


@pytest.mark.gpu
@pytest.mark.parametrize("friction_type", ["nonlinear", "linear"])
def test_body_body_contact_off_the_centres_of_mass_on_the_device(friction_type):
    """spheres off the centres of mass (origin_parent, origin_child != 0) on the GPU: 64 perturbed copies of the approach of
    tests/test_device_program_emu.py::off_centre_pair, 20 steps next to the oracle: equal iteration counts, states to 1e-6"""
    from dojo_amd import api
    from test_device_program_emu import off_centre_pair
    for joint in ("Floating", "Revolute"):
        spec, z0 = off_centre_pair(friction_type, joint)
        B = 64
        rng = np.random.default_rng(5)
        Z = np.tile(z0, (B, 1)); Z[:, 16:19] += 0.2 * rng.normal(size=(B, 3)); Z[:, 23:26] += 0.5 * rng.normal(size=(B, 3))
        gm = api.BatchedMechanism(spec, B, dtype="f64"); o = Oracle(spec)
        z = Z.copy(); n_apart = 0
        for k in range(20):
            zg, st, it = gm.step(z, np.zeros((B, spec.nu)))
            Zo, st_o, it_o = o.step_batch(z, np.zeros((B, spec.nu)), nthreads=8)[:3]
            same = (st == 0) & (st_o == 0) & (it == it_o)
            n_apart += int(((st != st_o) | ((st == 0) & (st_o == 0) & (it != it_o))).sum())      # (solves that run into max_iter on both sides are not "apart")
            assert np.abs(zg[same] - Zo[same]).max() < 1e-6, (joint, k, np.abs(zg[same] - Zo[same]).max())
            z = zg
        gm.close()
        assert n_apart <= 0.02 * 20 * B, n_apart


@pytest.mark.gpu
def test_body_body_contact_between_the_ends_of_a_chain_on_the_device():
    """a cut contact inside one tree on the GPU: 64 perturbed copies of tests/test_device_program_emu.py::folded_chain (the first and the last link of
    a folded three-link pendulum touch), 25 steps next to the oracle: states to 1e-6 wherever both sides take the same Newton path, at most 2 % of
    the environment-steps apart (the impact step runs into max_iter on both sides for part of the copies)"""
    from dojo_amd import api
    from test_device_program_emu import folded_chain
    B = 64
    rng = np.random.default_rng(11)
    spec = folded_chain()[0]
    z = np.stack([folded_chain(spread=rng)[1] for _ in range(B)])
    gm = api.BatchedMechanism(spec, B, dtype="f64"); o = Oracle(spec)
    n_apart = 0; touched = 0
    for k in range(25):
        zg, st, it = gm.step(z, np.zeros((B, spec.nu)))
        Zo, st_o, it_o = o.step_batch(z, np.zeros((B, spec.nu)), nthreads=8)[:3]
        same = (st == 0) & (st_o == 0) & (it == it_o)
        n_apart += int(((st != st_o) | ((st == 0) & (st_o == 0) & (it != it_o))).sum())
        assert np.abs(zg[same] - Zo[same]).max() < 1e-6, (k, np.abs(zg[same] - Zo[same]).max())
        touched += int((np.linalg.norm(Zo[:, 0:3] - Zo[:, 26:29], axis=1) < 0.4 + 1e-3).sum())
        z = Zo
    gm.close()
    assert touched > B and n_apart <= 0.02 * 25 * B, (touched, n_apart)


@pytest.mark.gpu
def test_ball_on_atlas_on_the_device():
    """a body-body contact in a mechanism of 32 bodies with four contacts per foot (tests/test_device_program_emu.py::ball_on_atlas): the tree-edge
    builds do not serve it, the contact travels as a cut element of the general lane-mapping builds (refused until round 5).  32 perturbed copies,
    six steps next to the oracle: states to 1e-6 wherever both sides take the same Newton path, at most 5 % of the environment-steps apart"""
    from dojo_amd import api
    from test_device_program_emu import ball_on_atlas
    spec, z0, u0 = ball_on_atlas()
    B = 32
    rng = np.random.default_rng(2)
    Z = np.tile(z0, (B, 1)); Z[:, -13:-10] += 0.01 * rng.normal(size=(B, 3)); Z[:, -10:-7] += 0.2 * rng.normal(size=(B, 3))
    U = np.tile(u0, (B, 1))
    gm = api.BatchedMechanism(spec, B, dtype="f64"); o = Oracle(spec)
    z = Z.copy(); n_apart = 0; loaded = 0
    for k in range(6):
        zg, st, it = gm.step(z, U)
        vel, ji, cs = gm.get_solution()
        Zo, st_o, it_o = o.step_batch(z, U, nthreads=8)[:3]
        same = (st == 0) & (st_o == 0) & (it == it_o)
        n_apart += int(((st != st_o) | ((st == 0) & (st_o == 0) & (it != it_o))).sum())
        assert same.any() and np.abs(zg[same] - Zo[same]).max() < 1e-6, (k, np.abs(zg[same] - Zo[same]).max())
        loaded += int((cs[:, -4] > 1e-2).sum())
        z = Zo
    gm.close()
    assert loaded > B and n_apart <= 0.05 * 6 * B, (loaded, n_apart)
