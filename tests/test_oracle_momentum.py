"""The reference's momentum tests restated on the oracle (test/momentum.jl:1-381, src/mechanics/momentum.jl:17-53): without gravity and
contacts the total linear momentum and the total angular momentum about the origin -- taken from the Storage rows (save_to_storage!,
storage.jl:50-67: the momenta of every body with the joint impulses of the step folded in) -- stay constant to 1e-8 at
rtol = btol = 1e-12, through joint forces, springs, dampers and the controller on the joints with at most five inputs (U = 0.5 during
the first 100 steps).  Box, Pendulum, Atlas, Quadruped, Snake and Twister with the loops over the fifteen joint prototypes; the
Humanoid case needs a URDF this repository has not extracted.  (The reference's Snake loop compares stale arrays -- :286-287 test
`mlin0` / `mang0` of the previous rollout; here every rollout is checked.)"""
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle

EPS0 = 1.0e-12
DT0 = 1.0e-2
OPTS = d.SolverOptions(rtol=EPS0, btol=EPS0)
JOINT_TYPES = ["Fixed", "Prismatic", "Planar", "FixedOrientation", "Revolute", "Cylindrical", "PlanarAxis", "FreeRevolute", "Orbital",
               "PrismaticOrbital", "PlanarOrbital", "FreeOrbital", "Spherical", "CylindricalFree", "PlanarFree"]


def controls(spec, H, U):
    """controller!(mechanism, k; U) of test/momentum.jl:2-11: U on every input of every joint with at most five inputs, steps 1..100"""
    out = np.zeros((H, spec.nu))
    for j in spec.joints:
        sl = spec.input_slice(j.name)
        if 0 < sl.stop - sl.start <= 5:
            out[:100, sl] = U
    return out


def momenta(spec, rows):
    """momentum(mechanism, storage, t)  momentum.jl:55-75: [Σ p_linear; Σ (p_angular_world + r × m (v − v_com))] per recorded step, r from the
    centre of mass, v = p_linear / m"""
    m = np.array([b.mass for b in spec.bodies])[None, :, None]
    x, px, pq = rows[:, :, 0:3], rows[:, :, 13:16], rows[:, :, 16:19]
    P = px.sum(axis=1)
    com = (m * x).sum(axis=1) / m.sum()
    vcom = P / m.sum()
    L = (pq + np.cross(x - com[:, None, :], m * (px / m - vcom[:, None, :]))).sum(axis=1)
    return P, L


def rollout(spec, z0, tend, U):
    o = Oracle(spec, opts=OPTS)
    H = int(np.ceil(tend / spec.timestep))
    rows, status = o.simulate_storage(z0, controls(spec, H, U))
    assert all(s == 0 for s in status)
    return rows


def drift(spec, z0, tend, U, start):
    P, L = momenta(spec, rollout(spec, z0, tend, U)[start:])
    return np.abs(P - P[0]).max(), np.abs(L - L[0]).max()


def chain_state(spec, scale):
    v0 = scale * np.array([1, 2, 3.0]) * DT0
    return d.initialize(spec, base_position=np.zeros(3), base_rotation_vector=np.array([0.5 * np.pi, 0, 0]), base_linear_velocity=v0, base_angular_velocity=v0)


def test_box():
    """:45-68: one free body, v = (1, 2, 3), ω = (10, 10, 10), 5 s"""
    spec = d.get_mechanism("block", timestep=DT0, gravity=0.0, contact=False)
    z0 = d.initialize(spec, velocity=[1, 2, 3.0], angular_velocity=[10, 10, 10.0])
    dl, da = drift(spec, z0, 5.0, 0.0, 4)
    assert dl < 1.0e-8 and da < 1.0e-8, (dl, da)


def test_pendulum():
    """:79-102: the pendulum spinning at 5 rad/s about its pin: the angular momentum (the linear one is the pin's business)"""
    spec = d.get_mechanism("pendulum", timestep=DT0, gravity=0.0)
    z0 = d.initialize(spec, angle=0.7, angular_velocity=5.0)
    dl, da = drift(spec, z0, 5.0, 0.0, 9)
    assert da < 1.0e-8, da


def test_atlas():
    """:155-184: 31 bodies, springs 10 and dampers 1 on every joint, torques 0.5 on every actuated joint during the first 100 steps, 5 s"""
    spec = d.get_mechanism("atlas", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=10.0, dampers=1.0, contact_feet=False, contact_body=False)
    dl, da = drift(spec, d.initialize(spec), 5.0, 0.5, 0)
    assert dl < 1.0e-8 and da < 1.0e-8, (dl, da)


def test_quadruped():
    """:195-224: springs 0.3, dampers 0.1, controller, 5 s"""
    spec = d.get_mechanism("quadruped", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=0.3, dampers=0.1, contact_feet=False, contact_body=False)
    dl, da = drift(spec, d.initialize(spec), 5.0, 0.5, 0)
    assert dl < 1.0e-8 and da < 1.0e-8, (dl, da)


# Q8 (reference quirk, replicated): a Translational joint's input reaches the bodies' torques HALVED -- impulse_transform already carries the
# 0.5 of the attitude Jacobian (src/joints/impulses.jl:7, "#TODO: 0.5 Q") and input_impulse! divides by two once more
# (src/joints/translational/input.jl:20-22: `Jτ2 += Jτaa/2`), while the forces are applied in full.  The force pair is balanced, its torques
# are not: under control the angular momentum moves unless the levers are parallel to the force (Prismatic / Cylindrical snakes) or the
# joint has no translational input.  The reference's loop over the joint types does not notice (it re-tests the arrays of the Revolute
# rollout, test/momentum.jl:286-287).
ANGULAR_CONSERVED_UNDER_CONTROL = {"Fixed", "Prismatic", "Revolute", "Cylindrical", "Orbital", "Spherical"}


@pytest.mark.parametrize("joint_type,scale,tend,U", [("Revolute", 100.0, 1.5, 0.0)] + [(t, 10.0, 1.5, 0.5) for t in JOINT_TYPES])
def test_snake(joint_type, scale, tend, U):
    """:235-289: five links, springs 4, dampers 20, thrown and spinning; the Revolute snake without control, each of the fifteen joint
    prototypes with the controller.  Linear momentum: 1e-8 throughout.  Angular momentum: 1e-8 throughout where the inputs are torques or
    axis-aligned forces, and from the step after the controller stops for the others (Q8 above)."""
    spec = d.get_mechanism("snake", timestep=DT0, gravity=0.0, num_bodies=5, springs=4.0, dampers=20.0, joint_type=joint_type, contact=False, radius=0.05)
    P, L = momenta(spec, rollout(spec, chain_state(spec, scale), tend, U))
    assert np.abs(P[4:] - P[4]).max() < 1.0e-8
    assert np.abs(L[101:] - L[101]).max() < 1.0e-8
    if U == 0.0 or joint_type in ANGULAR_CONSERVED_UNDER_CONTROL:
        assert np.abs(L[4:] - L[4]).max() < 1.0e-8, joint_type
    else:
        assert np.abs(L[4:101] - L[4]).max() > 1.0e-4, joint_type          # (the quirk is there: ~1e-2 .. 1 over the controlled second)


@pytest.mark.parametrize("joint_type,scale,tend", [("FixedOrientation", 100.0, 2.5)] + [(t, 10.0, 1.5) for t in JOINT_TYPES])
def test_twister(joint_type, scale, tend):
    """:300-381: five links, joint axes cycling, springs 4, dampers 20, no control"""
    spec = d.get_mechanism("twister", timestep=DT0, gravity=0.0, num_bodies=5, springs=4.0, dampers=20.0, joint_type=joint_type, contact=False, radius=0.05)
    dl, da = drift(spec, chain_state(spec, scale), tend, 0.0, 4)
    assert dl < 1.0e-8 and da < 1.0e-8, (joint_type, dl, da)


# ---- the same invariants on the device (GPU tier): dojo_simulate's Storage rows ----
DEVICE_CASES = {
    "box": (lambda: d.get_mechanism("block", timestep=DT0, gravity=0.0, contact=False), lambda s: d.initialize(s, velocity=[1, 2, 3.0], angular_velocity=[10, 10, 10.0]), 2.0, 0.0),
    "quadruped": (lambda: d.get_mechanism("quadruped", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=0.3, dampers=0.1, contact_feet=False, contact_body=False), lambda s: d.initialize(s), 2.0, 0.5),
    "snake_Revolute": (lambda: d.get_mechanism("snake", timestep=DT0, gravity=0.0, num_bodies=5, springs=4.0, dampers=20.0, joint_type="Revolute", contact=False, radius=0.05), lambda s: chain_state(s, 100.0), 1.5, 0.0),
    "snake_Planar": (lambda: d.get_mechanism("snake", timestep=DT0, gravity=0.0, num_bodies=5, springs=4.0, dampers=20.0, joint_type="Planar", contact=False, radius=0.05), lambda s: chain_state(s, 10.0), 1.5, 0.5),
    "twister_Spherical": (lambda: d.get_mechanism("twister", timestep=DT0, gravity=0.0, num_bodies=5, springs=4.0, dampers=20.0, joint_type="Spherical", contact=False, radius=0.05), lambda s: chain_state(s, 10.0), 1.5, 0.0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(DEVICE_CASES))
def test_momentum_conservation_on_the_device(key):
    """test/momentum.jl on the HIP path: rollouts at rtol = btol = 1e-12 through dojo_simulate, momenta from the device's Storage rows:
    the reference's 1e-8 (Q8 included: the Planar snake's angular momentum moves under control on the device exactly as on the oracle)
    and the device's momenta equal to the oracle rollout's."""
    from dojo_amd import api
    build, init, tend, U = DEVICE_CASES[key]
    spec = build(); z0 = init(spec)
    H = int(np.ceil(tend / spec.timestep))
    Uh = controls(spec, H, U)
    B = 4
    gm = api.BatchedMechanism(spec, B, dtype="f64", opts=OPTS)
    Z, S, st = gm.simulate(np.tile(z0, (B, 1)), np.repeat(Uh[:, None, :], B, axis=1) if spec.nu else None, steps=H)
    gm.close()
    assert (st == 0).all()
    P, L = momenta(spec, S[:, 0])
    assert np.abs(P[4:] - P[4]).max() < 1.0e-8
    if key == "snake_Planar":
        assert np.abs(L[101:] - L[101]).max() < 1.0e-8 and np.abs(L[4:101] - L[4]).max() > 1.0e-4
    else:
        assert np.abs(L[4:] - L[4]).max() < 1.0e-8
    Po, Lo = momenta(spec, rollout(spec, z0, tend, U))
    assert np.abs(P - Po).max() < 1.0e-9 and np.abs(L - Lo).max() < 1.0e-9
    assert np.array_equal(S[:, 0], S[:, B - 1])
