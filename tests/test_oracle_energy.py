"""The reference's energy tests restated on the oracle (test/energy.jl:1-641, src/mechanics/energy.jl:10-92): mechanical energy of
rollouts at rtol = btol = 1e-12, taken from the Storage rows (momentum-derived velocities) from t = 1 s on, after the controllers
have stopped -- conserved to the reference's own bounds (exactly without springs, to O(timestep) without drift with them).
Dice, Pendulum, Slider 1-3, Atlas, Quadruped, Twister and the loop over the fifteen joint prototypes; the Humanoid case needs a URDF
this repository has not extracted.  The oracle's kinetic / potential energy follow energy.jl line by line (Mechanism::energy_of_storage_row)."""
import numpy as np
import pytest
import dojo_amd as d
from oracle import Oracle

EPS0 = 1.0e-12
DT0 = 1.0e-2
START0 = int(np.floor(1 / DT0))            # Julia's start0 = floor(1 / timestep) + 1, 1-based
OPTS = d.SolverOptions(rtol=EPS0, btol=EPS0)
JOINT_TYPES = ["Fixed", "Prismatic", "Planar", "FixedOrientation", "Revolute", "Cylindrical", "PlanarAxis", "FreeRevolute", "Orbital",
               "PrismaticOrbital", "PlanarOrbital", "FreeOrbital", "Spherical", "CylindricalFree", "PlanarFree"]


def controls(spec, H, U):
    """controller!(mechanism, k; U) of test/energy.jl:24-37: U on every input of every joint with at most five inputs, for the first second"""
    out = np.zeros((H, spec.nu))
    N = int(np.floor(1 / spec.timestep))
    for j in spec.joints:
        sl = spec.input_slice(j.name)
        if 0 < sl.stop - sl.start <= 5:
            out[:N, sl] = U
    return out


def energy_drift(spec, z0, tend, U):
    o = Oracle(spec, opts=OPTS)
    H = int(np.ceil(tend / spec.timestep))
    rows, status = o.simulate_storage(z0, controls(spec, H, U))
    assert all(s == 0 for s in status)
    ke, pe = o.energy(rows)
    me = (ke + pe)[START0:]
    return np.abs((me - me[0]) / np.mean(me)).max(), rows, ke, pe


def test_dice():
    """test/energy.jl:98-130: one free body, gravity, initial linear and angular velocity: 1e-8"""
    spec = d.get_mechanism("block", timestep=DT0, gravity=-10.0, contact=False)
    z0 = d.initialize(spec, velocity=[1, 2, 3.0], angular_velocity=[1, 1, 1.0])
    drift, *_ = energy_drift(spec, z0, 5.0, 0.0)
    assert drift < 1.0e-8, drift


def test_pendulum():
    """:142-177: pendulum with a joint spring, no gravity, torque 0.5 during the first second: 1e-2 (O(timestep), no drift)"""
    spec = d.get_mechanism("pendulum", timestep=DT0, gravity=0.0, springs=1.0, dampers=0.0)
    z0 = d.initialize(spec, angle=0.5 * np.pi, angular_velocity=0.0)
    drift, *_ = energy_drift(spec, z0, 25.0, 0.5)
    assert drift < 1.0e-2, drift


def test_slider_1():
    """:188-232: mass on a spring, no gravity: amplitude and peak velocity of the analytic oscillator to 1e-4, energy to 1e-3"""
    k = 10.0
    spec = d.get_mechanism("slider", timestep=DT0, gravity=0.0, springs=k, dampers=0.0)
    z0 = d.initialize(spec, position=0.5)
    drift, rows, ke, pe = energy_drift(spec, z0, 5.0, 0.0)
    zmax = 0.5; vmax = 0.5 * np.sqrt(k / spec.bodies[0].mass)
    assert abs(rows[:, 0, 2].max() - zmax + 0.5) < 1.0e-4          # storage.x[1][t][3]
    assert abs(rows[:, 0, 21].max() - vmax) < 1.0e-4               # storage.vl[1][t][3]
    assert drift < 1.0e-3, drift


def test_slider_2():
    """:243-276: free fall along the slider, no spring: 1e-6"""
    spec = d.get_mechanism("slider", timestep=DT0, gravity=-9.81, springs=0.0, dampers=0.0)
    drift, *_ = energy_drift(spec, d.initialize(spec, position=0.5), 1.5, 0.0)
    assert drift < 1.0e-6, drift


def test_slider_3():
    """:287-318: gravity and spring: 1e-3"""
    spec = d.get_mechanism("slider", timestep=DT0, gravity=-9.81, springs=1.0, dampers=0.0)
    drift, *_ = energy_drift(spec, d.initialize(spec, position=0.1), 10.0, 0.0)
    assert drift < 1.0e-3, drift


def test_quadruped():
    """:427-459: thirteen bodies on joint springs away from their offsets, no gravity, no contact, no limits, no control: 1e-2"""
    spec = d.get_mechanism("quadruped", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=1.0, contact_feet=False, contact_body=False, joint_limits={})
    drift, *_ = energy_drift(spec, d.initialize(spec), 5.0, 0.0)
    assert drift < 1.0e-2, drift


def test_atlas():
    """:382-416: 31 bodies, springs on every joint, random body angular velocities (set_maximal_velocities!, inconsistent with the joints
    like the reference's), torques 0.05 during the first second: 3e-3"""
    spec = d.get_mechanism("atlas", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=1.0, contact_feet=False, contact_body=False)
    z0 = d.initialize(spec).reshape(spec.Nb, 13)
    z0[:, 10:13] = np.random.default_rng(7).random((spec.Nb, 3))
    drift, *_ = energy_drift(spec, z0.reshape(-1), 5.0, 0.05)
    assert drift < 3.0e-3, drift


@pytest.mark.parametrize("joint_type,bound", [("Revolute", 1.0e-3)] + [(t, 1.0e-2) for t in JOINT_TYPES])
def test_twister(joint_type, bound):
    """:562-640: five links, joint axes cycling, weak springs, thrown and spinning, torques 0.01 / 0.05 during the first second: 1e-3 for
    the Revolute twister, 1e-2 for each of the fifteen joint prototypes"""
    spec = d.get_mechanism("twister", timestep=DT0, gravity=0.0, num_bodies=5, springs=0.01, dampers=0.0, joint_type=joint_type, contact=False, radius=0.05)
    v0 = 10.0 * np.array([1, 2, 3.0]) * DT0
    z0 = d.initialize(spec, base_position=np.zeros(3), base_rotation_vector=np.array([0.5 * np.pi, 0, 0]), base_linear_velocity=v0, base_angular_velocity=v0)
    drift, *_ = energy_drift(spec, z0, 3.0, 0.01 if bound == 1.0e-3 else 0.05)
    assert drift < bound, (joint_type, drift)


# ---- the same invariant on the device (GPU tier): dojo_simulate's Storage rows, energy by the oracle's restatement of energy.jl ----
DEVICE_CASES = {
    "pendulum": (lambda: d.get_mechanism("pendulum", timestep=DT0, gravity=0.0, springs=1.0, dampers=0.0), dict(angle=0.5 * np.pi, angular_velocity=0.0), 6.0, 0.5, 1.0e-2),
    "slider": (lambda: d.get_mechanism("slider", timestep=DT0, gravity=-9.81, springs=1.0, dampers=0.0), dict(position=0.1), 6.0, 0.0, 1.0e-3),
    "quadruped": (lambda: d.get_mechanism("quadruped", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=1.0, contact_feet=False, contact_body=False, joint_limits={}), dict(), 3.0, 0.0, 1.0e-2),
    "atlas": (lambda: d.get_mechanism("atlas", timestep=DT0, gravity=0.0, parse_springs=False, parse_dampers=False, springs=1.0, contact_feet=False, contact_body=False), dict(), 3.0, 0.05, 3.0e-3),
    "twister_Revolute": (lambda: d.get_mechanism("twister", timestep=DT0, gravity=0.0, num_bodies=5, springs=0.01, dampers=0.0, joint_type="Revolute", contact=False, radius=0.05), None, 3.0, 0.01, 1.0e-3),
    "twister_Spherical": (lambda: d.get_mechanism("twister", timestep=DT0, gravity=0.0, num_bodies=5, springs=0.01, dampers=0.0, joint_type="Spherical", contact=False, radius=0.05), None, 3.0, 0.05, 1.0e-2),
    "twister_Prismatic": (lambda: d.get_mechanism("twister", timestep=DT0, gravity=0.0, num_bodies=5, springs=0.01, dampers=0.0, joint_type="Prismatic", contact=False, radius=0.05), None, 3.0, 0.05, 1.0e-2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(DEVICE_CASES))
def test_energy_conservation_on_the_device(key):
    """test/energy.jl on the HIP path: rollouts at rtol = btol = 1e-12 through dojo_simulate (device Storage rows), the reference's
    bounds on the drift of the mechanical energy after the controllers stop, and the device's energies equal to the oracle's rollout."""
    from dojo_amd import api
    build, init, tend, U, bound = DEVICE_CASES[key]
    spec = build()
    if init is None:
        v0 = 10.0 * np.array([1, 2, 3.0]) * DT0
        z0 = d.initialize(spec, base_position=np.zeros(3), base_rotation_vector=np.array([0.5 * np.pi, 0, 0]), base_linear_velocity=v0, base_angular_velocity=v0)
    else:
        z0 = d.initialize(spec, **init)
    if key == "atlas":
        z0 = z0.reshape(spec.Nb, 13); z0[:, 10:13] = np.random.default_rng(7).random((spec.Nb, 3)); z0 = z0.reshape(-1)
    H = int(np.ceil(tend / spec.timestep))
    Uh = controls(spec, H, U)
    B = 4
    gm = api.BatchedMechanism(spec, B, dtype="f64", opts=OPTS)
    Z, S, st = gm.simulate(np.tile(z0, (B, 1)), np.repeat(Uh[:, None, :], B, axis=1) if spec.nu else None, steps=H)
    gm.close()
    assert (st == 0).all()
    o = Oracle(spec, opts=OPTS)
    ke, pe = o.energy(S[:, 0])
    me = (ke + pe)[START0:]
    assert np.abs((me - me[0]) / np.mean(me)).max() < bound
    rows, status = o.simulate_storage(z0, Uh)
    ke_o, pe_o = o.energy(rows)
    assert np.abs((ke + pe) - (ke_o + pe_o)).max() <= 1e-6 * max(1.0, np.abs(ke_o + pe_o).max())
    assert np.array_equal(S[:, 0], S[:, B - 1])
