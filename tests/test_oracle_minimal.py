"""test/minimal.jl restated: the minimal <-> maximal coordinate maps of the C++ oracle
(src/mechanism/state.jl:9-66, src/joints/minimal.jl:134-196) survive the reference's own round-trip tests over all fifteen
joint prototypes and the mechanisms it lists, and the numpy restatement the host package ships (dojo_amd/coords.py, used to
build the synthetic inputs) agrees with the oracle -- so the device maps are checked against a pinned oracle, not against
a module of the product (tests/test_gpu_parity.py::test_minimal_maximal_maps)."""
import numpy as np
import pytest
import dojo_amd as d
from dojo_amd import coords
from oracle import Oracle

JOINT_TYPES = ["Fixed", "Prismatic", "Planar", "FixedOrientation", "Revolute", "Cylindrical", "PlanarAxis", "FreeRevolute", "Orbital",
               "PrismaticOrbital", "PlanarOrbital", "FreeOrbital", "Spherical", "CylindricalFree", "PlanarFree"]       # test/minimal.jl:5-21


def _round_trip(spec, seed=100):
    o = Oracle(spec)
    x0 = np.random.default_rng(seed).random(2 * spec.nu)           # Random.seed!(100); x0 = rand(nx)
    z0 = o.minimal_to_maximal(x0)
    x1 = o.maximal_to_minimal(z0)
    # unit quaternions, and the numpy restatement of the host package computes the same maps
    q = z0.reshape(spec.Nb, 13)[:, 6:10]
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-12
    assert np.abs(coords.minimal_to_maximal(spec, x0) - z0).max() < 1e-10
    assert np.abs(coords.maximal_to_minimal(spec, z0) - x1).max() < 1e-10
    return x0, x1, z0


@pytest.mark.parametrize("name,kw", [("block", {}), ("pendulum", {}), ("nslider", dict(num_bodies=5)), ("quadruped", {}), ("atlas", {}), ("ant", {}),
                                     ("raiberthopper", {}), ("cartpole", {}), ("slider", {}), ("sphere", {}), ("dzhanibekov", {}), ("tippetop", {})])
def test_minimal_to_maximal_to_minimal_mechanisms(name, kw):
    """test/minimal.jl:113-262 "Minimal to maximal to minimal" (the mechanisms of the list that the host builders restate)"""
    x0, x1, _ = _round_trip(d.get_mechanism(name, **kw))
    assert np.abs(x0 - x1).max() < 1e-8


@pytest.mark.parametrize("joint_type", JOINT_TYPES)
@pytest.mark.parametrize("name,kwname", [("npendulum", "rest_joint_type"), ("snake", "joint_type"), ("twister", "joint_type")])
def test_minimal_to_maximal_to_minimal_joint_types(name, kwname, joint_type):
    """test/minimal.jl:176-231: npendulum / snake / twister with five bodies for every joint prototype"""
    x0, x1, _ = _round_trip(d.get_mechanism(name, num_bodies=5, **{kwname: joint_type}))
    assert np.abs(x0 - x1).max() < 1e-8


@pytest.mark.parametrize("joint_type", JOINT_TYPES)
def test_get_and_set_minimal_coordinates_and_velocities(joint_type):
    """test/minimal.jl:65-106 "Minimal coordinates": along a ten-body snake with random orientation offsets, setting one joint's
    minimal coordinates / velocities and reading them back gives the same numbers, joint by joint down the chain."""
    spec = d.get_mechanism("snake", num_bodies=10, joint_type=joint_type)
    rng = np.random.default_rng(100)
    for j in spec.joints:
        q = rng.standard_normal(4); j.orientation_offset = q / np.linalg.norm(q)
    o = Oracle(spec)
    x = np.zeros(2 * spec.nu); off = 0
    vals = rng.random(2 * spec.joints[1].nu) if len(spec.joints) > 1 else np.zeros(0)
    for j in spec.joints:
        n = j.nu
        if j.parent >= 0:
            x[off:off + 2 * n] = vals[:2 * n]
        off += 2 * n
    z = o.minimal_to_maximal(x)
    assert np.abs(o.maximal_to_minimal(z) - x).max() < 1e-8


def test_maximal_to_minimal_of_the_synthetic_inputs_is_consistent():
    """the synthetic inputs of the parity tests are built from minimal coordinates: joints closed"""
    for cfg in (1, 2, 3, 4, 5):
        spec = d.baseline_config(cfg)
        Z, U = d.synthetic_inputs(spec, 4)
        o = Oracle(spec)
        for b in range(4):
            x = o.maximal_to_minimal(Z[b])
            assert np.abs(o.minimal_to_maximal(x) - Z[b]).max() < 1e-8
