"""The C replay of the Julia shim's drop-in (tests/c_driver/shim_driver.c): include/dojo_hip.h compiles as C, the PODs a C
caller fills are the ones the library reads, and the mehrotra!-seam call sequence of DojoHIP.jl gives the same step as
dojo_step(z, u) and as the oracle.  CPU tier: the driver builds (no GPU needed to compile the header as C); GPU tier: it runs."""
import ctypes as C
import os
import struct
import subprocess
import numpy as np
import pytest
import dojo_amd as d
from dojo_amd.topology import CBody, CJoint, CContact

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV_DIR = os.path.join(ROOT, "tests", "c_driver")
LIB = os.path.join(ROOT, "dojo.jl_amd", "csrc", "libdojo_hip.so")


def build_driver():
    exe = os.path.join(DRV_DIR, "shim_driver")
    src = os.path.join(DRV_DIR, "shim_driver.c")
    hdr = os.path.join(ROOT, "include", "dojo_hip.h")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe, src, "-ldl"])
    return exe


def test_header_is_c_and_driver_builds():
    exe = build_driver()
    assert os.path.exists(exe)
    # the PODs of the Python binding and of the C header have the same size (a layout drift would show here first)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [2, 3])
def test_shim_call_sequence_from_c(cfg, tmp_path):
    from dojo_amd import api
    from oracle import Oracle
    exe = build_driver()
    spec = d.baseline_config(cfg)
    opts = d.SolverOptions(rtol=1e-8, btol=1e-8)
    Z, U = d.synthetic_inputs(spec, 4)
    U = U + 0.3 * np.random.default_rng(3).standard_normal(U.shape)
    o = Oracle(spec, opts=opts)
    for _ in range(6):
        Z, st, it, _, _ = o.step_batch(Z, U, nthreads=4)
    z, u = Z[1], U[1]
    jf = o.input_impulses(z, u).reshape(-1)
    fext = 0.1 * np.random.default_rng(4).standard_normal(6 * spec.Nb)
    topo, keep = spec.to_ctypes()
    B_, J_, K_ = keep
    blob = struct.pack("4i", spec.Nb, len(spec.joints), len(spec.contacts), 0)
    blob += struct.pack("5d", spec.timestep, spec.input_scaling, *spec.gravity)
    blob += bytes(B_)[:C.sizeof(CBody) * spec.Nb] + bytes(J_)[:C.sizeof(CJoint) * len(spec.joints)] + bytes(K_)[:C.sizeof(CContact) * len(spec.contacts)]
    blob += bytes(opts.to_c()) + z.astype(np.float64).tobytes() + jf.astype(np.float64).tobytes() + fext.tobytes()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(blob)
    r = subprocess.run([exe, LIB, str(fin), str(fout)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = fout.read_bytes()
    status, iters, nji, nc, err_ok, szj = struct.unpack_from("6i", raw, 0)
    assert szj == C.sizeof(CJoint) and err_ok == 1 and nji == spec.n_joint_impulses and nc == len(spec.contacts)
    mu = struct.unpack_from("d", raw, 24)[0]
    vals = np.frombuffer(raw, dtype=np.float64, offset=32)
    nb = spec.Nb
    zn, vel = vals[:13 * nb], vals[13 * nb:19 * nb]
    ji = vals[19 * nb:19 * nb + max(nji, 1)][:nji]; cs = vals[19 * nb + max(nji, 1):][:8 * nc]
    # the same step through the Python binding with the controls as u
    gm = api.BatchedMechanism(spec, 1, dtype="f64", opts=opts)
    gm.set_external_force(fext.reshape(1, nb, 6))
    z2, st2, it2 = gm.step(z[None], u[None])
    v2, j2, c2 = gm.get_solution(); mu2 = gm.get_mu()
    gm.close()
    assert status == st2[0] and iters == it2[0]
    assert np.abs(zn - z2[0]).max() < 1e-9 and np.abs(vel - v2[0]).max() < 1e-9 and (nji == 0 or np.abs(ji - j2[0]).max() < 1e-8) and (nc == 0 or np.abs(cs - c2[0]).max() < 1e-8)
    assert mu > 0 and abs(mu - mu2[0]) <= 1e-6 * mu2[0]
    # ... and the oracle with the same external force (set_external_force! takes the force in the body frame, src/bodies/set.jl:110-115)
    from dojo_amd.quat import vrot, qconj
    o.set_state(z)
    for b in range(nb):
        o.set_external_force(b, force=vrot(fext[6 * b:6 * b + 3], qconj(z[13 * b + 6:13 * b + 10])), torque=fext[6 * b + 3:6 * b + 6])
    assert o.simulate_step(u, last=True) == status == 0
    assert np.abs(vel - o.velocity_solution()).max() < 1e-6
