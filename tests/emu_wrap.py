"""ctypes wrapper of the SIMT emulator (tests/emu/libemu.so): runs the shipped device source
(dojo.jl_amd/csrc/dojo_device.hpp) on CPU threads.  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np
from dojo_amd.topology import CTopology, CSolverOptions, SolverOptions

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_lib_lin = None


def _build(so, src, flags=()):
    """compile once, whoever comes first (pytest-xdist workers race here): under a file lock, into a temporary name, renamed into place"""
    import fcntl
    with open(so + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so) or any(os.path.getmtime(s_) > os.path.getmtime(so) for s_ in src):
            tmp = "%s.%d.tmp" % (so, os.getpid())
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", *flags, "-o", tmp, src[0]])
            os.replace(tmp, so)


def lib_linear():
    """the emulator compiled with -DDJ_LINEAR=1 (LinearContact builds of the device source: six cone pairs per contact)"""
    global _lib_lin
    if _lib_lin is None:
        so = os.path.join(_HERE, "emu", "libemu_lin.so")
        src = [os.path.join(_HERE, "emu", "emu.cpp")] + [os.path.join(_HERE, "..", "dojo.jl_amd", "csrc", f)
                                                         for f in ("dojo_device.hpp", "dojo_host.hpp", "dojo_math.hpp")]
        _build(so, src, ("-DDJ_LINEAR=1", "-DDJ_TSD=0"))      # (the flag set of the GPU's LinearContact builds)
        _lib_lin = C.CDLL(so)
        _lib_lin.emu_step.restype = C.c_int
    return _lib_lin


_lib_mlim = None


def lib_mlim():
    """the emulator compiled with -DDJ_MLIM=1 -DDJ_CUT=1 (the product's general lane-mapping builds): joint limits on several coordinates / both
    halves, kinematic loops"""
    global _lib_mlim
    if _lib_mlim is None:
        so = os.path.join(_HERE, "emu", "libemu_mlim.so")
        src = [os.path.join(_HERE, "emu", "emu.cpp")] + [os.path.join(_HERE, "..", "dojo.jl_amd", "csrc", f)
                                                         for f in ("dojo_device.hpp", "dojo_host.hpp", "dojo_math.hpp")]
        _build(so, src, ("-DDJ_MLIM=1", "-DDJ_CUT=1"))
        _lib_mlim = C.CDLL(so)
        _lib_mlim.emu_step.restype = C.c_int
    return _lib_mlim


_lib_ss = None


def lib_ss():
    """the emulator compiled like the GPU's body-body contact builds (-DDJ_TSD=0 with DJ_SS=1)"""
    global _lib_ss
    if _lib_ss is None:
        so = os.path.join(_HERE, "emu", "libemu_ss.so")
        src = [os.path.join(_HERE, "emu", "emu.cpp")] + [os.path.join(_HERE, "..", "dojo.jl_amd", "csrc", f)
                                                         for f in ("dojo_device.hpp", "dojo_host.hpp", "dojo_math.hpp")]
        _build(so, src, ("-DDJ_TSD=0",))
        _lib_ss = C.CDLL(so)
        _lib_ss.emu_step.restype = C.c_int
    return _lib_ss


def lib():
    global _lib
    if _lib is None:
        if os.environ.get("EMU_LIB"):                      # an experimental build (g++ ... -DDJ_xxx=1), no rebuild check
            _lib = C.CDLL(os.environ["EMU_LIB"]); _lib.emu_step.restype = C.c_int
            return _lib
        so = os.path.join(_HERE, "emu", "libemu.so")
        src = [os.path.join(_HERE, "emu", "emu.cpp")] + [os.path.join(_HERE, "..", "dojo.jl_amd", "csrc", f)
                                                         for f in ("dojo_device.hpp", "dojo_host.hpp", "dojo_math.hpp")]
        _build(so, src)
        _lib = C.CDLL(so)
        _lib.emu_step.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def emu_step(spec, Z, U=None, opts=None, dtype="f64", grad=False, grad_mode=0, envs_per_wave=1, debug=False, quad=False, fext=None, refine=None):
    """refine: stiffness threshold of the refining kernels (dojo_set_refinement); None = the library's policy (1e4 when the
    tolerances are tighter than the reference's defaults, never otherwise).  Handed to the emulator in EMU_REFINE_W."""
    o_ = opts or SolverOptions()
    thr = refine if refine is not None else (1e4 if (o_.rtol <= 1e-7 or o_.btol <= 1e-6) else float("inf"))
    os.environ["EMU_REFINE_W"] = "inf" if thr == float("inf") else repr(float(thr))
    Z = np.ascontiguousarray(np.atleast_2d(Z), dtype=np.float64); B = Z.shape[0]
    U = None if U is None else np.ascontiguousarray(np.atleast_2d(U), dtype=np.float64)
    topo, keep = spec.to_ctypes()
    o = (opts or SolverOptions()).to_c()
    nx, nu = 12 * spec.Nb, spec.nu
    Zn = np.zeros_like(Z); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    vel = np.zeros((B, 6 * spec.Nb)); jimp = np.zeros((B, max(spec.n_joint_impulses, 1))); linear = any(c.model == 2 for c in spec.contacts); cper = 12 if linear else 8
    csg = np.zeros((B, max(cper * len(spec.contacts), 1)))
    dz = np.zeros((B, nx, nx)) if grad else None
    du = np.zeros((B, max(nu, 1), nx)) if grad else None
    dbg = np.zeros((B, spec.Nb, 512)) if debug else None
    ncc = 5 * len(spec.contacts)
    dc = np.zeros((B, max(ncc, 1), nx)) if (grad and quad and ncc) else None
    stor = np.zeros((B, spec.Nb, 25))
    fext = None if fext is None else np.ascontiguousarray(np.asarray(fext, dtype=np.float64).reshape(B, 6 * spec.Nb))
    err = C.create_string_buffer(256)
    ss = any(getattr(c, "collision", 0) == 1 for c in spec.contacts)          # body-body contacts (tree edges): the emulator built with the GPU builds' flags
    mlim = any((j.tra.nlim > 1 or j.rot.nlim > 1 or (j.tra.nlim > 0 and j.rot.nlim > 0)) for j in spec.joints)     # limits on several coordinates / both halves
    mlim = mlim or len({j.child for j in spec.joints}) < len(spec.joints)                                             # ... or a body with two parent joints (a loop)
    tree_parent = {j.child: j.parent for j in spec.joints if not getattr(j, "loop", False)}
    mlim = mlim or any(getattr(c, "collision", 0) == 1 and tree_parent.get(c.child_body) != c.body for c in spec.contacts)   # ... or a body-body contact that is no tree edge
    # ... or a tree-edge body-body contact the quad builds do not serve (lane mapping, more than 16 bodies, several contacts per body): promoted to a cut element
    owners = [getattr(c, "child_body", -1) if getattr(c, "collision", 0) == 1 else c.body for c in spec.contacts]
    mlim = mlim or (ss and not linear and (not quad or spec.Nb > 16 or (owners and max(owners.count(b) for b in set(owners)) > 1)))
    rc = (lib_mlim() if mlim else lib_linear() if linear else lib_ss() if ss else lib()).emu_step(C.byref(topo), C.byref(o), grad_mode, {"f64": 0, "f32": 1, "f32mixed": 3}[dtype], int(quad), B, envs_per_wave,
                        _p(Z), _p(U), _p(Zn), _p(st), _p(it), _p(vel), _p(jimp), _p(csg), _p(dz), _p(du), _p(dbg), err, 256, _p(dc), _p(stor), _p(fext))
    if rc != 0:
        raise RuntimeError("emu_step: %d %s" % (rc, err.value.decode()))
    out = dict(z_next=Zn, status=st, iters=it, vel=vel, storage=stor, joint_imp=jimp[:, :spec.n_joint_impulses], contact_sg=csg[:, :cper * len(spec.contacts)])
    if debug:
        out["dbg"] = dbg
    if grad:
        # device layout is column-major per environment: dz[b][col][row]
        out["dz"] = dz.transpose(0, 2, 1).copy()
        if dc is not None:
            out["dc"] = dc.transpose(0, 2, 1).copy()          # [B, 12Nb, 5Nc]
        out["du"] = du.reshape(-1)[:B * nu * nx].reshape(B, nu, nx).transpose(0, 2, 1).copy() if nu else np.zeros((B, nx, 0))
    return out
