"""Flat description of a mechanism: Python mirror of the C-POD structs of
include/dojo_hip.h (DojoBody / DojoJoint / DojoContact / DojoTopology).

In the Julia deployment these structs are filled by the DojoHIP.jl shim from a live
`Mechanism` (julia/DojoHIP.jl); here the same structs are filled by the Python
mechanism builders (mechanisms.py) which restate the reference's builders
(DojoEnvironments/src/mechanisms/*/mechanism.jl).
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional
import numpy as np

d3 = C.c_double * 3
d4 = C.c_double * 4
d6 = C.c_double * 6
d9 = C.c_double * 9


class CBody(C.Structure):
    _fields_ = [("mass", C.c_double), ("inertia", d9)]


class CJointHalf(C.Structure):
    _fields_ = [("nl", C.c_int32), ("nlim", C.c_int32), ("cmask", d9), ("amask", d9),
                ("spring", C.c_double), ("damper", C.c_double), ("spring_offset", d3),
                ("limit_lo", d3), ("limit_hi", d3)]


class CJoint(C.Structure):
    _fields_ = [("parent", C.c_int32), ("child", C.c_int32), ("spring_on", C.c_int32), ("damper_on", C.c_int32),
                ("vertex_parent", d3), ("vertex_child", d3), ("orientation_offset", d4),
                ("tra", CJointHalf), ("rot", CJointHalf)]


class CContact(C.Structure):
    _fields_ = [("body", C.c_int32), ("model", C.c_int32), ("friction_coefficient", C.c_double),
                ("normal", d3), ("tangent", d6), ("origin", d3), ("radius", C.c_double), ("offset", d3),
                ("collision", C.c_int32), ("child_body", C.c_int32), ("child_origin", d3), ("child_radius", C.c_double)]


class CTopology(C.Structure):
    _fields_ = [("n_bodies", C.c_int32), ("n_joints", C.c_int32), ("n_contacts", C.c_int32), ("reserved", C.c_int32),
                ("timestep", C.c_double), ("input_scaling", C.c_double), ("gravity", d3),
                ("bodies", C.POINTER(CBody)), ("joints", C.POINTER(CJoint)), ("contacts", C.POINTER(CContact))]


class CSolverOptions(C.Structure):
    _fields_ = [("rtol", C.c_double), ("btol", C.c_double), ("undercut", C.c_double), ("no_progress_undercut", C.c_double),
                ("max_iter", C.c_int32), ("max_ls", C.c_int32), ("no_progress_max", C.c_int32), ("reserved", C.c_int32)]


class CDims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_bodies", "n_joints", "n_contacts", "nz", "nx", "nu",
                                          "n_joint_impulses", "n_solution", "lanes_per_env")]


@dataclass
class SolverOptions:
    """SolverOptions{T}, src/solver/options.jl:16-26 (same defaults)."""
    rtol: float = 1e-6
    btol: float = 1e-4
    max_iter: int = 50
    max_ls: int = 10
    undercut: float = float("inf")
    no_progress_max: int = 3
    no_progress_undercut: float = 10.0

    def to_c(self):
        return CSolverOptions(self.rtol, self.btol, self.undercut, self.no_progress_undercut,
                              self.max_iter, self.max_ls, self.no_progress_max, 0)


@dataclass
class BodySpec:
    name: str
    mass: float
    inertia: np.ndarray            # 3x3


@dataclass
class JointHalfSpec:
    """Translational{T,Nλ,...} / Rotational{T,Nλ,...}"""
    nl: int                        # Nλ
    axis: np.ndarray = field(default_factory=lambda: np.zeros(3))
    spring: float = 0.0
    damper: float = 0.0
    spring_offset: np.ndarray = field(default_factory=lambda: np.zeros(0))
    limits: Optional[tuple] = None  # (lo[], hi[])

    def masks(self):
        """constraint_mask / nullspace_mask, src/joints/joint.jl:56-64"""
        from .quat import orthogonal_rows
        V1, V2, V3 = orthogonal_rows(self.axis)
        I = np.eye(3)
        cm = {0: np.zeros((0, 3)), 1: V3[None, :], 2: np.stack([V1, V2]), 3: I}[self.nl]
        am = {0: I, 1: np.stack([V1, V2]), 2: V3[None, :], 3: np.zeros((0, 3))}[self.nl]
        return cm, am

    @property
    def nlim(self):
        return 0 if self.limits is None else len(self.limits[0])

    @property
    def N(self):
        return self.nl + 4 * self.nlim

    @property
    def nu(self):
        return 3 - self.nl


@dataclass
class JointSpec:
    name: str
    parent: int                    # body index, -1 = origin
    child: int
    tra: JointHalfSpec
    rot: JointHalfSpec
    vertex_parent: np.ndarray = field(default_factory=lambda: np.zeros(3))
    vertex_child: np.ndarray = field(default_factory=lambda: np.zeros(3))
    orientation_offset: np.ndarray = field(default_factory=lambda: np.array([1.0, 0, 0, 0]))
    loop: bool = False             # a loop-closing joint (URDF <loop_joint>, src/mechanism/urdf.jl): its child body already hangs on another joint

    @property
    def spring_on(self):
        return self.tra.spring != 0 or self.rot.spring != 0

    @property
    def damper_on(self):
        return self.tra.damper != 0 or self.rot.damper != 0

    @property
    def N(self):
        return self.tra.N + self.rot.N

    @property
    def nu(self):
        return self.tra.nu + self.rot.nu


@dataclass
class ContactSpec:
    name: str
    body: int
    friction_coefficient: float
    normal: np.ndarray
    tangent: np.ndarray            # 2x3
    origin: np.ndarray
    radius: float = 0.0
    offset: np.ndarray = field(default_factory=lambda: np.zeros(3))
    model: int = 0                 # 0: NonlinearContact, 1: ImpactContact (src/contacts/impact.jl)
    collision: int = 0             # 0: SphereHalfSpaceCollision (child = origin), 1: SphereSphereCollision (src/contacts/collisions/sphere_sphere.jl)
    child_body: int = -1           # collision 1: the contact's child body (its joint hangs on `body`); (origin, radius) = the parent's sphere
    child_origin: np.ndarray = field(default_factory=lambda: np.zeros(3))
    child_radius: float = 0.0


@dataclass
class MechanismSpec:
    """What the boundary needs to know about a Dojo `Mechanism`."""
    name: str
    bodies: List[BodySpec]
    joints: List[JointSpec]
    contacts: List[ContactSpec]
    timestep: float = 0.01
    input_scaling: Optional[float] = None
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))

    def __post_init__(self):
        if self.input_scaling is None:
            self.input_scaling = self.timestep
        g = np.atleast_1d(np.asarray(self.gravity, dtype=float))
        self.gravity = np.array([0.0, 0.0, g[0]]) if g.size == 1 else g   # get_gravity, src/mechanism/gravity.jl

    # dimensions (same formulas as the reference)
    @property
    def Nb(self): return len(self.bodies)
    @property
    def nz(self): return 13 * self.Nb
    @property
    def nx(self): return 12 * self.Nb
    @property
    def nu(self): return sum(j.nu for j in self.joints)
    @property
    def n_joint_impulses(self): return sum(j.N for j in self.joints)
    @property
    def n_solution(self): return self.n_joint_impulses + 6 * self.Nb + sum({1: 2, 2: 12}.get(c.model, 8) for c in self.contacts)      # [s; γ] per contact: N = 8 (NonlinearContact), 2 (ImpactContact), 12 (LinearContact)

    def body_index(self, name): return [b.name for b in self.bodies].index(name)
    def joint_index(self, name): return [j.name for j in self.joints].index(name)

    def input_slice(self, joint_name):
        off = 0
        for j in self.joints:
            if j.name == joint_name:
                return slice(off, off + j.nu)
            off += j.nu
        raise KeyError(joint_name)

    def to_ctypes(self):
        """-> (CTopology, keepalive)"""
        nb, nj, nc = len(self.bodies), len(self.joints), len(self.contacts)
        B = (CBody * max(nb, 1))()
        J = (CJoint * max(nj, 1))()
        K = (CContact * max(nc, 1))()
        for i, b in enumerate(self.bodies):
            B[i].mass = float(b.mass)
            B[i].inertia = d9(*np.asarray(b.inertia, dtype=float).reshape(9))

        def fill_half(h, s):
            cm, am = s.masks()
            h.nl, h.nlim = s.nl, s.nlim
            h.cmask = d9(*np.concatenate([cm.reshape(-1), np.zeros(9 - cm.size)]))
            h.amask = d9(*np.concatenate([am.reshape(-1), np.zeros(9 - am.size)]))
            h.spring, h.damper = float(s.spring), float(s.damper)
            so = np.zeros(3); so[:len(s.spring_offset)] = s.spring_offset
            h.spring_offset = d3(*so)
            lo, hi = np.zeros(3), np.zeros(3)
            if s.limits is not None:
                lo[:s.nlim] = s.limits[0]; hi[:s.nlim] = s.limits[1]
            h.limit_lo, h.limit_hi = d3(*lo), d3(*hi)

        # The C side takes the FIRST joint that names a body as its tree joint and every later one as loop-closing (dojo_host.hpp, build_host_model);
        # the host-side traversals (coords._root_to_leaves) go by JointSpec.loop.  The two must describe the same spanning tree.
        seen = set()
        for j in self.joints:
            if getattr(j, "loop", False) != (j.child in seen):
                raise ValueError("joint %r: %s" % (j.name, "flagged loop=True but no earlier joint has body %d as its child: list its tree joint first" % j.child
                                                   if getattr(j, "loop", False) else "body %d already hangs on an earlier joint: flag this one loop=True" % j.child))
            seen.add(j.child)
        for i, j in enumerate(self.joints):
            J[i].parent, J[i].child = j.parent, j.child
            J[i].spring_on, J[i].damper_on = int(j.spring_on), int(j.damper_on)
            J[i].vertex_parent = d3(*j.vertex_parent)
            J[i].vertex_child = d3(*j.vertex_child)
            J[i].orientation_offset = d4(*j.orientation_offset)
            fill_half(J[i].tra, j.tra)
            fill_half(J[i].rot, j.rot)
        for i, c in enumerate(self.contacts):
            K[i].body = c.body
            K[i].model = int(c.model)
            K[i].friction_coefficient = float(c.friction_coefficient)
            K[i].normal = d3(*c.normal)
            K[i].tangent = d6(*np.asarray(c.tangent).reshape(6))
            K[i].origin = d3(*c.origin)
            K[i].radius = float(c.radius)
            K[i].offset = d3(*c.offset)
            K[i].collision, K[i].child_body = int(c.collision), int(c.child_body)
            K[i].child_origin = d3(*c.child_origin); K[i].child_radius = float(c.child_radius)
        T = CTopology(nb, nj, nc, 0, float(self.timestep), float(self.input_scaling), d3(*self.gravity),
                      C.cast(B, C.POINTER(CBody)), C.cast(J, C.POINTER(CJoint)), C.cast(K, C.POINTER(CContact)))
        return T, (B, J, K)
