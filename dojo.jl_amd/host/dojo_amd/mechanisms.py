"""Mechanism builders for the five BASELINE.json configs, restating
DojoEnvironments/src/mechanisms/{pendulum,block,ant,quadruped,atlas}/mechanism.jl and the
URDF -> maximal-coordinate conversion of src/mechanism/urdf.jl.

These run once at set-up time on the host (they stand in for the Julia builders, which
cannot run here); the result is a MechanismSpec = the flat topology the C ABI consumes.
Body order is URDF document order (the reference's Dict order is not reproducible,
SURVEY.md Appendix A-2): everything is looked up by name.
"""
import json
import os
import numpy as np
from .quat import (qmul, qinv, vrot, rpy_to_quat, orthogonal_rows, axis_angle_to_quaternion)
from .topology import BodySpec, JointHalfSpec, JointSpec, ContactSpec, MechanismSpec

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
X_AXIS, Y_AXIS, Z_AXIS = np.eye(3)


# ----------------------------------------------------------------------------------
# joint prototypes, src/joints/prototypes.jl
# ----------------------------------------------------------------------------------
def Floating(name, parent, child):                                   # prototypes.jl:429-447
    return JointSpec(name, parent, child, JointHalfSpec(0, spring_offset=np.zeros(3)), JointHalfSpec(0, spring_offset=np.zeros(3)))


def Fixed(name, parent, child, parent_vertex=np.zeros(3), child_vertex=np.zeros(3), orientation_offset=np.array([1.0, 0, 0, 0])):
    return JointSpec(name, parent, child, JointHalfSpec(3), JointHalfSpec(3),                 # prototypes.jl:6-16
                     np.array(parent_vertex, float), np.array(child_vertex, float), np.array(orientation_offset, float))


def Revolute(name, parent, child, axis, parent_vertex=np.zeros(3), child_vertex=np.zeros(3),
             orientation_offset=np.array([1.0, 0, 0, 0]), spring=0.0, damper=0.0, rot_spring_offset=np.zeros(1), rot_joint_limits=None):
    # prototypes.jl:94-118: the same spring/damper value goes to both halves; Translational{T,3} ignores it
    return JointSpec(name, parent, child,
                     JointHalfSpec(3, spring=spring, damper=damper),
                     JointHalfSpec(2, axis=np.array(axis, float), spring=spring, damper=damper,
                                   spring_offset=np.array(rot_spring_offset, float), limits=rot_joint_limits),
                     np.array(parent_vertex, float), np.array(child_vertex, float), np.array(orientation_offset, float))


def Prismatic(name, parent, child, axis, parent_vertex=np.zeros(3), child_vertex=np.zeros(3),
              orientation_offset=np.array([1.0, 0, 0, 0]), spring=0.0, damper=0.0, tra_spring_offset=np.zeros(1), tra_joint_limits=None):
    return JointSpec(name, parent, child,                                                     # prototypes.jl:21-41
                     JointHalfSpec(2, axis=np.array(axis, float), spring=spring, damper=damper,
                                   spring_offset=np.array(tra_spring_offset, float), limits=tra_joint_limits),
                     JointHalfSpec(3, spring=spring, damper=damper),
                     np.array(parent_vertex, float), np.array(child_vertex, float), np.array(orientation_offset, float))


def Spherical(name, parent, child, parent_vertex=np.zeros(3), child_vertex=np.zeros(3),
              orientation_offset=np.array([1.0, 0, 0, 0]), spring=0.0, damper=0.0):
    return JointSpec(name, parent, child, JointHalfSpec(3, spring=spring, damper=damper),       # prototypes.jl:343-362
                     JointHalfSpec(0, spring=spring, damper=damper, spring_offset=np.zeros(3)),
                     np.array(parent_vertex, float), np.array(child_vertex, float), np.array(orientation_offset, float))


# Dojo.Prototype(joint_type, pbody, cbody, axis; ...)  src/joints/prototypes.jl:455-480: (Nλ translational, Nλ rotational)
PROTOTYPES = {"Fixed": (3, 3), "Prismatic": (2, 3), "Planar": (1, 3), "FixedOrientation": (0, 3), "Revolute": (3, 2), "Cylindrical": (2, 2),
              "PlanarAxis": (1, 2), "FreeRevolute": (0, 2), "Orbital": (3, 1), "PrismaticOrbital": (2, 1), "PlanarOrbital": (1, 1),
              "FreeOrbital": (0, 1), "Spherical": (3, 0), "CylindricalFree": (2, 0), "PlanarFree": (1, 0), "Floating": (0, 0)}


def Prototype(joint_type, name, parent, child, axis, parent_vertex=np.zeros(3), child_vertex=np.zeros(3),
              orientation_offset=np.array([1.0, 0, 0, 0]), spring=0.0, damper=0.0, tra_spring_offset=None, rot_spring_offset=None):
    """Any of the sixteen joint prototypes: Translational{T,Nλt} + Rotational{T,Nλr} about / along `axis` with the same spring and
    damper value on both halves (prototypes.jl:6-447).  CylindricalFree / PlanarFree take no orientation offset there."""
    nl_t, nl_r = PROTOTYPES[joint_type]
    if joint_type in ("CylindricalFree", "PlanarFree", "Floating"):
        orientation_offset = np.array([1.0, 0, 0, 0])
    tso = np.zeros(3 - nl_t) if tra_spring_offset is None else np.array(tra_spring_offset, float)
    rso = np.zeros(3 - nl_r) if rot_spring_offset is None else np.array(rot_spring_offset, float)
    return JointSpec(name, parent, child,
                     JointHalfSpec(nl_t, axis=np.array(axis, float), spring=spring, damper=damper, spring_offset=tso),
                     JointHalfSpec(nl_r, axis=np.array(axis, float), spring=spring, damper=damper, spring_offset=rso),
                     np.array(parent_vertex, float), np.array(child_vertex, float), np.array(orientation_offset, float))


def box_inertia(x, y, z, m):                                          # src/bodies/shapes.jl:90
    return m / 12.0 * np.diag([y * y + z * z, x * x + z * z, x * x + y * y])


CONTACT_MODELS = {"nonlinear": 0, "impact": 1, "linear": 2}     # "linear": LinearContact, src/contacts/linear.jl (forward only, like the reference)


def contact_constraint(name, body, normal, friction_coefficient=1.0, contact_origin=np.zeros(3), contact_radius=0.0,
                       contact_offset=np.zeros(3), contact_type="nonlinear"):
    """NonlinearContact(body, normal, μ; ...)  src/contacts/nonlinear.jl:24-48;  contact_type="impact":
    ImpactContact(body, normal; ...)  src/contacts/impact.jl:20-39 (no friction: μ and the tangents are unused)"""
    V1, V2, V3 = orthogonal_rows(normal)
    A = np.stack([V1, V2, V3], axis=1)          # orthogonal_columns
    Ainv = np.linalg.inv(A)
    return ContactSpec(name, body, float(friction_coefficient), Ainv[2].copy(), Ainv[0:2].copy(),
                       np.array(contact_origin, float), float(contact_radius), np.array(contact_offset, float),
                       CONTACT_MODELS[contact_type])


def sphere_sphere_contact(name, parent_body, child_body, radius_parent, radius_child, friction_coefficient=0.5, contact_type="nonlinear",
                          origin_parent=np.zeros(3), origin_child=np.zeros(3)):
    """ContactConstraint((NonlinearContact | LinearContact | ImpactContact with SphereSphereCollision(origin_parent, origin_child, r_parent,
    r_child), parent_id, child_id))  src/contacts/collisions/sphere_sphere.jl:11-16; test/collisions.jl:19-52 builds it with the spheres about
    the two centres of mass.  Forward only.  `child_body`'s joint hanging on `parent_body` makes the contact an edge of the tree (quad mapping);
    any other pair of bodies is served by the general lane-mapping builds (at most two such contacts / loop joints per mechanism)."""
    return ContactSpec(name, parent_body, float(friction_coefficient), np.zeros(3), np.zeros((2, 3)), np.array(origin_parent, float), float(radius_parent), np.zeros(3),
                       CONTACT_MODELS[contact_type], collision=1, child_body=child_body, child_origin=np.array(origin_child, float), child_radius=float(radius_child))


def get_two_spheres(timestep=0.1, input_scaling=None, gravity=-9.81, radius_body1=0.5, radius_body2=0.5, mass_body1=1.0, mass_body2=1.0,
                    friction_type="nonlinear", friction_coefficient=0.5, joint_world_body1="Fixed", free_on="body1"):
    """get_two_body of test/collisions.jl:2-58: a sphere on a Fixed (or Floating) joint to the world and a second, free sphere that
    touches it through a SphereSphereCollision contact.  The reference gives the second sphere no joint at all; here it carries a
    Floating joint (no rows, inputs left at zero) to the first sphere, which makes the contact an edge of the tree next to it."""
    bodies = [BodySpec("sphere1", mass_body1, sphere_inertia(radius_body1, mass_body1)), BodySpec("sphere2", mass_body2, sphere_inertia(radius_body2, mass_body2))]
    # free_on="world": the second sphere hangs on the ORIGIN (a free body, as in the reference), so the contact connects two bodies that are no
    # tree neighbours -- a cut element of the general lane-mapping builds (round 5); "body1": the contact is an edge of the tree (quad builds)
    joints = [Fixed("joint", -1, 0) if joint_world_body1 == "Fixed" else Floating("joint", -1, 0), Floating("free", 0 if free_on == "body1" else -1, 1)]
    contacts = [sphere_sphere_contact("body_body", 0, 1, radius_body1, radius_body2, friction_coefficient, friction_type)]
    return MechanismSpec("two_spheres", bodies, joints, contacts, timestep, input_scaling, gravity)


def set_limits(spec, joint_limits):
    """DojoEnvironments/src/utilities.jl:41-58 + src/joints/limits.jl:31-61 (one-dimensional joints only)"""
    for jname, lim in joint_limits.items():
        j = spec.joints[spec.joint_index(jname)]
        if j.tra.nu == 0 and j.rot.nu == 1:
            j.rot.limits = (np.array([lim[0]], float), np.array([lim[1]], float))
        elif j.tra.nu == 1 and j.rot.nu == 0:
            j.tra.limits = (np.array([lim[0]], float), np.array([lim[1]], float))
        else:
            raise ValueError("joint limits can only be set for one-dimensional joints")


def add_limits(spec, joint_name, tra_limits=None, rot_limits=None):
    """add_limits(mech, joint; tra_limits, rot_limits)  src/joints/limits.jl:31-61: limits (lo[], hi[]) on ALL free coordinates of a joint half
    (Nb½ = the half's number of minimal coordinates: `constraint` broadcasts them against `minimal_coordinates`, limits.jl:1-17), on either
    half or both -- e.g. three rotation-vector limits on a Spherical joint, two on a Planar joint's translation, one + one on a Cylindrical."""
    j = spec.joints[spec.joint_index(joint_name)]
    for half, lim in ((j.tra, tra_limits), (j.rot, rot_limits)):
        if lim is None:
            continue
        lo, hi = np.atleast_1d(np.array(lim[0], float)), np.atleast_1d(np.array(lim[1], float))
        if len(lo) != half.nu or len(hi) != half.nu:
            raise ValueError("limits on %s need %d entries per side (all free coordinates of the half)" % (joint_name, half.nu))
        half.limits = (lo, hi)
    return spec


def get_fourbar(timestep=0.01, input_scaling=None, gravity=-9.81, springs=0.0, dampers=0.0, parse_dampers=True):
    """DojoEnvironments/src/mechanisms/fourbar/mechanism.jl:1-36 + dependencies/fourbar.urdf (floating = false): four unit links (boxes 0.1 x 0.1 x 1,
    centre of mass half a metre below their joint), link1 and link3 on continuous joints about x to the world at z = 2.1, link2 / link4 below
    them, and the LOOP joint joint24 between the free ends of link2 and link4 (URDF <loop_joint>): a kinematic loop, the graph is no tree.
    URDF damping 1.0 on every joint (parse_dampers), joints in document order: jointb1, joint12, jointb3, joint34, joint24."""
    J = np.diag([0.0841667, 0.0841667, 0.00166667])
    bodies = [BodySpec("link%d" % i, 1.0, J.copy()) for i in (1, 2, 3, 4)]
    dmp = 1.0 if parse_dampers else 0.0
    top, up, dn = np.array([0, 0, 2.1]), 0.5 * Z_AXIS, -0.5 * Z_AXIS
    joints = [Revolute("jointb1", -1, 0, X_AXIS, parent_vertex=top, child_vertex=up, damper=dmp),
              Revolute("joint12", 0, 1, X_AXIS, parent_vertex=dn, child_vertex=up, damper=dmp),
              Revolute("jointb3", -1, 2, X_AXIS, parent_vertex=top, child_vertex=up, damper=dmp),
              Revolute("joint34", 2, 3, X_AXIS, parent_vertex=dn, child_vertex=up, damper=dmp),
              Revolute("joint24", 1, 3, X_AXIS, parent_vertex=dn, child_vertex=dn, damper=dmp)]
    joints[4].loop = True
    spec = MechanismSpec("fourbar", bodies, joints, [], timestep, input_scaling, gravity)
    if not parse_dampers:
        _set_per_joint(spec, springs, dampers)
    return spec


def get_limited_chain(kind="spherical", timestep=0.01, input_scaling=None, gravity=-9.81, dampers=0.05):
    """Test mechanisms for joint limits on several coordinates (src/joints/limits.jl:1-61; the DojoEnvironments builders only limit
    one-dimensional joints): a link hanging from the origin on
      "spherical":   a Spherical joint with limits on its three rotation-vector coordinates, a second link below it on a Revolute joint;
      "planar":      a Planar joint (translation free in two directions) with limits on both, rotation locked;
      "cylindrical": a Cylindrical joint (one translation + one rotation about the same axis) with a limit on either half;
      "mixed":       spherical (3 limits) -> cylindrical (1 + 1) -> a foot with a contact."""
    L = 0.5
    def link(n): return BodySpec(n, 1.0, box_inertia(0.1, 0.1, L, 1.0))
    if kind == "spherical":
        bodies = [link("upper"), link("lower")]
        joints = [Prototype("Spherical", "shoulder", -1, 0, Z_AXIS, parent_vertex=np.array([0, 0, 1.5]), child_vertex=0.5 * L * Z_AXIS, damper=dampers),
                  Revolute("elbow", 0, 1, X_AXIS, parent_vertex=-0.5 * L * Z_AXIS, child_vertex=0.5 * L * Z_AXIS, damper=dampers)]
        spec = MechanismSpec("limited_spherical", bodies, joints, [], timestep, input_scaling, gravity)
        add_limits(spec, "shoulder", rot_limits=([-0.4, -0.3, -0.5], [0.3, 0.5, 0.4]))
    elif kind == "planar":
        bodies = [link("plate")]
        joints = [Prototype("Planar", "table", -1, 0, X_AXIS, parent_vertex=np.array([0, 0, 1.0]), damper=dampers)]     # free: the two directions normal to x
        spec = MechanismSpec("limited_planar", bodies, joints, [], timestep, input_scaling, gravity)
        add_limits(spec, "table", tra_limits=([-0.2, -0.3], [0.25, 0.1]))
    elif kind == "cylindrical":
        bodies = [link("rod")]
        joints = [Prototype("Cylindrical", "sleeve", -1, 0, Z_AXIS, parent_vertex=np.array([0, 0, 1.0]), damper=dampers)]
        spec = MechanismSpec("limited_cylindrical", bodies, joints, [], timestep, input_scaling, gravity)
        add_limits(spec, "sleeve", tra_limits=([-0.3], [0.2]), rot_limits=([-0.5], [0.4]))
    elif kind == "mixed":
        bodies = [link("upper"), link("rod"), BodySpec("foot", 0.5, sphere_inertia(0.1, 0.5))]
        joints = [Prototype("Spherical", "shoulder", -1, 0, Z_AXIS, parent_vertex=np.array([0, 0, 1.3]), child_vertex=0.5 * L * Z_AXIS, damper=dampers),
                  Prototype("Cylindrical", "sleeve", 0, 1, Z_AXIS, parent_vertex=-0.5 * L * Z_AXIS, child_vertex=0.5 * L * Z_AXIS, damper=dampers),
                  Revolute("ankle", 1, 2, Y_AXIS, parent_vertex=-0.5 * L * Z_AXIS, child_vertex=0.1 * Z_AXIS, damper=dampers)]
        contacts = [contact_constraint("sole", 2, Z_AXIS, 0.8, contact_radius=0.1)]
        spec = MechanismSpec("limited_mixed", bodies, joints, contacts, timestep, input_scaling, gravity)
        add_limits(spec, "shoulder", rot_limits=([-0.6, -0.5, -0.5], [0.5, 0.6, 0.5]))
        add_limits(spec, "sleeve", tra_limits=([-0.15], [0.15]), rot_limits=([-0.4], [0.4]))
    else:
        raise ValueError(kind)
    return spec


def set_springs_dampers(spec, springs=0.0, dampers=0.0):
    """set_springs!/set_dampers!  DojoEnvironments/src/utilities.jl:1-39 (floating base skipped)"""
    for j in spec.joints:
        if j.N == 0:
            continue
        if springs != 0:
            j.tra.spring = j.rot.spring = float(springs)
        if dampers != 0:
            j.tra.damper = j.rot.damper = float(dampers)


# ----------------------------------------------------------------------------------
# pendulum / block   (hand-built mechanisms)
# ----------------------------------------------------------------------------------
def get_pendulum(timestep=0.01, input_scaling=None, gravity=-9.81, mass=1.0, link_length=1.0, springs=0.0, dampers=0.0,
                 joint_limits=None, spring_offset=np.zeros(1), orientation_offset=np.array([1.0, 0, 0, 0])):
    """DojoEnvironments/src/mechanisms/pendulum/mechanism.jl:1-44"""
    bodies = [BodySpec("pendulum", mass, box_inertia(0.1, 0.1, link_length, mass))]
    joints = [Revolute("joint", -1, 0, X_AXIS, parent_vertex=(link_length + 0.1) * Z_AXIS, child_vertex=0.5 * link_length * Z_AXIS,
                       rot_spring_offset=spring_offset, orientation_offset=orientation_offset)]
    spec = MechanismSpec("pendulum", bodies, joints, [], timestep, input_scaling, gravity)
    set_springs_dampers(spec, springs, dampers)
    set_limits(spec, joint_limits or {})
    return spec


def get_block(timestep=0.01, input_scaling=None, gravity=-9.81, mass=1.0, edge_length=0.5, friction_coefficient=0.8,
              contact=True, contact_radius=0.0, contact_corners=8, contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/block/mechanism.jl:1-70.  contact_corners=4 keeps the four
    bottom corners only (BASELINE.json config 2); the reference's default is all 8."""
    bodies = [BodySpec("block", mass, box_inertia(edge_length, edge_length, edge_length, mass))]
    joints = [Floating("floating_base", -1, 0)]
    h = edge_length / 2
    origins = [[h, h, -h], [h, -h, -h], [-h, h, -h], [-h, -h, -h], [h, h, h], [h, -h, h], [-h, h, h], [-h, -h, h]]
    contacts = []
    if contact:
        for i, o in enumerate(origins[:contact_corners]):
            contacts.append(contact_constraint("contact%d" % (i + 1), 0, Z_AXIS, friction_coefficient, o, contact_radius, contact_type=contact_type))
    return MechanismSpec("block", bodies, joints, contacts, timestep, input_scaling, gravity)


def _set_per_joint(spec, springs, dampers):
    """set_springs!/set_dampers! with one value per joint (DojoEnvironments/src/utilities.jl:11-19,31-39) or a scalar"""
    if np.ndim(springs) == 0 and np.ndim(dampers) == 0:
        return set_springs_dampers(spec, float(springs), float(dampers))
    sp = np.broadcast_to(np.asarray(springs, float), (len(spec.joints),))
    da = np.broadcast_to(np.asarray(dampers, float), (len(spec.joints),))
    for j, k, c in zip(spec.joints, sp, da):
        if j.N == 0:
            continue
        if k != 0:
            j.tra.spring = j.rot.spring = float(k)
        if c != 0:
            j.tra.damper = j.rot.damper = float(c)


def sphere_inertia(r, m):                                             # src/bodies/shapes.jl (Sphere): 2/5 m r^2
    return 0.4 * m * r * r * np.eye(3)


def get_slider(timestep=0.01, input_scaling=None, gravity=-9.81, springs=0.0, dampers=0.0, joint_limits=None):
    """DojoEnvironments/src/mechanisms/slider/mechanism.jl:1-39: one box on a Prismatic joint along z"""
    bodies = [BodySpec("pbody", 1.0, box_inertia(0.1, 0.1, 1.0, 1.0))]
    joints = [Prismatic("joint", -1, 0, Z_AXIS, child_vertex=Z_AXIS / 2)]
    spec = MechanismSpec("slider", bodies, joints, [], timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    if joint_limits:
        set_limits(spec, joint_limits)
    return spec


def cylinder_inertia(r, h, m):                                        # src/bodies/shapes.jl:130 (Cylinder, axis z), as the reference has it
    return 0.5 * m * np.diag([r * r + h * h / 6.0, r * r + h * h / 6.0, r * r])


def get_nslider(timestep=0.01, input_scaling=None, gravity=-9.81, num_bodies=5, springs=0.0, dampers=0.0):
    """DojoEnvironments/src/mechanisms/nslider/mechanism.jl:1-40: cylinders on Prismatic joints along z, each to the previous one"""
    bodies = [BodySpec("body:%d" % (i + 1), 1.0, cylinder_inertia(0.05, 1.0, 1.0)) for i in range(num_bodies)]
    joints = [Prismatic("joint:1", -1, 0, Z_AXIS)]
    for i in range(1, num_bodies):
        joints.append(Prismatic("joint:%d" % (i + 1), i - 1, i, Z_AXIS, parent_vertex=-0.05 * Y_AXIS, child_vertex=0.05 * Y_AXIS))
    spec = MechanismSpec("nslider", bodies, joints, [], timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    return spec


def get_npendulum(timestep=0.01, input_scaling=None, gravity=-9.81, num_bodies=5, mass=1.0, link_length=1.0, springs=0.0, dampers=0.0,
                  base_joint_type="Revolute", rest_joint_type="Revolute"):
    """DojoEnvironments/src/mechanisms/npendulum/mechanism.jl:1-43"""
    bodies = [BodySpec("body:%d" % (i + 1), mass, box_inertia(0.05, 0.05, link_length, mass)) for i in range(num_bodies)]
    joints = [Prototype(base_joint_type, "joint:1", -1, 0, X_AXIS, parent_vertex=(link_length + 0.1) * Z_AXIS * num_bodies, child_vertex=Z_AXIS * link_length / 2)]
    for i in range(1, num_bodies):
        joints.append(Prototype(rest_joint_type, "joint:%d" % (i + 1), i - 1, i, X_AXIS, parent_vertex=-Z_AXIS * link_length / 2, child_vertex=Z_AXIS * link_length / 2))
    spec = MechanismSpec("npendulum", bodies, joints, [], timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    return spec


def get_snake(timestep=0.01, input_scaling=None, gravity=-9.81, num_bodies=2, link_length=1.0, radius=0.05, springs=0.0, dampers=0.0,
              joint_type="Spherical", friction_coefficient=0.8, contact=True, contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/snake/mechanism.jl:1-66: boxes chained along x by any joint prototype below a floating base,
    one contact at either end of every link (body order of the contacts: all +x ends, then all −x ends)"""
    bodies = [BodySpec("body:%d" % (i + 1), link_length, box_inertia(link_length, 3 * radius, 2 * radius, link_length)) for i in range(num_bodies)]
    joints = [Floating("floating_base", -1, 0)]
    for i in range(1, num_bodies):
        joints.append(Prototype(joint_type, "joint:%d" % (i + 1), i - 1, i, X_AXIS, parent_vertex=-X_AXIS * link_length / 2, child_vertex=X_AXIS * link_length / 2))
    contacts = []
    if contact:
        for sgn in (1.0, -1.0):
            for i in range(num_bodies):
                contacts.append(contact_constraint("contact:%d" % (len(contacts) + 1), i, Z_AXIS, friction_coefficient, sgn * X_AXIS * link_length / 2,
                                                   contact_type=contact_type))
    spec = MechanismSpec("snake", bodies, joints, contacts, timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    return spec


def get_twister(timestep=0.01, input_scaling=None, gravity=-9.81, num_bodies=5, height=1.0, radius=0.05, springs=0.0, dampers=0.0,
                joint_type="Prismatic", friction_coefficient=0.8, contact=True, contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/twister/mechanism.jl:1-68: like the snake with the joint axis cycling through y, z, x"""
    bodies = [BodySpec("body:%d" % (i + 1), height, box_inertia(height, 3 * radius, 2 * radius, height)) for i in range(num_bodies)]
    axes = [X_AXIS, Y_AXIS, Z_AXIS]
    joints = [Floating("floating_base", -1, 0)]
    for i in range(2, num_bodies + 1):                                  # Julia's i = 2:num_bodies, axes[i % 3 + 1] (1-based)
        joints.append(Prototype(joint_type, "joint:%d" % i, i - 2, i - 1, axes[i % 3], parent_vertex=-X_AXIS * height / 2, child_vertex=X_AXIS * height / 2))
    contacts = []
    if contact:                                                         # [bodies[1]; bodies]: +x end of the first link, −x end of every link
        origins = [X_AXIS * height / 2] + [-X_AXIS * height / 2] * num_bodies
        for k, b in enumerate([0] + list(range(num_bodies))):
            contacts.append(contact_constraint("contact:%d" % (k + 1), b, Z_AXIS, friction_coefficient, origins[k], contact_type=contact_type))
    spec = MechanismSpec("twister", bodies, joints, contacts, timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    return spec


def get_sphere(timestep=0.01, input_scaling=None, gravity=-9.81, mass=1.0, radius=0.5, friction_coefficient=0.8, contact=True, contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/sphere/mechanism.jl:1-45"""
    bodies = [BodySpec("sphere", mass, sphere_inertia(radius, mass))]
    contacts = [contact_constraint("contact", 0, Z_AXIS, friction_coefficient, contact_radius=radius, contact_type=contact_type)] if contact else []
    return MechanismSpec("sphere", bodies, [Floating("floating_base", -1, 0)], contacts, timestep, input_scaling, gravity)


def capsule_inertia(r, h, m):                                         # src/bodies/shapes.jl:157-180 (Capsule, axis z)
    vc, vh = np.pi * h * r ** 2, np.pi * 4.0 / 3.0 * r ** 3 / 2.0
    mc, mh = m * vc / (vc + 2 * vh), m * vh / (vc + 2 * vh)
    dd = 3.0 / 8.0 * r + 0.5 * h
    ixx = mc * (h * h / 12.0 + r * r / 4.0) + 2.0 * (83.0 / 320 * mh * r * r + mh * dd * dd)
    izz = mc * 0.5 * r * r + 2.0 * (mh * 0.4 * r * r / 2.0)
    return np.diag([ixx, ixx, izz])


def get_cartpole(timestep=0.01, input_scaling=None, gravity=-9.81, slider_mass=1.0, pendulum_mass=1.0, link_length=1.0, radius=0.075,
                 springs=0.0, dampers=0.0, joint_limits=None):
    """DojoEnvironments/src/mechanisms/cartpole/mechanism.jl:1-48"""
    bodies = [BodySpec("cart", slider_mass, capsule_inertia(1.5 * radius, 1.0, slider_mass)), BodySpec("pole", pendulum_mass, capsule_inertia(radius, link_length, pendulum_mass))]
    joints = [Prismatic("cart_joint", -1, 0, Y_AXIS), Revolute("pole_joint", 0, 1, X_AXIS, child_vertex=-0.5 * link_length * Z_AXIS)]
    spec = MechanismSpec("cartpole", bodies, joints, [], timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    if joint_limits:
        set_limits(spec, joint_limits)
    return spec


def get_block2d(timestep=0.01, input_scaling=None, gravity=-9.81, mass=1.0, edge_length=0.5, friction_coefficient=0.8, contact=True,
                contact_radius=0.0, contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/block2d/mechanism.jl:1-54: a box on a PlanarAxis joint (y-z plane) with its four corners as contacts"""
    bodies = [BodySpec("block", mass, box_inertia(edge_length, edge_length, edge_length, mass))]
    joints = [Prototype("PlanarAxis", "joint", -1, 0, X_AXIS)]
    h = edge_length / 2
    contacts = [contact_constraint("contact%d" % (i + 1), 0, Z_AXIS, friction_coefficient, o, contact_radius, contact_type=contact_type)
                for i, o in enumerate([[0, h, h], [0, h, -h], [0, -h, h], [0, -h, -h]])] if contact else []
    return MechanismSpec("block2d", bodies, joints, contacts, timestep, input_scaling, gravity)


def get_dzhanibekov(timestep=0.01, input_scaling=None, gravity=-9.81):
    """DojoEnvironments/src/mechanisms/dzhanibekov/mechanism.jl:1-36: two capsules fixed to each other, floating"""
    bodies = [BodySpec("main", 1.0, capsule_inertia(0.1, 1.0, 1.0)), BodySpec("side", 0.5, capsule_inertia(0.05, 0.35, 0.5))]
    joints = [Floating("floating", -1, 0), Fixed("fixed", 0, 1, child_vertex=[-0.25, 0, 0])]
    return MechanismSpec("dzhanibekov", bodies, joints, [], timestep, input_scaling, gravity)


def get_tippetop(timestep=0.01, input_scaling=None, gravity=-9.81, mass=1.0, radius=0.5, scale=0.2, friction_coefficient=0.4, contact=True,
                 contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/tippetop/mechanism.jl:1-54: two spheres fixed to each other, one contact each"""
    bodies = [BodySpec("sphere1", mass, sphere_inertia(radius, mass)), BodySpec("sphere2", mass * scale ** 3, sphere_inertia(radius * scale, mass * scale ** 3))]
    joints = [Floating("floating_joint", -1, 0), Fixed("fixed_joint", 0, 1, parent_vertex=[0, 0, radius])]
    contacts = [contact_constraint("contact%d" % (i + 1), i, Z_AXIS, friction_coefficient, contact_radius=r, contact_type=contact_type)
                for i, r in enumerate([radius, radius * scale])] if contact else []
    return MechanismSpec("tippetop", bodies, joints, contacts, timestep, input_scaling, gravity)


def get_raiberthopper(timestep=0.05, input_scaling=None, gravity=-9.81, body_mass=4.18, foot_mass=0.52, body_radius=0.1, foot_radius=0.05,
                      springs=(0.0, 0.0), dampers=(0.0, 0.1), friction_coefficient=0.5, contact_foot=True, contact_body=True,
                      contact_type="nonlinear"):
    """DojoEnvironments/src/mechanisms/raiberthopper/mechanism.jl:1-68: a floating sphere with a foot on a damped Prismatic leg"""
    bodies = [BodySpec("body", body_mass, sphere_inertia(body_radius, body_mass)), BodySpec("foot", foot_mass, sphere_inertia(foot_radius, foot_mass))]
    joints = [Floating("floating_base", -1, 0), Prismatic("leg", 0, 1, Z_AXIS)]
    contacts = []
    if contact_foot:
        contacts.append(contact_constraint("foot_contact", 1, Z_AXIS, friction_coefficient, contact_radius=foot_radius, contact_type=contact_type))
    if contact_body:
        contacts.append(contact_constraint("body_contact", 0, Z_AXIS, friction_coefficient, contact_radius=body_radius, contact_type=contact_type))
    spec = MechanismSpec("raiberthopper", bodies, joints, contacts, timestep, input_scaling, gravity)
    _set_per_joint(spec, springs, dampers)
    return spec


# ----------------------------------------------------------------------------------
# URDF mechanisms   src/mechanism/urdf.jl, src/mechanism/constructor.jl:89-109
# ----------------------------------------------------------------------------------
def mechanism_from_urdf_data(name, data, timestep, input_scaling, gravity, floating=True, parse_dampers=True, keep_fixed_joints=True):
    links, ujoints = data["links"], data["joints"]
    lidx = {l["name"]: i for i, l in enumerate(links)}
    children = {j["child"] for j in ujoints}
    roots = [l["name"] for l in links if l["name"] not in children]
    assert len(roots) == 1, "Multiple origins"
    assert floating, "only floating-base URDF mechanisms are built here"
    bodies = []
    for l in links:                                   # parse_link, urdf.jl:171-200: inertia used as given
        ixx, ixy, ixz, iyy, iyz, izz = l["inertia"]
        bodies.append(BodySpec(l["name"], l["mass"], np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])))
    x_in = [np.array(l["xyz"], float) for l in links]
    q_in = [rpy_to_quat(l["rpy"]) for l in links]

    joints = [Floating("floating_base", -1, lidx[roots[0]])]            # urdf.jl:328-331
    raw = [dict(xyz=np.zeros(3), q=np.array([1.0, 0, 0, 0]))]
    for j in ujoints:                                                  # parse_joint / joint_selector, urdf.jl:211-268
        p, c = lidx[j["parent"]], lidx[j["child"]]
        damper = j["damping"] if parse_dampers else 0.0
        if j["type"] in ("revolute", "continuous"):
            joints.append(Revolute(j["name"], p, c, j["axis"], damper=damper))
        elif j["type"] == "prismatic":
            joints.append(Prismatic(j["name"], p, c, j["axis"], damper=damper))
        elif j["type"] == "fixed":
            joints.append(Fixed(j["name"], p, c))
        elif j["type"] == "ball":
            joints.append(Spherical(j["name"], p, c, damper=damper))
        else:
            raise NotImplementedError("URDF joint type " + j["type"])
        raw.append(dict(xyz=np.array(j["xyz"], float), q=rpy_to_quat(j["rpy"])))

    # set_parsed_values!, urdf.jl:420-507: rewrite joints/bodies into COM frames, root -> leaves
    nb = len(bodies)
    xb = [np.zeros(3)] * nb; qb = [np.array([1.0, 0, 0, 0])] * nb     # body world poses at zero configuration
    xjw = {}; qjw = {}                                               # joint world poses, keyed by child body
    parent_joint_of = {j.child: k for k, j in enumerate(joints)}
    order, frontier = [], [0]
    while frontier:                                                  # breadth-first from the floating base
        k = frontier.pop(0); order.append(k)
        frontier += [m for m, j in enumerate(joints) if j.parent == joints[k].child]
    assert len(order) == len(joints), "kinematic loops are not supported"
    for k in order:
        j = joints[k]; c = j.child
        if j.parent < 0:
            xP, qP = np.zeros(3), np.array([1.0, 0, 0, 0]); xpj, qpj = np.zeros(3), np.array([1.0, 0, 0, 0])
        else:
            xP, qP = xb[j.parent], qb[j.parent]; xpj, qpj = xjw[j.parent], qjw[j.parent]
        xjl = vrot(xpj + vrot(raw[k]["xyz"], qpj) - xP, qinv(qP))     # urdf.jl:470
        qjl = qmul(qmul(qinv(qP), qpj), raw[k]["q"])                  # urdf.jl:471
        xjw[c] = xP + vrot(xjl, qP); qjw[c] = qmul(qP, qjl)           # urdf.jl:474-477
        j.orientation_offset = qmul(qjl, q_in[c])                    # urdf.jl:480
        j.vertex_parent = xjl                                        # urdf.jl:483
        j.vertex_child = vrot(-x_in[c], qinv(q_in[c]))               # urdf.jl:484
        qb[c] = qmul(qP, j.orientation_offset)                       # bodies/set.jl:59-70
        xb[c] = xP + vrot(j.vertex_parent, qP) - vrot(j.vertex_child, qb[c])
    if not keep_fixed_joints:
        assert all(j.N != 6 for j in joints), "reduce_fixed_joints: not needed by the BASELINE configs (A1, Atlas-simple have no fixed joints)"
    return MechanismSpec(name, bodies, joints, [], timestep, input_scaling, gravity)


def _load(name):
    with open(os.path.join(_DATA, name + ".json")) as f:
        return json.load(f)


def get_ant(timestep=0.05, input_scaling=None, gravity=-9.81, springs=0.0, dampers=0.0, parse_springs=True, parse_dampers=True,
            joint_limits=None, keep_fixed_joints=True, friction_coefficient=0.5, contact_feet=True, contact_body=True):
    """DojoEnvironments/src/mechanisms/ant/mechanism.jl:1-91"""
    data = _load("ant")
    spec = mechanism_from_urdf_data("ant", data, timestep, input_scaling, gravity, True, parse_dampers, keep_fixed_joints)
    set_springs_dampers(spec, 0.0 if parse_springs else springs, 0.0 if parse_dampers else dampers)
    if joint_limits is None:
        d = np.pi / 180
        joint_limits = {"hip_1": [-30 * d, 30 * d], "ankle_1": [30 * d, 70 * d], "hip_2": [-30 * d, 30 * d], "ankle_2": [-70 * d, -30 * d],
                        "hip_3": [-30 * d, 30 * d], "ankle_3": [-70 * d, -30 * d], "hip_4": [-30 * d, 30 * d], "ankle_4": [30 * d, 70 * d]}
    set_limits(spec, joint_limits)
    radius = {l["name"]: l["radius"] for l in data["links"]}
    if contact_feet:
        names = ["front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot"]
        origins = [[0.2, 0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0], [0.2, -0.2, 0]]
        for n, o in zip(names, origins):
            spec.contacts.append(contact_constraint(n + "_contact", spec.body_index(n), Z_AXIS, friction_coefficient, o, radius[n]))
    if contact_body:
        spec.contacts.append(contact_constraint("torso_contact", spec.body_index("torso"), Z_AXIS, friction_coefficient, np.zeros(3), radius["torso"]))
        names = ["aux_1", "aux_2", "aux_3", "aux_4"]
        origins = [[-0.1, -0.1, 0], [0.1, -0.1, 0], [0.1, 0.1, 0], [-0.1, 0.1, 0]]
        for n, o in zip(names, origins):
            spec.contacts.append(contact_constraint(n + "_contact", spec.body_index(n), Z_AXIS, friction_coefficient, o, radius[n]))
    return spec


def get_quadruped(timestep=0.01, input_scaling=None, gravity=-9.81, springs=0.0, dampers=0.0, parse_springs=True, parse_dampers=True,
                  spring_offset=True, joint_limits=None, keep_fixed_joints=False, friction_coefficient=0.8,
                  contact_feet=True, contact_body=True):
    """DojoEnvironments/src/mechanisms/quadruped/mechanism.jl:1-109"""
    spec = mechanism_from_urdf_data("quadruped", _load("quadruped"), timestep, input_scaling, gravity, True, parse_dampers, keep_fixed_joints)
    set_springs_dampers(spec, 0.0 if parse_springs else springs, 0.0 if parse_dampers else dampers)
    groups = ["FR", "FL", "RR", "RL"]
    if spring_offset:
        for g in groups:
            spec.joints[spec.joint_index(g + "_hip_joint")].rot.spring_offset = np.array([0.0])
            spec.joints[spec.joint_index(g + "_thigh_joint")].rot.spring_offset = np.array([0.9])
            spec.joints[spec.joint_index(g + "_calf_joint")].rot.spring_offset = np.array([-1.425])
    if joint_limits is None:
        joint_limits = {}
        for g in groups:
            joint_limits[g + "_hip_joint"] = [-0.5, 0.5]; joint_limits[g + "_thigh_joint"] = [-0.5, 1.5]; joint_limits[g + "_calf_joint"] = [-2.5, -1.0]
    set_limits(spec, joint_limits)
    if contact_feet:
        for g in groups:
            spec.contacts.append(contact_constraint(g + "_calf_contact", spec.body_index(g + "_calf"), Z_AXIS, friction_coefficient, [-0.006, 0, -0.092], 0.021))
    if contact_body:
        for g, o in zip(groups, [[-0.005, -0.023, -0.16], [-0.005, 0.023, -0.16], [-0.005, -0.023, -0.16], [-0.005, 0.023, -0.16]]):
            spec.contacts.append(contact_constraint(g + "_thigh_contact", spec.body_index(g + "_thigh"), Z_AXIS, friction_coefficient, o, 0.023))
        for g in groups:
            spec.contacts.append(contact_constraint(g + "_hip_contact", spec.body_index(g + "_hip"), Z_AXIS, friction_coefficient, [0, 0.05, 0], 0.05))
    return spec


def get_atlas(timestep=0.01, input_scaling=None, gravity=-9.81, springs=0.0, dampers=0.0, parse_springs=True, parse_dampers=True,
              joint_limits=None, keep_fixed_joints=False, friction_coefficient=0.8, contact_feet=True, contact_body=True):
    """DojoEnvironments/src/mechanisms/atlas/mechanism.jl:1-108 (urdf = :atlas_simple)"""
    spec = mechanism_from_urdf_data("atlas", _load("atlas"), timestep, input_scaling, gravity, True, parse_dampers, keep_fixed_joints)
    set_springs_dampers(spec, 0.0 if parse_springs else springs, 0.0 if parse_dampers else dampers)
    set_limits(spec, joint_limits or {})
    if contact_feet:
        origins = [[-0.08, -0.04, 0.015], [0.12, -0.02, 0.015], [-0.08, 0.04, 0.015], [0.12, 0.02, 0.015]]
        for side in ("l", "r"):
            for n, o in zip(["RR", "FR", "RL", "RR"], origins):
                spec.contacts.append(contact_constraint(side + "_" + n, spec.body_index(side + "_foot"), Z_AXIS, friction_coefficient, o, 0.025))
    if contact_body:
        names = ["l_hand", "r_hand", "l_lleg", "r_lleg", "l_clav", "r_clav", "pelvis", "l_uarm", "r_uarm", "head", "utorso", "utorso"]
        origins = [[0, 0, 0], [0, 0, 0], [0.025, 0, 0.175], [0.025, 0, 0.175], [0, -0.05, -0.075], [0, -0.05, -0.075], [0, 0, 0.05],
                   [0, -0.185, 0], [0, -0.185, 0], [0, 0, 0], [-0.095, 0, 0.25], [-0.095, 0, -0.2]]
        radii = [0.06, 0.06, 0.075, 0.075, 0.11, 0.11, 0.19, 0.085, 0.085, 0.175, 0.15, 0.15]
        for k, (n, o, r) in enumerate(zip(names, origins, radii)):
            spec.contacts.append(contact_constraint("body_contact_%d" % k, spec.body_index(n), Z_AXIS, friction_coefficient, o, r))
    return spec


def get_mechanism(name, **kwargs):
    """DojoEnvironments.get_mechanism(:name; kwargs...)  DojoEnvironments/src/mechanisms.jl"""
    return {"pendulum": get_pendulum, "block": get_block, "ant": get_ant, "quadruped": get_quadruped, "atlas": get_atlas,
            "slider": get_slider, "nslider": get_nslider, "raiberthopper": get_raiberthopper,
            "npendulum": get_npendulum, "snake": get_snake, "twister": get_twister, "sphere": get_sphere,
            "cartpole": get_cartpole, "block2d": get_block2d, "dzhanibekov": get_dzhanibekov, "tippetop": get_tippetop, "limited_chain": get_limited_chain, "fourbar": get_fourbar}[name](**kwargs)


# the five BASELINE.json configurations
def baseline_config(i):
    if i == 1: return get_pendulum()
    if i == 2: return get_block(contact_corners=4)
    if i == 3: return get_ant(contact_body=False)
    if i == 4: return get_quadruped(contact_body=False)
    if i == 5: return get_atlas(contact_body=False)
    raise ValueError(i)
