"""Random tree mechanisms for the parity tests (test infrastructure): bodies of random mass / inertia hanging on random
Revolute / Spherical / Fixed joints (random axes, vertices, orientation offsets, springs, dampers, limits) below a Floating
or a jointed root, with sphere-half-space contacts on random bodies, and a random closed-joint state."""
import numpy as np
import dojo_amd as d
from dojo_amd import coords
from dojo_amd.mechanisms import Floating, Fixed, Revolute, Spherical, box_inertia, contact_constraint, Z_AXIS


def _rand_quat(rng, scale):
    v = rng.normal(size=3) * scale
    q = np.concatenate([[1.0], v]); return q / np.linalg.norm(q)


def _translational_joint(rng, kind, name, parent, k, pv, cv, qo, tra_limits=False):
    """joints with free translations (src/joints/prototypes.jl: Prismatic, Planar, Cylindrical, FixedOrientation) with
    springs, dampers and spring offsets on the translational half (and on the rotational one where it is free)"""
    axis = rng.normal(size=3)
    nl_t, nl_r = {"prismatic": (2, 3), "planar": (1, 3), "cylindrical": (2, 2), "fixed_orientation": (0, 3)}[kind]
    sp, da = float(rng.choice([0.0, 4.0])), float(rng.choice([0.0, 0.6]))
    tra = d.JointHalfSpec(nl_t, axis=axis, spring=sp, damper=da, spring_offset=rng.uniform(-0.2, 0.2, size=3 - nl_t))
    rot = d.JointHalfSpec(nl_r, axis=axis, spring=sp, damper=da, spring_offset=rng.uniform(-0.2, 0.2, size=3 - nl_r))
    if tra_limits and nl_t == 2 and rng.random() < 0.6:              # limits on the free translational coordinate (Prismatic-type joints)
        tra.limits = (np.array([rng.uniform(-0.3, -0.05)]), np.array([rng.uniform(0.05, 0.3)]))
    return d.JointSpec(name, parent, k, tra, rot, np.array(pv, float), np.array(cv, float), np.array(qo, float))


def random_mechanism(seed, contact_type="nonlinear", nb=None, translational=False, tra_limits=False):
    """translational=True mixes in joints with free translations (their own seeds: the other tests' mechanisms do not change)"""
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(2, 8)) if nb is None else int(nb)
    bodies = []; joints = []; contacts = []
    nchild = [0] * nb
    for k in range(nb):
        dims = rng.uniform(0.1, 0.5, size=3); m = float(rng.uniform(0.3, 2.0))
        bodies.append(d.BodySpec("b%d" % k, m, box_inertia(dims[0], dims[1], dims[2], m)))
    floating = rng.random() < 0.6
    for k in range(nb):
        if k == 0:
            parent = -1
            kind = "floating" if floating else rng.choice(["revolute", "spherical"])
        else:
            cand = [p for p in range(k) if nchild[p] < 4]
            parent = int(rng.choice(cand[-3:] if (nb > 8 and rng.random() < 0.5) else cand)); nchild[parent] += 1
            kind = rng.choice(["revolute", "revolute", "spherical", "fixed"])
            if translational and rng.random() < 0.6:
                kind = rng.choice(["prismatic", "prismatic", "planar", "cylindrical", "fixed_orientation"])
        pv, cv = rng.uniform(-0.3, 0.3, size=3), rng.uniform(-0.3, 0.3, size=3)
        qo = _rand_quat(rng, 0.3)
        name = "j%d" % k
        if kind == "floating":
            joints.append(Floating(name, parent, k))
        elif kind == "fixed":
            joints.append(Fixed(name, parent, k, pv, cv, qo))
        elif kind in ("prismatic", "planar", "cylindrical", "fixed_orientation"):
            joints.append(_translational_joint(rng, kind, name, parent, k, pv, cv, qo, tra_limits))
        elif kind == "spherical":
            joints.append(Spherical(name, parent, k, pv, cv, qo, spring=float(rng.choice([0.0, 2.0])), damper=float(rng.choice([0.0, 0.5]))))
        else:
            lim = ([-0.6], [0.9]) if rng.random() < 0.5 else None
            joints.append(Revolute(name, parent, k, rng.normal(size=3), pv, cv, qo, spring=float(rng.choice([0.0, 3.0])),
                                   damper=float(rng.choice([0.0, 0.3])), rot_spring_offset=np.array([rng.uniform(-0.2, 0.2)]), rot_joint_limits=lim))
    if floating:
        for _ in range(int(rng.integers(0, 4))):
            b = int(rng.integers(0, nb))
            if sum(c.body == b for c in contacts) < 2:
                contacts.append(contact_constraint("c%d" % len(contacts), b, Z_AXIS, float(rng.uniform(0.2, 0.9)), rng.uniform(-0.2, 0.2, size=3),
                                                   float(rng.uniform(0.0, 0.1)), contact_type=contact_type))
    spec = d.MechanismSpec("random%d" % seed, bodies, joints, contacts, 0.01, None, -9.81)
    # a state with closed joints from random minimal coordinates
    x = np.zeros(2 * spec.nu); o = 0
    for j in joints:
        n = j.nu
        if n == 6:
            x[o:o + 6] = np.concatenate([[0, 0, rng.uniform(0.3, 0.8)], rng.normal(size=3) * 0.3]); x[o + 6:o + 12] = rng.normal(size=6) * 0.5
        elif n > 0:
            x[o:o + n] = rng.uniform(-0.5, 0.5, size=n); x[o + n:o + 2 * n] = rng.normal(size=n) * 0.5
        o += 2 * n
    z = coords.minimal_to_maximal(spec, x)
    u = rng.normal(size=spec.nu) * 0.3
    return spec, z, u



def forest_mechanism():
    """several trees in one mechanism (bodies hanging on the origin independently): a two-link pendulum, a free sphere with a floor contact,
    and a single damped link -- three roots, one of them with a branch below it, control batches that span trees"""
    import numpy as np
    from dojo_amd.mechanisms import BodySpec, MechanismSpec, Floating, Revolute, box_inertia, sphere_inertia, contact_constraint
    ll = 0.8
    bodies = [BodySpec("a1", 1.0, box_inertia(0.1, 0.1, ll, 1.0)), BodySpec("a2", 0.7, box_inertia(0.1, 0.1, ll, 0.7)),
              BodySpec("ball", 1.3, sphere_inertia(0.25, 1.3)), BodySpec("b1", 0.9, box_inertia(0.1, 0.1, ll, 0.9))]
    joints = [Revolute("ja1", -1, 0, np.array([1.0, 0, 0]), parent_vertex=np.array([0, 0, 2.0]), child_vertex=np.array([0, 0, ll / 2])),
              Revolute("ja2", 0, 1, np.array([0, 1.0, 0]), parent_vertex=np.array([0, 0, -ll / 2]), child_vertex=np.array([0, 0, ll / 2])),
              Floating("jball", -1, 2),
              Revolute("jb1", -1, 3, np.array([0, 1.0, 0]), parent_vertex=np.array([1.5, 0, 2.0]), child_vertex=np.array([0, 0, ll / 2]))]
    joints[3].tra.damper = joints[3].rot.damper = 0.3
    contacts = [contact_constraint("floor", 2, np.array([0, 0, 1.0]), 0.5, contact_radius=0.25)]
    return MechanismSpec("forest", bodies, joints, contacts, 0.01, None, np.array([0.0, 0.0, -9.81]))


def forest_state(spec, oracle, seed=5, pre=3):
    """a state of forest_mechanism with the ball in contact and everything moving, and a control vector"""
    import numpy as np
    rng = np.random.default_rng(seed)
    z = oracle.minimal_to_maximal(np.zeros(2 * spec.nu)).reshape(4, 13).copy()
    z[2, 0:3] = [0.3, -0.2, 0.2505]; z[2, 3:6] = [0.4, 0.1, -0.2]; z[2, 10:13] = [0.5, -1.0, 0.3]          # the ball just above the floor, moving
    z = z.reshape(-1)
    u = 0.3 * rng.normal(size=spec.nu)
    for _ in range(pre):
        z, info = oracle.step(z, u); assert info["status"] == 0
    return z, u
