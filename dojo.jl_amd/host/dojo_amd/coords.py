"""Minimal <-> maximal coordinate maps on the host (numpy, set-up time) and the nominal /
synthetic initial states of the BASELINE configs.

minimal_to_maximal : src/mechanism/state.jl:9-22 + src/joints/minimal.jl:160-232
maximal_to_minimal : src/mechanism/state.jl:44-66 + translational/minimal.jl:56-113,
                     rotational/minimal.jl:62-118
initialize_*       : DojoEnvironments/src/mechanisms/*/mechanism.jl `initialize_<name>!`
"""
import numpy as np
from .quat import (qmul, qinv, vrot, axis_angle_to_quaternion, rotation_vector, next_orientation, angular_velocity)


def _root_to_leaves(spec):
    """joints of the spanning tree, root to leaves; loop-closing joints are left out (set_minimal_coordinates!(...; exclude_ids = [loop_joint_id]),
    DojoEnvironments/src/mechanisms/fourbar/mechanism.jl:44-50: their coordinates follow from the tree's)"""
    order, frontier = [], [k for k, j in enumerate(spec.joints) if j.parent < 0 and not getattr(j, "loop", False)]
    while frontier:
        k = frontier.pop(0); order.append(k)
        frontier += [m for m, j in enumerate(spec.joints) if j.parent == spec.joints[k].child and not getattr(j, "loop", False)]
    return order


def minimal_dimension(spec):
    return 2 * spec.nu


def minimal_to_maximal(spec, x):
    """x: per joint [Δx; Δθ; Δv; Δω] (2·nu_j each, joints in mechanism order) -> z (13·Nb)"""
    dt = spec.timestep
    x = np.asarray(x, dtype=float)
    offs, o = [], 0
    for j in spec.joints:
        offs.append(o); o += 2 * j.nu
    z = np.zeros(13 * spec.Nb)
    origin = (np.zeros(3), np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3))
    for k in _root_to_leaves(spec):
        j = spec.joints[k]
        nu, nt = j.nu, j.tra.nu
        xm = x[offs[k]:offs[k] + 2 * nu]
        dx, dth, dv, dw = xm[:nt], xm[nt:nu], xm[nu:nu + nt], xm[nu + nt:]
        if j.parent < 0:
            xa, va, qa, wa = origin
        else:
            p = z[13 * j.parent:13 * j.parent + 13]
            xa, va, qa, wa = p[0:3], p[3:6], p[6:10], p[10:13]
        _, At = j.tra.masks(); _, Ar = j.rot.masks()
        pa, pb, qoff = j.vertex_parent, j.vertex_child, j.orientation_offset
        # positions (minimal.jl:205-207)
        dq = axis_angle_to_quaternion(Ar.T @ dth) if Ar.shape[0] else np.array([1.0, 0, 0, 0])
        qb = qmul(qmul(qa, qoff), dq)
        xb = xa + vrot(pa + (At.T @ dx if At.shape[0] else 0.0), qa) - vrot(pb, qb)
        # previous configuration (minimal.jl:210-218)
        xa1 = xa - va * dt
        qa1 = next_orientation(qa, -wa, dt)
        dx1 = dx - dv * dt
        dq1 = qmul(dq, qinv(axis_angle_to_quaternion(Ar.T @ (dw * dt)))) if Ar.shape[0] else dq
        qb1 = qmul(qmul(qa1, qoff), dq1)
        xb1 = xa1 + vrot(pa + (At.T @ dx1 if At.shape[0] else 0.0), qa1) - vrot(pb, qb1)
        vb = (xb - xb1) / dt
        wb = angular_velocity(qb1, qb, dt)
        z[13 * j.child:13 * j.child + 13] = np.concatenate([xb, vb, qb, wb])
    return z


def maximal_to_minimal(spec, z):
    dt = spec.timestep
    out = []
    for j in spec.joints:
        c = z[13 * j.child:13 * j.child + 13]
        xb, vb, qb, wb = c[0:3], c[3:6], c[6:10], c[10:13]
        if j.parent < 0:
            xa, va, qa, wa = np.zeros(3), np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
        else:
            p = z[13 * j.parent:13 * j.parent + 13]
            xa, va, qa, wa = p[0:3], p[3:6], p[6:10], p[10:13]
        _, At = j.tra.masks(); _, Ar = j.rot.masks()
        pa, pb, qoff = j.vertex_parent, j.vertex_child, j.orientation_offset

        def disp(xa_, qa_, xb_, qb_):
            return vrot(xb_ + vrot(pb, qb_) - (xa_ + vrot(pa, qa_)), qinv(qa_))
        xa1, xb1 = xa - va * dt, xb - vb * dt
        qa1, qb1 = next_orientation(qa, -wa, dt), next_orientation(qb, -wb, dt)
        q = qmul(qmul(qinv(qoff), qinv(qa)), qb)
        q1 = qmul(qmul(qinv(qoff), qinv(qa1)), qb1)
        ct = At @ disp(xa, qa, xb, qb)
        cr = Ar @ rotation_vector(q)
        vt = At @ (disp(xa, qa, xb, qb) - disp(xa1, qa1, xb1, qb1)) / dt
        vr = Ar @ rotation_vector(qmul(qinv(q1), q)) / dt
        out += [ct, cr, vt, vr]
    return np.concatenate(out) if out else np.zeros(0)


def minimal_state_dict(spec, coords=None, vels=None):
    """Build a minimal state vector from {joint_name: coordinates} / {joint_name: velocities}."""
    x = np.zeros(2 * spec.nu)
    o = 0
    for j in spec.joints:
        if coords and j.name in coords:
            x[o:o + j.nu] = coords[j.name]
        if vels and j.name in vels:
            x[o + j.nu:o + 2 * j.nu] = vels[j.name]
        o += 2 * j.nu
    return x


# ----------------------------------------------------------------------------------
# nominal states
# ----------------------------------------------------------------------------------
def nominal_minimal(spec, **kw):
    n = spec.name
    if n == "pendulum":                      # initialize_pendulum!: angle = π/4
        return minimal_state_dict(spec, {"joint": [kw.get("angle", np.pi / 4)]}, {"joint": [kw.get("angular_velocity", 0.0)]})
    if n == "block":                         # initialize_block!: position [0,0,1] + edge/2 (+ radius)
        pos = np.array(kw.get("position", [0, 0, 1.0]), float)
        edge = (12.0 * spec.bodies[0].inertia[0, 0] / spec.bodies[0].mass / 2.0) ** 0.5
        off = spec.contacts[0].radius if spec.contacts else 0.0
        pos = pos + np.array([0, 0, edge / 2 + off])
        x = minimal_state_dict(spec, {"floating_base": np.concatenate([pos, kw.get("rotation_vector", np.zeros(3))])})
        return x
    if n == "ant":                           # initialize_ant!: z = 0.5, ankles ±0.25π
        a = kw.get("ankle_angle", 0.25) * np.pi
        c = {"floating_base": [0, 0, 0.5, 0, 0, 0]}
        for i in (1, 4): c["hip_%d" % i] = [0.0]; c["ankle_%d" % i] = [a]
        for i in (2, 3): c["hip_%d" % i] = [0.0]; c["ankle_%d" % i] = [-a]
        return minimal_state_dict(spec, c)
    if n == "quadruped":                     # initialize_quadruped! (quadruped/mechanism.jl:111-126): z = 0.43 + body_position, thigh π/4, calf −π/2
        bp = np.array(kw.get("body_position", [0, 0, 0]), float) + np.array([0, 0, 0.43])
        c = {"floating_base": [bp[0], bp[1], bp[2], 0, 0, 0]}
        for g in ("FR", "FL", "RR", "RL"):
            c[g + "_hip_joint"] = [kw.get("hip_angle", 0.0)]; c[g + "_thigh_joint"] = [kw.get("thigh_angle", np.pi / 4)]
            c[g + "_calf_joint"] = [kw.get("calf_angle", -np.pi / 2)]
        return minimal_state_dict(spec, c)
    if n == "atlas":                         # initialize_atlas!: z = 0.9385
        return minimal_state_dict(spec, {"floating_base": [0, 0, 0.9385, 0, 0, 0]})
    if n in ("slider", "nslider"):           # initialize_slider! / initialize_nslider!: zero coordinates (+ optional position / velocity of the first joint)
        x = np.zeros(2 * spec.nu)
        x[0] = kw.get("position", 0.0); x[1] = kw.get("velocity", 0.0)
        return x
    if n in ("snake", "twister"):            # initialize_snake! / initialize_twister!: zero coordinates, base pose from the keywords
        bp = np.array(kw.get("base_position", [0, 0, 0.3]), float)
        return minimal_state_dict(spec, {"floating_base": np.concatenate([bp, kw.get("base_rotation_vector", np.zeros(3))])},
                                  {"floating_base": np.concatenate([kw.get("base_linear_velocity", np.zeros(3)), kw.get("base_angular_velocity", np.zeros(3))])})
    if n == "npendulum":                     # initialize_npendulum!: base angle π/4 on the first joint's rotational coordinates
        x = np.zeros(2 * spec.nu)
        j0 = spec.joints[0]
        if j0.rot.nu > 0:
            x[j0.tra.nu] = kw.get("base_angle", np.pi / 4)
        return x
    if n == "sphere":                        # initialize_sphere!: position [0,0,1] + radius
        r = spec.contacts[0].radius if spec.contacts else 0.5
        pos = np.array(kw.get("position", [0, 0, 1.0]), float) + np.array([0, 0, r])
        return minimal_state_dict(spec, {"floating_base": np.concatenate([pos, np.zeros(3)])},
                                  {"floating_base": np.concatenate([kw.get("velocity", [1.0, 0, 0]), kw.get("angular_velocity", np.zeros(3))])})
    if n == "cartpole":                      # initialize_cartpole!: position 0, pole at π/4
        return minimal_state_dict(spec, {"cart_joint": [kw.get("position", 0.0)], "pole_joint": [kw.get("orientation", np.pi / 4)]})
    if n == "block2d":                       # initialize_block2d!: (y, z) = position + half the edge, rotation about x
        edge = (12.0 * spec.bodies[0].inertia[0, 0] / spec.bodies[0].mass / 2.0) ** 0.5
        off = spec.contacts[0].radius if spec.contacts else 0.0
        pos = np.array(kw.get("position", [0.0, 1.0]), float)
        x = np.zeros(2 * spec.nu)
        x[0:3] = [pos[0], pos[1] + edge / 2 + off, kw.get("orientation", 0.0)]
        x[3:6] = list(kw.get("velocity", [0.0, 0.0])) + [kw.get("angular_velocity", 0.5)]
        return x
    if n == "dzhanibekov":                   # initialize_dzhanibekov!: at z = 1 spinning about x with a small perturbation
        return minimal_state_dict(spec, {"floating": [0, 0, 1.0, 0, 0, 0]}, {"floating": list(kw.get("linear_velocity", [0, 0, 0])) + list(kw.get("angular_velocity", [10.0, 0.01, 0.0]))})
    if n == "tippetop":                      # initialize_tippetop!: resting height, spinning about z
        r = spec.contacts[0].radius if spec.contacts else 0.5
        return minimal_state_dict(spec, {"floating_joint": [0, 0, r + kw.get("height", 0.0), 0, 0, 0]},
                                  {"floating_joint": list(kw.get("body_linear_velocity", [0, 0.1, 0])) + list(kw.get("body_angular_velocity", [0.1, 0.1, 50.0]))})
    if n == "raiberthopper":                 # initialize_raiberthopper! (raiberthopper/mechanism.jl:70-82): body above the foot, leg_length = 0.5
        leg = kw.get("leg_length", 0.5)
        bp = np.array(kw.get("body_position", [0, 0, 0]), float) + np.array([0, 0, leg + spec.contacts[0].radius if spec.contacts else leg + 0.05])
        return minimal_state_dict(spec, {"floating_base": [bp[0], bp[1], bp[2], 0, 0, 0], "leg": [-leg]})
    if n == "fourbar":                       # initialize_fourbar! (fourbar/mechanism.jl:38-52): base_angle = inner_angle = π/4, the loop joint left out
        ba, ia = kw.get("base_angle", np.pi / 4), kw.get("inner_angle", np.pi / 4)
        return minimal_state_dict(spec, {"jointb1": [ba + ia], "jointb3": [ba - ia], "joint12": [-2 * ia], "joint34": [2 * ia]})
    if n.startswith("limited_"):             # mechanisms.get_limited_chain (tests of multi-coordinate joint limits): zero coordinates
        return np.zeros(2 * spec.nu)
    raise ValueError(n)


def initialize(spec, **kw):
    """-> nominal maximal state z (13·Nb)"""
    if spec.name == "block" and ("velocity" in kw or "angular_velocity" in kw or "orientation" in kw):
        # initialize_block! sets maximal velocities directly (block/mechanism.jl:72-95)
        z = minimal_to_maximal(spec, nominal_minimal(spec, **kw))
        z[3:6] = kw.get("velocity", np.zeros(3)); z[10:13] = kw.get("angular_velocity", np.zeros(3))
        if "orientation" in kw: z[6:10] = kw["orientation"]
        return z
    return minimal_to_maximal(spec, nominal_minimal(spec, **kw))


# ----------------------------------------------------------------------------------
# synthetic inputs (BASELINE.md §3 / SURVEY.md §8d): identical bytes for CPU and GPU
# ----------------------------------------------------------------------------------
_CONFIG_ID = {"pendulum": 1, "block": 2, "ant": 3, "quadruped": 4, "atlas": 5}


# Perturbation sizes: base height above the nominal pose U(0, height), base rotation vector N(0, rot_sigma), joint coordinates nominal
# +- joint_range (uniformly inside the limits where a joint has them), minimal velocities N(0, vel_sigma).
#   distribution = "baseline": BASELINE.md section 3's perturbation, the same for every mechanism -- what bench.py measures by default.
#   distribution = "standing" (opt-in): Atlas stands on two feet with four coplanar contacts each; thrown about like the Ant it lands on
#       foot edges and tumbles, and 5-9 % of such states stall the reference's solver at max_iter (the oracle as well).  Here its states are
#       drawn around the reference's own initialize_atlas! pose (DojoEnvironments/src/mechanisms/atlas/mechanism.jl:110-118: standing,
#       z = 0.9385) with a small drop, tilt and joint scatter: the first ~10 closed-loop steps are the landing (the oracle still stalls
#       on 2-8 % of them), after which every solve converges in 8-9 iterations under random torques.  The other mechanisms are unchanged.
# distribution = None takes DOJO_SYNTH_DISTRIBUTION from the environment (tests/conftest.py sets "standing": the Atlas gates of the parity
# tests were tuned there; the full-batch test runs the BASELINE distribution too), else "baseline".
_SYNTH_GENERAL = dict(height=0.3, rot_sigma=0.1, vel_sigma=0.5, joint_range=0.2)
_SYNTH_STANDING = {"atlas": dict(height=0.02, rot_sigma=0.02, vel_sigma=0.1, joint_range=0.05)}
_SYNTH_DEFAULTS = _SYNTH_STANDING          # (the name round 5 used)


def synthetic_distribution(distribution=None):
    import os
    dist = distribution or os.environ.get("DOJO_SYNTH_DISTRIBUTION", "baseline")
    if dist not in ("baseline", "standing"):
        raise ValueError("synthetic_inputs: distribution must be 'baseline' or 'standing', not %r" % (dist,))
    return dist


def synthetic_inputs(spec, batch, seed=20241008, height=None, rot_sigma=None, vel_sigma=None, u_sigma=0.5, joint_range=None, distribution=None, offset=0):
    """Perturb the nominal state in minimal coordinates and map to maximal with the host FK so
    joints stay closed.  Counter-based RNG (Philox) keyed by (seed, config, field); row b of every
    field depends only on b, so a smaller batch is a prefix of a larger one -- and `offset` gives rows offset .. offset + batch - 1
    of that one sequence (a rank's contiguous shard of a batch all ranks agree on: bench.py --batch-total)."""
    if offset:
        return _synthetic_rows(spec, (int(offset), int(offset) + int(batch)), seed, height, rot_sigma, vel_sigma, u_sigma, joint_range, distribution)
    return _synthetic_rows(spec, (0, int(batch)), seed, height, rot_sigma, vel_sigma, u_sigma, joint_range, distribution)


def _synthetic_rows(spec, rows, seed, height, rot_sigma, vel_sigma, u_sigma, joint_range, distribution):
    lo_, hi_ = rows
    batch = hi_
    dflt = dict(_SYNTH_GENERAL, **(_SYNTH_STANDING.get(spec.name, {}) if synthetic_distribution(distribution) == "standing" else {}))
    cid = _CONFIG_ID.get(spec.name, 9)
    height = dflt["height"] if height is None else height; rot_sigma = dflt["rot_sigma"] if rot_sigma is None else rot_sigma
    vel_sigma = dflt["vel_sigma"] if vel_sigma is None else vel_sigma; joint_range = dflt["joint_range"] if joint_range is None else joint_range

    def rng(field):
        return np.random.Generator(np.random.Philox(key=[seed, cid * 16 + field]))
    x0 = nominal_minimal(spec)
    nu = spec.nu
    U_h = rng(0).uniform(0.0, height, size=(batch,))
    N_r = rng(1).normal(0.0, rot_sigma, size=(batch, 3))
    U_j = rng(2).uniform(0.0, 1.0, size=(batch, max(nu, 1)))
    N_v = rng(3).normal(0.0, vel_sigma, size=(batch, max(nu, 1)))
    N_u = rng(4).normal(0.0, u_sigma, size=(batch, max(nu, 1)))
    Z = np.zeros((batch, 13 * spec.Nb)); U = np.zeros((batch, nu))
    for b in range(lo_, batch):
        x = x0.copy(); o = 0; iu = 0
        for j in spec.joints:
            n = j.nu
            if j.N == 0 and j.parent < 0:                       # floating base
                x[o + 2] += U_h[b]; x[o + 3:o + 6] = N_r[b]
                x[o + n:o + 2 * n] = N_v[b, iu:iu + n]
            else:
                for half, lo in ((j.tra, 0), (j.rot, j.tra.nu)):
                    for k in range(half.nu):
                        if half.limits is not None and k < half.nlim:
                            a, c = half.limits[0][k], half.limits[1][k]
                            x[o + lo + k] = a + (c - a) * (0.05 + 0.9 * U_j[b, iu + lo + k])   # inside the limits
                        else:
                            x[o + lo + k] += (2.0 * joint_range) * U_j[b, iu + lo + k] - joint_range      # (0.4 U - 0.2 at the general range, bit for bit as before)
                x[o + n:o + 2 * n] = N_v[b, iu:iu + n]
                U[b, iu:iu + n] = N_u[b, iu:iu + n]
            o += 2 * n; iu += n
        Z[b] = minimal_to_maximal(spec, x)
    return Z[lo_:], U[lo_:]


def fp32_abi_state(Z):
    """The fp64 state an fp32-ABI buffer stands for (include/dojo_hip.h): values rounded to fp32, every quaternion
    renormalized -- what the CPU oracle is given when it is compared with an fp32-ABI handle."""
    Zr = np.asarray(Z, dtype=np.float32).astype(np.float64)
    shp = Zr.shape
    Zb = Zr.reshape(-1, 13)
    Zb[:, 6:10] /= np.linalg.norm(Zb[:, 6:10], axis=1, keepdims=True)
    return Zb.reshape(shp)
