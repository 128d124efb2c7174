"""Host-side quaternion helpers (numpy, fp64) used only at set-up time: building
mechanisms from URDF data and mapping minimal -> maximal coordinates.

Conventions follow the reference (Hamilton product, q = (s, v1, v2, v3)):
src/orientation/quaternion.jl, rotate.jl, axis_angle.jl, mrp.jl.
"""
import numpy as np


def qmul(a, b):
    s1, x1, y1, z1 = a
    s2, x2, y2, z2 = b
    return np.array([s1 * s2 - x1 * x2 - y1 * y2 - z1 * z2,
                     s1 * x2 + x1 * s2 + y1 * z2 - z1 * y2,
                     s1 * y2 - x1 * z2 + y1 * s2 + z1 * x2,
                     s1 * z2 + x1 * y2 - y1 * x2 + z1 * s2])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qinv(q):
    return qconj(q) / np.dot(q, q)


def vrot(v, q):
    """vector_rotate(v, q) = Vmat(q * (0, v) / q)  (rotate.jl:2-5)"""
    p = np.array([0.0, v[0], v[1], v[2]])
    return qmul(qmul(q, p), qinv(q))[1:]


def rotation_matrix(q):
    return np.stack([vrot(e, q) for e in np.eye(3)], axis=1)


def skew(p):
    return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0.0]])


def rpy_to_quat(rpy):
    """q = RotZ(yaw) * RotY(pitch) * RotX(roll)  (src/mechanism/urdf.jl:48-58)"""
    r, p, y = rpy
    qx = np.array([np.cos(r / 2), np.sin(r / 2), 0, 0])
    qy = np.array([np.cos(p / 2), 0, np.sin(p / 2), 0])
    qz = np.array([np.cos(y / 2), 0, 0, np.sin(y / 2)])
    return qmul(qmul(qz, qy), qx)


def axis_angle_to_quaternion(x):
    """axis_angle.jl:1-11"""
    x = np.asarray(x, dtype=float)
    th = np.linalg.norm(x)
    if th > 0:
        return np.concatenate([[np.cos(0.5 * th)], np.sin(0.5 * th) * x / th])
    return np.array([1.0, 0, 0, 0])


def rotation_vector(q):
    """mrp.jl:61-64: 4 atan(|mrp|) * axis"""
    m = np.asarray(q[1:]) / (q[0] + 1.0)
    mag = np.linalg.norm(m)
    if mag > 0:
        return 4.0 * np.arctan(mag) * m / mag
    return np.zeros(3)


def quaternion_map(w, dt):
    return np.concatenate([[np.sqrt(4.0 / dt ** 2 - np.dot(w, w))], w])


def next_orientation(q2, w, dt):
    """integrator.jl:15"""
    return qmul(q2, quaternion_map(w, dt)) * dt / 2


def angular_velocity(q1, q2, dt):
    """integrator.jl:25-27: 2/dt * V * L(q1)' * q2"""
    return 2.0 / dt * qmul(qconj(q1), q2)[1:]


def orthogonal_rows(axis):
    """src/joints/orthogonal.jl:1-12.  V1, V2 come from an SVD of skew(axis): any
    orthonormal completion is a valid gauge; V3 = normalized axis."""
    axis = np.asarray(axis, dtype=float)
    if np.linalg.norm(axis) > 0:
        axis = axis / np.linalg.norm(axis)
    vt = np.linalg.svd(skew(axis))[2]
    return vt[0].copy(), vt[1].copy(), axis.copy()
