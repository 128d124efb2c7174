"""Finite-difference coordinate Jacobians of the ORACLE's minimal <-> maximal maps (test helper, shared by the CPU and GPU tiers).

Restates minimal_to_maximal_jacobian(x) [12Nb x 2nu] (src/gradients/state.jl:128-179) and maximal_to_minimal_jacobian(z)
[2nu x 12Nb] (src/gradients/state.jl:1-56) by central differences of the oracle's maps (pinned by test/minimal.jl restated,
tests/test_oracle_minimal.py), with the attitude convention dq = q (x) (sqrt(1-|phi|^2), phi) of the reference's attitude Jacobians.
"""
import numpy as np
from dojo_amd.quat import qmul, qconj


def fd_coordinate_jacobians(o, xp, zp, h=1e-6):
    """o: oracle.Oracle of the mechanism; xp: minimal state the min->max Jacobian is taken at; zp: maximal state of the max->min one"""
    Nb, nm = o.Nb, 2 * o.nu

    def reduce(zd, z0):                       # maximal difference quotient -> [x v phi w] per body
        out = np.zeros(12 * Nb)
        for b in range(Nb):
            out[12 * b:12 * b + 6] = zd[13 * b:13 * b + 6]
            out[12 * b + 6:12 * b + 9] = qmul(qconj(z0[13 * b + 6:13 * b + 10]), zd[13 * b + 6:13 * b + 10])[1:]
            out[12 * b + 9:12 * b + 12] = zd[13 * b + 10:13 * b + 13]
        return out
    z0 = o.minimal_to_maximal(xp)
    Jm = np.zeros((12 * Nb, nm))
    for j in range(nm):
        e = np.zeros(nm); e[j] = h
        Jm[:, j] = reduce((o.minimal_to_maximal(xp + e) - o.minimal_to_maximal(xp - e)) / (2 * h), z0)
    JM = np.zeros((nm, 12 * Nb))
    for b in range(Nb):
        for i in range(12):
            zs = []
            for sgn in (1.0, -1.0):
                z = zp.copy()
                if i < 6: z[13 * b + i] += sgn * h
                elif i < 9:
                    ph = np.zeros(3); ph[i - 6] = sgn * h
                    z[13 * b + 6:13 * b + 10] = qmul(zp[13 * b + 6:13 * b + 10], np.concatenate([[np.sqrt(1 - h * h)], ph]))
                else: z[13 * b + 10 + (i - 9)] += sgn * h
                zs.append(o.maximal_to_minimal(z))
            JM[:, 12 * b + i] = (zs[0] - zs[1]) / (2 * h)
    return Jm, JM
