"""numpy model of the device's elimination on the hard Ant cases: condensed system, supernodal tree elimination leaves -> root
in a fixed pivot order without pivoting (LU form = unpivoted LU of the permuted matrix), and variants."""
import os, sys
import numpy as np, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from oracle import Oracle
from ift_model import exact_solve

def lu_nopivot(A):
    A = A.copy(); n = A.shape[0]
    for k in range(n - 1):
        A[k + 1:, k] /= A[k, k]
        A[k + 1:, k + 1:] -= np.outer(A[k + 1:, k], A[k, k + 1:])
    return A
def lu_solve_nopivot(LU, B):
    n = LU.shape[0]; X = B.copy()
    for k in range(n): X[k + 1:] -= np.outer(LU[k + 1:, k], X[k])
    for k in range(n - 1, -1, -1):
        X[k] /= LU[k, k]; X[:k] -= np.outer(LU[:k, k], X[k])
    return X

def layout(spec):
    """index sets in the oracle's node order [joints; bodies; contacts]"""
    joff = []; off = 0; J = []
    for j in spec.joints:
        N = j.tra.nl + 4 * j.tra.nlim + j.rot.nl + 4 * j.rot.nlim
        # half layout: [s(nlim) s(nlim) γ γ λ(nl)]: tra then rot   (Nb = 2 nlim: s_up s_lo γ_up γ_lo)
        t0 = off; tl = list(range(t0 + 4 * j.tra.nlim, t0 + 4 * j.tra.nlim + j.tra.nl)); tc = list(range(t0, t0 + 4 * j.tra.nlim))
        r0 = t0 + 4 * j.tra.nlim + j.tra.nl
        rl = list(range(r0 + 4 * j.rot.nlim, r0 + 4 * j.rot.nlim + j.rot.nl)); rc = list(range(r0, r0 + 4 * j.rot.nlim))
        J.append(dict(child=j.child, parent=j.parent, lt=tl, lr=rl, cone=tc + rc)); off += N
    nj = off
    Nb = spec.Nb
    boff = [nj + 6 * b for b in range(Nb)]
    coff = nj + 6 * Nb
    return J, nj, boff, coff

if __name__ == "__main__":
    D_ = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "hunt_cfg3_tol0.npz"))
    order = np.argsort(-D_["meta"][:, 3])
    spec = d.baseline_config(3)
    o = Oracle(spec)
    J, nj, boff, coff = layout(spec)
    Nb = spec.Nb
    jof = {j["child"]: j for j in J}
    level = {}
    def lev(b):
        p = jof[b]["parent"]; return 0 if p < 0 else 1 + lev(p)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    orders = {"v lt w lr": "vtwr", "v w lt lr": "vwtr", "lt lr v w": "trvw", "v lt lr w": "vtrw"}
    for ci in order[:ncase]:
        z, u = D_["z"][ci], D_["u"][ci]
        zn, info = o.step(z, u)
        A = o.full_matrix(); Dm = o.data_matrix(); n = A.shape[0]
        X = exact_solve(A, Dm)
        cone = [i for j in J for i in j["cone"]] + list(range(coff, n))
        keep = [i for i in range(n) if i not in set(cone)]
        Acc = A[np.ix_(cone, cone)]; Akc = A[np.ix_(keep, cone)]; Ack = A[np.ix_(cone, keep)]
        Ac = A[np.ix_(keep, keep)] - Akc @ np.linalg.solve(Acc, Ack)
        Dc = Dm[keep] - Akc @ np.linalg.solve(Acc, Dm[cone])
        pos = {g: i for i, g in enumerate(keep)}
        vel = [pos[boff[b] + i] for b in range(Nb) for i in range(6)]
        Xe = X[keep]
        xs = max(1.0, float(np.abs(Xe[vel]).max()))
        res = []
        for name, od in orders.items():
            perm = []
            for b in sorted(range(Nb), key=lambda b: -lev(b)):
                grp = {"v": [pos[boff[b] + i] for i in range(3)], "w": [pos[boff[b] + 3 + i] for i in range(3)], "t": [pos[i] for i in jof[b]["lt"]], "r": [pos[i] for i in jof[b]["lr"]]}
                for ch in od: perm += grp[ch]
            Ap = Ac[np.ix_(perm, perm)]
            LU = lu_nopivot(Ap)
            Xp = lu_solve_nopivot(LU, Dc[perm])
            Xn = np.empty_like(Xp); Xn[perm] = Xp
            e0 = np.abs(Xn[vel] - Xe[vel]).max() / xs
            # one step of refinement against the condensed matrix (fp64 residual)
            R = Dc - Ac @ Xn
            Xr = Xn.copy(); Xr[perm] += lu_solve_nopivot(LU, R[perm])
            e1 = np.abs(Xr[vel] - Xe[vel]).max() / xs
            growth = np.abs(LU).max() / np.abs(Ap).max()
            res.append("%s: %.1e ref %.1e (g %.0e)" % (name, e0, e1, growth))
        print("case %3d it %2d | " % (ci, info["iters"]) + " | ".join(res))
