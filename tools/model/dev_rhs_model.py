"""device condensed system + device right-hand sides (emulator dump) solved exactly, against the device's own result and the oracle"""
import sys, os
import numpy as np, scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tree_model import lu_nopivot, lu_solve_nopivot
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from oracle import Oracle
def exact_solve(A, B, rounds=6):
    lu = sla.lu_factor(A)
    X = sla.lu_solve(lu, B).astype(np.longdouble); Al = A.astype(np.longdouble); Bl = B.astype(np.longdouble)
    for _ in range(rounds): X = X + sla.lu_solve(lu, (Bl - Al @ X).astype(np.float64)).astype(np.longdouble)
    return X
rows = {}; rhs = {}
for ln in open(sys.argv[1]):
    t = ln.split()
    if ln.startswith("BLK"):
        k, q, par = int(t[1]), int(t[2]), int(t[3]); v = np.array(t[4:], dtype=np.float64)
        rows[(k, q)] = (par, v[:36].reshape(3, 12), v[36:54].reshape(3, 6), v[54:72].reshape(6, 3), v[72:90].reshape(3, 6))
    elif ln.startswith("QRHS"):
        k = int(t[1]); v = np.array(t[2:], dtype=np.float64); rhs[k] = (v[0], v[1:301], v[301:337])
Nb = 1 + max(k for k, _ in rows); n = 12 * Nb
M = np.zeros((n, n)); parent = {}
for (k, q), (par, S, U, L, D) in rows.items():
    parent[k] = par
    M[12 * k + 3 * q:12 * k + 3 * q + 3, 12 * k:12 * k + 12] += S
    if par >= 0:
        M[12 * k + 3 * q:12 * k + 3 * q + 3, 12 * par:12 * par + 6] += U
        M[12 * par:12 * par + 6, 12 * k + 3 * q:12 * k + 3 * q + 3] += L
        if q < 2: M[12 * par + 3 * q:12 * par + 3 * q + 3, 12 * par:12 * par + 6] += D
spec = d.baseline_config(3)
nlim = {j.child: j.rot.nlim for j in spec.joints}
ROWNV, ROWNJ, RPAR, UOWN, UPAR, SLO, SLP = 0, 36, 72, 144, 180, 288, 294
def blk(a, off, role, cI): return np.array([a[off + role * 18 + i * 6 + cI] for i in range(3)])
B = np.zeros((n, 12 * Nb))
for kk in range(Nb):
    for cI in range(6):
        # configuration column cI (x2: 0..2, phi2: 3..5) -> output column 12 kk + (cI < 3 ? cI : cI + 3)
        col = 12 * kk + (cI if cI < 3 else cI + 3)
        wk, a, own = rhs[kk]
        b = B[:, col]
        for role in range(2): b[12 * kk + 3 * role:12 * kk + 3 * role + 3] += blk(own, 0, role, cI)
        for role in (2, 3): b[12 * kk + 3 * role:12 * kk + 3 * role + 3] += blk(a, ROWNJ, role & 1, cI)
        if nlim[kk] > 0: b[12 * kk + 11] += wk * a[SLO + cI]
        if parent[kk] >= 0:
            for role in range(2): b[12 * parent[kk] + 3 * role:12 * parent[kk] + 3 * role + 3] += blk(a, UOWN, role, cI)
        for c in range(Nb):
            if parent[c] != kk: continue
            wkc, ac, _ = rhs[c]
            for role in range(4): b[12 * c + 3 * role:12 * c + 3 * role + 3] += blk(ac, RPAR, role, cI)
            if nlim[c] > 0: b[12 * c + 11] += wkc * ac[SLP + cI]
            for role in range(2): b[12 * kk + 3 * role:12 * kk + 3 * role + 3] += blk(ac, UPAR, role, cI)
        # velocity column
        colv = 12 * kk + 3 + (cI if cI < 3 else cI + 3)
        for role in range(2): B[12 * kk + 3 * role:12 * kk + 3 * role + 3, colv] += blk(a, ROWNV, role, cI)
X = exact_solve(M, B)
vel = np.array([12 * b + i for b in range(Nb) for i in range(6)])
dev = np.load(sys.argv[2])
# device result: rows 3:6 (dv) and 9:12 (dw) of each body
dvrows = np.array([12 * b + 3 + i for b in range(Nb) for i in range(3)] + [12 * b + 9 + i for b in range(Nb) for i in range(3)])
xrows = np.array([12 * b + i for b in range(Nb) for i in range(3)] + [12 * b + 3 + i for b in range(Nb) for i in range(3)])
D_ = np.load(ROOT + "/gpurun_out/hunt_cfg3_tol0.npz"); order = np.argsort(-D_["meta"][:, 3]); ci = order[int(sys.argv[3])]
o = Oracle(spec); o.step(D_["z"][ci], D_["u"][ci]); gz, gu = o.gradients(0)
Xd = dev["dz"][dvrows]; Xo = gz[dvrows]; Xm = np.array(X[xrows], dtype=np.float64)
print("device vs oracle      : %.3e" % np.abs(Xd - Xo).max())
print("exact(M_dev,b_dev) vs oracle: %.3e" % np.abs(Xm - Xo).max())
print("device vs exact(M_dev,b_dev): %.3e" % np.abs(Xd - Xm).max())
def lev(b): return 0 if parent[b] < 0 else 1 + lev(parent[b])
perm = []
for b in sorted(range(Nb), key=lambda b: -lev(b)):
    for g in (0, 2, 1, 3): perm += [12 * b + 3 * g + i for i in range(3)]
LU = lu_nopivot(M[np.ix_(perm, perm)])
Xp = lu_solve_nopivot(LU, B[perm]); Xn = np.empty_like(Xp); Xn[perm] = Xp
print("numpy unpivoted LU-form on (M_dev,b_dev) vs exact(M_dev,b_dev): %.3e ; vs oracle %.3e" % (np.abs(Xn[xrows] - Xm).max(), np.abs(Xn[xrows] - Xo).max()))
E = np.abs(Xm - Xo); print("worst col (exact dev system vs oracle):", np.unravel_index(E.argmax(), E.shape))

# ---- the device's block algorithm in numpy: per supernode S_k (12x12), U_k (12x6), L_k (6x12), D_k (6x6 onto the parent) ----
Sb = {}; Ub = {}; Lb = {}; Db = {}
for k in range(Nb):
    Sb[k] = M[12 * k:12 * k + 12, 12 * k:12 * k + 12].copy()
for (k, q), (par, S, U, L, D) in rows.items():
    Sb[k][3 * q:3 * q + 3, :] = S          # raw own rows (without the children's Dup contributions, which M already carries)
    Ub.setdefault(k, np.zeros((12, 6)))[3 * q:3 * q + 3] = U
    Lb.setdefault(k, np.zeros((6, 12)))[:, 3 * q:3 * q + 3] = L
    if q < 2: Db.setdefault(k, np.zeros((6, 6)))[3 * q:3 * q + 3] = D
pord = [0, 1, 2, 6, 7, 8, 3, 4, 5, 9, 10, 11]
def mk_solver(S, mode):
    if mode == "inv":
        # Gauss-Jordan explicit inverse without pivoting in the device's order
        A = S[np.ix_(pord, pord)].copy(); n_ = 12; Inv = np.eye(n_)
        for p in range(n_):
            ip = 1.0 / A[p, p]
            for r in range(n_):
                if r == p: continue
                f = A[r, p] * ip; A[r] -= f * A[p]; Inv[r] -= f * Inv[p]
            A[p] *= ip; Inv[p] *= ip
        Sinv = np.empty((12, 12)); Sinv[np.ix_(pord, pord)] = Inv
        return lambda R: Sinv @ R
    LU = lu_nopivot(S[np.ix_(pord, pord)])
    def sol(R):
        X_ = lu_solve_nopivot(LU, R[pord]); O = np.empty_like(X_); O[pord] = X_; return O
    return sol
order_up = sorted(range(Nb), key=lambda b: -lev(b))
for mode in ("inv", "lu"):
    for down in ("y - S^-1(U xp)", "S^-1(r - U xp)"):
        Sk = {k: Sb[k].copy() for k in range(Nb)}; sol = {}; R = {k: B[12 * k:12 * k + 12].copy() for k in range(Nb)}
        Y = {}
        for k in order_up:
            sol[k] = mk_solver(Sk[k], mode)
            Y[k] = sol[k](R[k])
            p_ = parent[k]
            if p_ >= 0:
                Sk[p_][:6, :6] += Db[k] - Lb[k] @ sol[k](Ub[k])
                R[p_][:6] -= Lb[k] @ Y[k]        # (the direct u contributions are already in B)
        Xs = {}
        for k in reversed(order_up):
            p_ = parent[k]
            if p_ < 0: Xs[k] = Y[k]
            elif down.startswith("y"): Xs[k] = Y[k] - sol[k](Ub[k] @ Xs[p_][:6])
            else: Xs[k] = sol[k](R[k] - Ub[k] @ Xs[p_][:6])
        Xall = np.concatenate([Xs[k] for k in range(Nb)])
        print("block algorithm [%s, down: %s]: vs exact %.3e" % (mode, down, np.abs(Xall[xrows] - Xm).max()))
