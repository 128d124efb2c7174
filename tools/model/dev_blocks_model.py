"""assemble the device's condensed system from the emulator's dump (DJ_DUMP_BLOCKS) and examine the elimination on it"""
import sys, os
import numpy as np, scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tree_model import lu_nopivot, lu_solve_nopivot
def exact_solve(A, B, rounds=5):
    lu = sla.lu_factor(A)
    X = sla.lu_solve(lu, B).astype(np.longdouble); Al = A.astype(np.longdouble); Bl = B.astype(np.longdouble)
    for _ in range(rounds): X = X + sla.lu_solve(lu, (Bl - Al @ X).astype(np.float64)).astype(np.longdouble)
    return X
rows = {}
for ln in open(sys.argv[1]):
    if not ln.startswith("BLK"): continue
    t = ln.split(); k, q, par = int(t[1]), int(t[2]), int(t[3]); v = np.array(t[4:], dtype=np.float64)
    rows[(k, q)] = (par, v[:36].reshape(3, 12), v[36:54].reshape(3, 6), v[54:72].reshape(6, 3), v[72:90].reshape(3, 6))
Nb = 1 + max(k for k, _ in rows)
n = 12 * Nb
M = np.zeros((n, n)); parent = {}
for (k, q), (par, S, U, L, D) in rows.items():
    parent[k] = par
    M[12 * k + 3 * q:12 * k + 3 * q + 3, 12 * k:12 * k + 12] += S
    if par >= 0:
        M[12 * k + 3 * q:12 * k + 3 * q + 3, 12 * par:12 * par + 6] += U
        M[12 * par:12 * par + 6, 12 * k + 3 * q:12 * k + 3 * q + 3] += L
        if q < 2: M[12 * par + 3 * q:12 * par + 3 * q + 3, 12 * par:12 * par + 6] += D
def lev(b): return 0 if parent[b] < 0 else 1 + lev(parent[b])
rng = np.random.default_rng(0)
B = rng.standard_normal((n, 8))
X = exact_solve(M, B)
print("cond(M) %.2e  max|M| %.2e" % (np.linalg.cond(M), np.abs(M).max()))
vel = [12 * b + i for b in range(Nb) for i in range(6)]
for name, od in {"v lt w lr": [0, 2, 1, 3], "v w lt lr": [0, 1, 2, 3]}.items():
    perm = []
    for b in sorted(range(Nb), key=lambda b: -lev(b)):
        for g in od: perm += [12 * b + 3 * g + i for i in range(3)]
    LU = lu_nopivot(M[np.ix_(perm, perm)])
    Xp = lu_solve_nopivot(LU, B[perm]); Xn = np.empty_like(Xp); Xn[perm] = Xp
    print(name, "unpivoted LU-form: rel err (velocity rows) %.2e  all rows %.2e | growth %.1e | min |pivot| %.2e" % (np.abs(Xn[vel] - X[vel]).max() / np.abs(X[vel]).max(), np.abs(Xn - X).max() / np.abs(X).max(), np.abs(LU).max() / np.abs(M).max(), np.abs(np.diag(LU)).min()))
Xd = np.linalg.solve(M, B)
print("partial pivoting dense: %.2e" % (np.abs(Xd[vel] - X[vel]).max() / np.abs(X[vel]).max()))
np.save("/tmp/M_dev.npy", M)
