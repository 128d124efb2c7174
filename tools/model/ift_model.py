"""numpy model of the IFT linear algebra on the hard Ant cases: where do the digits go?
A X = D with A = full_matrix(system) at the solution (uncondensed, node order [joints; bodies; contacts]).
 exact     : LU + iterative refinement with long-double residuals
 cond      : contact blocks condensed (Schur complement in fp64), condensed system solved exactly -> error of the condensation alone
"""
import os, sys
import numpy as np, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from oracle import Oracle

def exact_solve(A, B, rounds=4):
    lu = sla.lu_factor(A)
    X = sla.lu_solve(lu, B).astype(np.longdouble)
    Al = A.astype(np.longdouble); Bl = B.astype(np.longdouble)
    for _ in range(rounds):
        R = Bl - Al @ X
        X = X + sla.lu_solve(lu, R.astype(np.float64)).astype(np.longdouble)
    return X

if __name__ == "__main__":
  D_ = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "hunt_cfg3_tol0.npz"))
  order = np.argsort(-D_["meta"][:, 3])
  spec = d.baseline_config(3)
  o = Oracle(spec)
  ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 4
  for ci in order[:ncase]:
      z, u = D_["z"][ci], D_["u"][ci]
      zn, info = o.step(z, u)
      A = o.full_matrix(); Dm = o.data_matrix()
      n = A.shape[0]
      Ne, Nb, Nc = o.Ne, o.Nb, o.Nc
      nj = n - 6 * Nb - 8 * Nc
      ib = np.arange(nj, nj + 6 * Nb); ic = np.arange(nj + 6 * Nb, n); ij = np.arange(nj)
      X = exact_solve(A, Dm)
      xs = np.abs(X[ib]).max()
      # condensation of the contact blocks
      keep = np.concatenate([ij, ib])
      Acc = A[np.ix_(ic, ic)]; Akc = A[np.ix_(keep, ic)]; Ack = A[np.ix_(ic, keep)]
      Ainv_ck = np.linalg.solve(Acc, Ack)
      Ac = A[np.ix_(keep, keep)] - Akc @ Ainv_ck
      Dc = Dm[keep] - Akc @ np.linalg.solve(Acc, Dm[ic])
      Xc = exact_solve(Ac, Dc)
      e_cond = np.abs(Xc[nj:] - X[ib]).max() / max(1.0, float(xs))
      # plain fp64 partial-pivot LU on the full system
      Xd = np.linalg.solve(A, Dm)
      e_dense = np.abs(Xd[ib] - X[ib]).max() / max(1.0, float(xs))
      Xcd = np.linalg.solve(Ac, Dc)
      e_cd = np.abs(Xcd[nj:] - X[ib]).max() / max(1.0, float(xs))
      print("case %3d it %2d  cond(A) %.1e cond(Ac) %.1e | max|X_v| %.1e | condensation-only err %.2e | dense fp64 LU err %.2e | condensed + fp64 LU %.2e | max gamma/s %.1e"
            % (ci, info["iters"], np.linalg.cond(A), np.linalg.cond(Ac), xs, e_cond, e_dense, e_cd, np.abs(Ainv_ck).max()))
