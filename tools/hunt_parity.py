"""One-off hunt on the GPU: step a large batch of distinct seeded states on the device and on the oracle, and dump the
environment-steps where they disagree (state, gradient, iteration count) for analysis under the emulator.
usage: hunt_parity.py cfg batch steps tol [grad_tol]   -> gpurun_out/hunt_cfg<cfg>.npz + a summary on stdout"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle

cfg, B, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tol = 0.0 if sys.argv[4] == "default" else float(sys.argv[4])        # "default": the reference's rtol 1e-6, btol 1e-4
with_grad = len(sys.argv) <= 5 or sys.argv[5] != "nograd"
refine = os.environ.get("HUNT_REFINE_W")
nth = os.cpu_count() or 8
spec = d.baseline_config(cfg)
opts = d.SolverOptions(rtol=tol, btol=tol) if tol > 0 else d.SolverOptions()
Z, U = d.synthetic_inputs(spec, B)
gm = api.BatchedMechanism(spec, B, dtype="f64", opts=opts)
if refine is not None:
    gm.set_refinement(float(refine))
o = Oracle(spec, opts=opts)
bad = []
DIAG = os.environ.get("HUNT_DIAG") == "1"
if DIAG:
    gm.diagnostics(read=False)
t0 = time.time()
for k in range(steps):
    zn, st, it = gm.step(Z, U, with_gradient=with_grad)
    if with_grad:
        dz, du = gm.gradients()
    Zo, st_o, it_o, dz_o, du_o = o.step_batch(Z, U, with_grad=with_grad, nthreads=nth)
    ok = (st == 0) & (st_o == 0)
    ez = np.abs(zn - Zo).max(axis=1)
    if with_grad:
        eg = np.array([max(np.abs(dz[b] - dz_o[b]).max() / max(1.0, np.abs(dz_o[b]).max()), np.abs(du[b] - du_o[b]).max() / max(1.0, np.abs(du_o[b]).max())) for b in range(B)])
    else:
        eg = np.zeros(B)
    q = lambda a, p: float(np.quantile(a[ok], p)) if ok.any() else float("nan")
    print("step %2d conv gpu %.4f orc %.4f | iters differ %4d | status differ %3d | ez q50 %.1e q99 %.1e max %.1e (>1e-6: %d) | eg q50 %.1e q99 %.1e max %.1e (>1e-6: %d) | %.0fs"
          % (k, (st == 0).mean(), (st_o == 0).mean(), int((it[ok] != it_o[ok]).sum()), int((st != st_o).sum()), q(ez, .5), q(ez, .99), q(ez, 1.0), int((ez[ok] > 1e-6).sum()),
             q(eg, .5), q(eg, .99), q(eg, 1.0), int((eg[ok] > 1e-6).sum()), time.time() - t0), flush=True)
    if DIAG:
        dg = gm.diagnostics()
        w_, g_ = dg[ok, 0], dg[ok, 1]
        hit = eg[ok] > 1e-6
        print("   diag: growth q50 %.1e q90 %.1e q99 %.1e q99.9 %.1e | stiffness q50 %.1e q99 %.1e" % (np.quantile(g_, .5), np.quantile(g_, .9), np.quantile(g_, .99), np.quantile(g_, .999), np.quantile(w_, .5), np.quantile(w_, .99)))
        for thr in (1e4, 1e5, 1e6, 1e7):
            fl = g_ > thr
            print("   growth > %.0e: flagged %.4f of the envs, catches %d of %d with eg > 1e-6; worst uncaught eg %.1e" % (thr, fl.mean(), int((fl & hit).sum()), int(hit.sum()), eg[ok][~fl].max()))
        if hit.any():
            print("   eg>1e-6 envs: growth", np.array2string(np.sort(g_[hit]), precision=1), "stiffness", np.array2string(np.sort(w_[hit]), precision=1))
    score = np.where(ok, np.maximum(ez, eg), 0.0) + np.where(st != st_o, 1.0, 0.0)
    for b in np.argsort(-score)[:8]:
        if score[b] > 1e-7:
            bad.append(dict(step=k, env=int(b), z=Z[b].copy(), u=U[b].copy(), ez=ez[b], eg=eg[b], it=int(it[b]), it_o=int(it_o[b]), st=int(st[b]), st_o=int(st_o[b])))
    Z = Zo
gm.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if bad:
    np.savez(os.path.join(ROOT, "gpurun_out", "hunt_cfg%d_tol%g.npz" % (cfg, tol)), z=np.array([b["z"] for b in bad]), u=np.array([b["u"] for b in bad]),
             meta=np.array([[b["step"], b["env"], b["ez"], b["eg"], b["it"], b["it_o"], b["st"], b["st_o"]] for b in bad]))
print("dumped", len(bad), "cases")
