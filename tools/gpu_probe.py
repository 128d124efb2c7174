"""GPU probe: timing sweeps + parity debugging dumps (writes gpurun_out/)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle

out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
what = sys.argv[1] if len(sys.argv) > 1 else "all"

if what in ("all", "sweep"):
    spec = d.baseline_config(3)
    Z0, U0 = d.synthetic_inputs(spec, 64)
    for B in (256, 1024, 4096, 16384):
        for grad in (False, True):
            Z = np.tile(Z0, (B // 64, 1)); U = np.tile(U0, (B // 64, 1))
            gm = api.BatchedMechanism(spec, B, dtype="f32")
            gm.set_groups(1)        # (last_kernel_ms: one launch of the whole batch per kernel; with groups it reports the last group's, include/dojo_hip.h)
            z = Z.astype(np.float32)
            ms = []
            for k in range(4):
                zn, st, it = gm.step(z, U, with_gradient=grad); ms.append(gm.last_kernel_ms()); z = zn
            print("sweep ant B=%d grad=%d kernel ms %s  steps/s %.0f  iters %.1f ok %.3f" % (B, grad, ["%.2f" % m for m in ms], B / (min(ms) * 1e-3), it.mean(), (st == 0).mean()), flush=True)
            gm.close()

if what in ("all", "parity"):
    spec = d.baseline_config(3)
    opts = d.SolverOptions(rtol=1e-8, btol=1e-8)
    B = 64
    Z, U = d.synthetic_inputs(spec, B)
    gm = api.BatchedMechanism(spec, B, dtype="f64", opts=opts)
    o = Oracle(spec, opts=opts)
    z = Z.copy(); dumps = []
    for k in range(12):
        zg, st, it = gm.step(z, U)
        zo, st_o, it_o, _, _ = o.step_batch(z, U, nthreads=16)
        err = np.abs(zg - zo).max(axis=1)
        w = int(err.argmax())
        print("parity step %d worst env %d err %.3e status gpu %d oracle %d iters gpu %d oracle %d; n(err>1e-6)=%d" % (k, w, err[w], st[w], st_o[w], it[w], it_o[w], (err > 1e-6).sum()), flush=True)
        if err[w] > 1e-6:
            dumps.append(dict(step=k, env=w, z=z[w].tolist(), u=U[w].tolist(), zg=zg[w].tolist(), zo=zo[w].tolist()))
        z = zo
    json.dump(dumps, open(os.path.join(out, "parity_dbg.json"), "w"))
    gm.close()

if what in ("all", "grad"):
    for cfg, pre, tol in ((3, 12, 1e-5), (3, 12, 1e-6), (3, 12, 1e-7), (4, 30, 1e-6), (5, 6, 1e-6), (2, 120, 1e-7)):
        spec = d.baseline_config(cfg)
        opts = d.SolverOptions(rtol=tol, btol=tol)
        B = 32
        Z, U = d.synthetic_inputs(spec, B)
        o = Oracle(spec, opts=opts)
        for _ in range(pre):
            Z, st, it, _, _ = o.step_batch(Z, U, nthreads=16)
        gm = api.BatchedMechanism(spec, B, dtype="f64", opts=opts)
        zn, st, it = gm.step(Z, U, with_gradient=True)
        dz, du = gm.gradients()
        Zo, st_o, it_o, dz_o, du_o = o.step_batch(Z, U, with_grad=True, grad_mode=0, nthreads=16)
        ok = np.nonzero((st == 0) & (st_o == 0))[0]
        ez = np.array([np.abs(dz[b] - dz_o[b]).max() / max(1.0, np.abs(dz_o[b]).max()) for b in ok])
        es = np.array([np.abs(zn[b] - Zo[b]).max() for b in ok])
        print("grad cfg %d pre %d tol %.0e: n_ok %d/%d  state err q50 %.1e q90 %.1e max %.1e | dz rel err q50 %.1e q75 %.1e q90 %.1e max %.1e | scale max %.1e" % (
            cfg, pre, tol, len(ok), B, *np.quantile(es, [0.5, 0.9, 1.0]), *np.quantile(ez, [0.5, 0.75, 0.9, 1.0]), max(np.abs(dz_o[b]).max() for b in ok)), flush=True)
        gm.close()

if what == "phases":
    # needs the instrumented library: DOJO_HIP_LIB=dojo.jl_amd/csrc/libdojo_hip_prof.so (tools/build_variant.sh prof -DDJ_PROF);
    # the step kernel then reports per-phase cycle counts of every wave through the `vel` export
    cfg = int(os.environ.get("PHASES_CFG", "3"))              # 3: Ant (libdojo_hip_prof.so of the default variant); 5: Atlas (VMAXC=4 VQUAD=2 tools/build_variant.sh prof -DDJ_PROF)
    spec = d.baseline_config(cfg)
    Z0, U0 = d.synthetic_inputs(spec, 64)
    B = 4096 if cfg == 3 else 2048
    Z = np.tile(Z0, (B // 64, 1)); U = np.tile(U0, (B // 64, 1))
    gm = api.BatchedMechanism(spec, B, dtype="f32")
    gm.set_groups(1)        # (last_kernel_ms: one launch of the whole batch per kernel; with groups it reports the last group's, include/dojo_hip.h)
    z = Z.astype(np.float32)
    for grad in (False, True):
        for k in range(3):
            zn, st, it = gm.step(z, U, with_gradient=grad); z = zn
        vel, ji, cs = gm.get_solution()
        ph = np.concatenate([vel[:, :6], vel[:, 8:9]], axis=1).astype(np.float64); tot = vel[:, 6].astype(np.float64); iters = vel[:, 7]
        names = ["assemble+condense", "factorize", "solve(corrector)", "line search", "solve(affine)", "grad data", "grad sweeps"]
        print("phases grad=%d kernel %.2f ms; mean cycles/wave %.0f, iters %.2f (max %d)" % (grad, gm.last_kernel_ms(), tot.mean(), iters.mean(), iters.max()))
        for n, v in zip(names, ph.mean(axis=0)):
            print("   %-20s %10.0f cycles  %5.1f%%" % (n, v, 100 * v / tot.mean()))
        print("   %-20s %10.0f cycles  %5.1f%%" % ("other", tot.mean() - ph.mean(axis=0).sum(), 100 * (1 - ph.mean(axis=0).sum() / tot.mean())))
        if grad:
            g = vel[:, 12:21].astype(np.float64).mean(axis=0)
            print("   IFT kernel: total %.0f cycles/wave = linearize %.0f + LU %.0f, data blocks %.0f, sweeps %.0f = up-sweep phase 1 %.0f + root phase %.0f + down-sweep %.0f (of which phase B %.0f)" % (g[4], g[0], g[1], g[2], g[3], g[7], g[8], g[5], g[6]))
        json.dump(dict(total=tot.tolist(), iters=iters.tolist()), open(os.path.join(out, "phase_hist_grad%d.json" % grad), "w"))
    gm.close()

if what == "phases2":
    # needs a -DDJ_PROF2 build (DOJO_HIP_LIB): 24 cycle counters of the Newton loop per wave, through the `vel` export (slots 24..47)
    spec = d.baseline_config(3)
    Z0, U0 = d.synthetic_inputs(spec, 64)
    B = 4096
    Z = np.tile(Z0, (B // 64, 1)); U = np.tile(U0, (B // 64, 1))
    gm = api.BatchedMechanism(spec, B, dtype="f32")
    gm.set_groups(1)
    z = Z.astype(np.float32)
    for k in range(3):
        zn, st, it = gm.step(z, U, with_gradient=False); z = zn
    vel, ji, cs = gm.get_solution()
    c = vel[:, 24:48].astype(np.float64).mean(axis=0)
    names = ["cone rhs", "solve: prologue", "solve: forward sweep", "solve: backward sweep", "solve: recovery", "cone line search", "centering + correction", "snapshot",
             "candidate_step", "evalF: parent + kinematics", "evalF: joint", "evalF: contacts", "evalF: gather + rest", "violations", "trial logic + bookkeeping",
             "evalT: parent + kinematics", "evalT: joint", "evalT: contacts", "evalT: gather + blocks", "condense", "factorize", "loop top", "-", "-"]
    print("phases2 kernel %.2f ms; iters %.2f; sum of counters %.0f cycles/wave" % (gm.last_kernel_ms(), it.mean(), c.sum()))
    for n, v in zip(names, c):
        if n != "-": print("   %-28s %9.0f cycles %5.1f%%  (%.0f per iteration)" % (n, v, 100 * v / c.sum(), v / it.mean()))
    # ... and the IFT kernel's sweeps (slots 48..63, written by dojo_grad_kernel)
    for k in range(2):
        zn, st, it = gm.step(z, U, with_gradient=True); z = zn
    vel, ji, cs = gm.get_solution()
    g = vel[:, 48:64].astype(np.float64).mean(axis=0)
    gn = ["up-sweep: children's messages", "up-sweep: right-hand sides", "up-sweep: forward substitution", "up-sweep: messages + park stores",
          "down-sweep: pop + prefetch", "down-sweep: parent's x, T x", "down-sweep: backward substitution", "down-sweep: posts + output stores"]
    print("IFT sweeps (cycles per wave; the branch phases only)")
    for n, v in zip(gn, g): print("   %-36s %9.0f" % (n, v))
    pn = ["prologue: kinematics + joint_eval<2>", "prologue: slack rows, mlim", "prologue: body / joint rows, joint_impulse_cfg_jac", "prologue: contacts", "prologue: control columns", "prologue: condensation maps (GK)"]
    print("IFT prologue behind lu_prepare (cycles per wave)")
    for n, v in zip(pn, g[8:14]): print("   %-48s %9.0f" % (n, v))
    gm.close()

if what == "stragglers":
    # closed-loop rollout like bench.py (fp64 ABI so the inputs are exact): dump the (z, u) of every environment-step that
    # did not converge, with the iteration histogram, for a CPU-side comparison with the oracle
    spec = d.baseline_config(3)
    B = 4096
    Z0, U0 = d.synthetic_inputs(spec, 64)
    Z = np.tile(Z0, (B // 64, 1)); mask = (np.abs(np.tile(U0, (B // 64, 1))) > 0)
    rng = np.random.Generator(np.random.Philox(key=[20241008, 1000]))
    gm = api.BatchedMechanism(spec, B, dtype="f64")
    gm.set_groups(1)        # (last_kernel_ms: one launch of the whole batch per kernel; with groups it reports the last group's, include/dojo_hip.h)
    z = Z.copy(); dumps = []; hist = np.zeros(52, int)
    for k in range(23):
        U = 0.5 * rng.standard_normal((B, spec.nu)) * mask
        zn, st, it = gm.step(z, U)
        hist += np.bincount(np.clip(it, 0, 51), minlength=52)
        bad = np.nonzero(st != 0)[0]
        for b in bad[:8]:
            dumps.append(dict(step=k, env=int(b), status=int(st[b]), iters=int(it[b]), z=z[b].tolist(), u=U[b].tolist()))
        print("step %d: failed %d, iters mean %.2f max %d, kernel %.2f ms" % (k, len(bad), it.mean(), it.max(), gm.last_kernel_ms()), flush=True)
        z = zn
    print("iteration histogram:", {i: int(h) for i, h in enumerate(hist) if h})
    json.dump(dumps, open(os.path.join(out, "stragglers.json"), "w"))
    gm.close()

if what == "configs":
    # the five BASELINE.json configurations at their own batch sizes (per GPU), kernel time of one step
    for cfg, B, dt_, grad in ((1, 1024, "f64", True), (2, 1024, "f64", False), (3, 4096, "f32", True), (4, 1024, "f32", False), (5, 256, "f32", True)):
        spec = d.baseline_config(cfg)
        Z0, U0 = d.synthetic_inputs(spec, min(B, 64))
        reps = (B + len(Z0) - 1) // len(Z0)
        Z = np.tile(Z0, (reps, 1))[:B]; U = np.tile(U0, (reps, 1))[:B]
        gm = api.BatchedMechanism(spec, B, dtype=dt_)
        z = Z.astype(gm.np_dtype); ms = []
        for k in range(4):
            zn, st, it = gm.step(z, U, with_gradient=grad); ms.append(gm.last_kernel_times()); z = zn
        a, b = min(m[0] for m in ms), min(m[1] for m in ms)
        print("config %d %-10s B=%d %s grad=%d: step kernel %.3f ms, IFT kernel %.3f ms -> %.0f env-steps/s; iters %.1f ok %.3f" % (
            cfg, spec.name, B, dt_, grad, a, b, B / ((a + b) * 1e-3), it.mean(), (st == 0).mean()), flush=True)
        gm.close()

if what == "rollout":
    # simulate!-style forward rollout (dojo_rollout): the library steps the batch as environment groups on internal streams
    spec = d.baseline_config(3)
    B, H = 4096, 40
    Z0, U0 = d.synthetic_inputs(spec, 64)
    Z = np.tile(Z0, (B // 64, 1)); mask = (np.abs(np.tile(U0, (B // 64, 1))) > 0)
    rng = np.random.Generator(np.random.Philox(key=[20241008, 1000]))
    Uh = 0.5 * rng.standard_normal((H, B, spec.nu)) * mask
    gm = api.BatchedMechanism(spec, B, dtype="f32")
    gm.set_groups(1)        # (last_kernel_ms: one launch of the whole batch per kernel; with groups it reports the last group's, include/dojo_hip.h)
    for rep in range(3):
        t0 = time.perf_counter()
        traj, st = gm.rollout(Z.astype(np.float32), Uh.astype(np.float32), record=False)
        el = time.perf_counter() - t0
        print("rollout ant B=%d H=%d: kernel time per step %.2f ms -> %.0f env-steps/s (host wall incl. copies %.1f ms/step); converged %.4f" % (
            B, H, gm.last_kernel_ms(), B / (gm.last_kernel_ms() * 1e-3), 1e3 * el / H, (st == 0).mean()), flush=True)
    gm.close()

if what == "atlas":
    spec = d.baseline_config(5)
    Z0, U0 = d.synthetic_inputs(spec, 64)
    for B in (256, 1024, 2048):
        for grad in (False, True):
            Z = np.tile(Z0, (B // 64, 1)); U = np.tile(U0, (B // 64, 1))
            gm = api.BatchedMechanism(spec, B, dtype="f32")
            z = Z.astype(np.float32); ms = []
            for k in range(3):
                zn, st, it = gm.step(z, U, with_gradient=grad); ms.append(gm.last_kernel_times()); z = zn
            a, b = min(m[0] for m in ms), min(m[1] for m in ms)
            print("atlas B=%d grad=%d: step kernel %.2f ms, IFT kernel %.2f ms -> %.0f env-steps/s; iters %.1f (max %d) ok %.3f" % (B, grad, a, b, B / ((a + b) * 1e-3), it.mean(), it.max(), (st == 0).mean()), flush=True)
            gm.close()

if what == "pcie":
    # host-pointer entry points (dojo_step + dojo_gradients): the rate including the PCIe copies, for DESIGN.md
    spec = d.baseline_config(3)
    B = 4096
    Z0, U0 = d.synthetic_inputs(spec, 64)
    Z = np.tile(Z0, (B // 64, 1)).astype(np.float32); U = np.tile(U0, (B // 64, 1)).astype(np.float32)
    gm = api.BatchedMechanism(spec, B, dtype="f32")
    gm.set_groups(1)        # (last_kernel_ms: one launch of the whole batch per kernel; with groups it reports the last group's, include/dojo_hip.h)
    for grad in (False, True):
        ts = []
        for k in range(4):
            t0 = time.perf_counter()
            zn, st, it = gm.step(Z, U, with_gradient=grad)
            if grad: dz, du = gm.gradients()
            ts.append(time.perf_counter() - t0)
        print("host-pointer path ant B=%d grad=%d: %.1f ms per step (kernels %.2f ms) -> %.0f env-steps/s including PCIe + host transposes" % (B, grad, 1e3 * min(ts), gm.last_kernel_ms(), B / min(ts)), flush=True)
    gm.close()
