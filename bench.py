#!/usr/bin/env python3
"""bench.py -- differentiable env-steps/s of the batched contact-implicit step on MI355X.

One "step" = one pass of the hot path (forward Mehrotra solve + IFT gradients) over a batch of
B = 4096 Ant environments per GPU (BASELINE.json configs[2], the config the metric is quoted on),
closed loop: z_{k+1} = step(z_k, u_k) with synthetic controls, all buffers resident in HBM.
N > 1: one process per GPU (torch.distributed / RCCL), the batch is sharded (weak scaling: B per
GPU is fixed) with no data-path collective; the final trajectories are all-gathered once per
rollout chunk over RCCL inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector fp64: half the 157.3 TFLOP/s fp32 vector peak of the guide (32 flop/clk/SIMD; tools/ubench/valu_rate.hip measures one fp64 FMA per 4.5 cycles and wave)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (3 = Ant)")
    ap.add_argument("--io-dtype", default="f32")
    ap.add_argument("--no-grad", action="store_true")
    ap.add_argument("--chunks", type=int, default=16, help="the per-GPU batch is stepped as this many independent groups of environments, "
                    "each on its own HIP stream, so that the few environments that run into max_iter do not idle the GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--refine", type=float, default=None, help="stiffness threshold of the solve refinement (dojo_set_refinement); default: the library's")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1: nccl (= RCCL, production) or gloo "
                    "(plumbing check of the N > 1 path on a box with fewer GPUs than ranks: ranks share devices, the gather goes through the host)")
    args = ap.parse_args()

    # one hardware queue per environment group: ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware
    # queues, and streams that share a queue serialize (measured: 4 groups on the default 4 queues run at 0.6x, not 1.1x)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, args.chunks + 8)))
    import numpy as np
    import torch
    import dojo_amd as d
    from dojo_amd import api

    from dojo_amd import distributed as D
    import torch.distributed as dist
    rank, world, local = D.init_from_env(backend=args.backend)    # "nccl" IS RCCL on ROCm
    if args.backend != "nccl":
        local = local % torch.cuda.device_count()                 # plumbing check only: ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    spec = d.baseline_config(args.config)
    B, K, W = args.batch, args.steps, args.warmup
    tdt = torch.float32 if args.io_dtype == "f32" else torch.float64
    w = 4 if args.io_dtype == "f32" else 8
    # synthetic inputs: 64 seeded environments per rank tiled over the batch, distinct seeds per rank
    Z0, U0 = d.synthetic_inputs(spec, 64, seed=20241008 + rank)
    reps = (B + 63) // 64
    z = torch.tensor(np.tile(Z0, (reps, 1))[:B], dtype=tdt, device=dev).contiguous()
    rng = np.random.Generator(np.random.Philox(key=[20241008, 1000 + rank]))
    Uall = torch.tensor(0.5 * rng.standard_normal((K + W, B, spec.nu)) * (np.abs(np.tile(U0, (reps, 1))[:B]) > 0), dtype=tdt, device=dev).contiguous()
    zn = torch.empty_like(z)
    status = torch.empty(B, dtype=torch.int32, device=dev); iters = torch.empty(B, dtype=torch.int32, device=dev)
    grad = not args.no_grad
    dz = torch.empty((B, spec.nx, spec.nx), dtype=tdt, device=dev) if grad else None
    du = torch.empty((B, max(spec.nu, 1), spec.nx), dtype=tdt, device=dev) if grad else None

    # The batch is stepped as `chunks` independent groups of environments, one handle + one HIP stream each.  Environments
    # are independent, so no group ever waits for another: while the rare environment that runs into
    # max_iter = 50 (5x the mean iteration count, one wavefront) finishes in one group, the other groups' kernels fill the GPU.
    NCH = max(1, min(args.chunks, B // 64))
    bounds = [(c * B) // NCH for c in range(NCH + 1)]
    lib = api.lib()
    groups = []
    for c in range(NCH):
        lo, hi = bounds[c], bounds[c + 1]
        groups.append({"lo": lo, "hi": hi, "gm": api.BatchedMechanism(spec, hi - lo, dtype=args.io_dtype, device=local),
                       "stream": torch.cuda.Stream(device=dev)})
        if args.refine is not None:
            groups[-1]["gm"].set_refinement(args.refine)

    def ptr(t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    def one_step(k):
        nonlocal z, zn
        for g in groups:
            lo, hi = g["lo"], g["hi"]
            api._chk(lib.dojo_step_dev(g["gm"].h, ptr(z[lo:hi]), ptr(Uall[k][lo:hi]), ptr(zn[lo:hi]), ptr(status[lo:hi]), ptr(iters[lo:hi]),
                                       ptr(dz[lo:hi]) if grad else None, ptr(du[lo:hi]) if grad else None, C.c_void_p(g["stream"].cuda_stream)))
        z, zn = zn, z

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()                     # inputs were produced on the default stream
    for k in range(W):
        one_step(k)
    barrier()
    for g in groups:
        g["gm"].kernel_time_totals(reset=True)   # kernel timing: hipEvents on each launch stream, accumulated without host waits
    t0 = time.perf_counter()
    for k in range(W, W + K):
        one_step(k)
    for g in groups:
        torch.cuda.current_stream().wait_stream(g["stream"])
    if args.backend == "nccl":
        z_all = D.all_gather_states(z, world)  # all-gather of the final states over RCCL/xGMI, once per rollout chunk
    else:
        torch.cuda.synchronize(); z_all = D.all_gather_states(z.cpu(), world)
    barrier()
    el = D.max_over_ranks(time.perf_counter() - t0, world, device=dev if args.backend == "nccl" else "cpu")
    tot = [g["gm"].kernel_time_totals() for g in groups]
    kernel_ms = [(a / n, b / n) for a, b, n in tot if n > 0]
    ok_frac = float((status == 0).float().mean().item())
    mean_iters = float(iters.float().mean().item())

    if rank == 0:
        nb, nu = spec.Nb, spec.nu
        bytes_fwd = (26 * nb + nu) * w + 8
        bytes_grad = 12 * nb * (12 * nb + nu) * w if grad else 0
        # one step = two launches: dojo_step_kernel (Newton loop) and dojo_grad_kernel (IFT back-solves).  Algorithmic bytes per
        # launch: the step kernel reads z,u and writes z_next,status,iters; the IFT kernel writes dz,du.
        step_ms = sum(a for a, _ in kernel_ms) / len(kernel_ms)
        ift_ms = sum(b for _, b in kernel_ms) / len(kernel_ms)
        traffic = measured_traffic()

        def roof(kernel, ms, nbytes):
            ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            t = traffic.get(kernel)
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": t["bytes_per_launch"] * (B // NCH) / t["envs_per_launch"] if t else None,     # scaled to this launch size
                    "traffic_source": t["source"] if t else None,
                    "kernel": kernel, "avg_kernel_ms": ms, "algorithmic_bytes_per_launch": nbytes,
                    "executed_fp64_flops_per_launch": (t["fp64_flops_per_launch"] * (B // NCH) / t["envs_per_launch"]) if (t and t.get("fp64_flops_per_launch")) else None}
        Bl = B // NCH                                                 # environments per launch
        util = measured_valu_utilization()

        def with_valu(r):
            if r is not None and r["kernel"] in util:
                r["valu_active_frac"] = util[r["kernel"]]["valu_active_frac"]      # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the committed PMC pass
                r["valu_source"] = util[r["kernel"]]["source"]
            return r
        r_step = roof("dojo_step_kernel", step_ms, bytes_fwd * Bl)
        r_ift = roof("dojo_grad_kernel", ift_ms, bytes_grad * Bl) if grad else None
        r_step, r_ift = with_valu(r_step), with_valu(r_ift)
        dominant, other = (r_step, r_ift) if (r_ift is None or step_ms >= ift_ms) else (r_ift, r_step)
        dominant["note"] = ("VALU-issue-bound fp64 lane program (DESIGN.md §8; tools/ubench): the KKT systems never leave registers/LDS, "
                            "so the algorithmic HBM bytes are tiny and frac against HBM is reported only because the contract asks for it")
        # the bound that matters for this path: executed fp64 vector FLOPs (PMC pass SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 of the committed
        # profile, both kernels, all environment groups of this GPU) over the wall-clock step, against the vector-fp64 peak.  (The per-launch
        # durations above overlap across the groups' streams, so a per-launch rate would under-state the device by the number of groups.)
        fl = [r.get("executed_fp64_flops_per_launch") for r in (r_step, r_ift) if r is not None]
        if all(f is not None for f in fl):
            tot = sum(fl) * NCH
            ach = tot / (el / K) / 1e12
            dominant["valu_fp64"] = {"executed_flops_per_step": tot, "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "scope": "both kernels, all %d groups, wall-clock step" % NCH}
        res = {
            "metric": "differentiable env-steps/sec (fwd+grad) at batch=4096; grad inf-err vs CPU" if grad else "env-steps/sec (fwd only)",
            "value": world * B * K / el, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * el / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (("BASELINE.json configs[2]: Ant (13 bodies as built by the reference, 8 revolute+limits, 4 fixed, 4 foot contacts), "
                                     if args.config == 3 else "BASELINE.json configs[%d]: %s (%d bodies, %d contacts), " % (args.config - 1, spec.name, spec.Nb, len(spec.contacts)))
                                    + "batch=%d per GPU, %s, closed-loop rollout with random controls" % (B, "fwd + IFT gradients" if grad else "forward only")),
                       "io_dtype": args.io_dtype, "arithmetic": "fp64 state/residual/factorization, %s buffers at the ABI" % args.io_dtype,
                       "solver_options": "reference defaults (rtol 1e-6, btol 1e-4, max_iter 50, max_ls 10)",
                       "parallelism": "batch-sharded x%d, no data-path collective; per GPU %d independent environment groups of %d on their own HIP streams" % (world, NCH, B // NCH),
                       "converged_fraction_last_step": ok_frac, "mean_newton_iters_last_step": mean_iters},
            "roofline": dominant,
        }
        if other is not None:
            res["roofline_second_kernel"] = other
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(spec, grad)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def measured_traffic():
    """HBM bytes per launch from the committed PMC passes of this same command (tools/gpu_pmc.sh ->
    profiles/*_pmc_traffic.json): 2 x FETCH_SIZE + WRITE_SIZE, in KB, corrected as the MI355X guide prescribes."""
    import glob
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            for k, v in json.load(open(f)).items():
                out[k] = {"bytes_per_launch": v["bytes_per_launch"], "envs_per_launch": v.get("envs_per_launch", 4096), "source": os.path.relpath(f, ROOT),
                          "fp64_flops_per_launch": v.get("fp64_flops_per_launch")}
        except Exception:
            pass
    return out


def measured_valu_utilization():
    """What actually bounds the kernels: the fraction of wave cycles with a VALU instruction in flight, from the committed
    PMC pass (profiles/*_pmc_summary.txt: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both in quad-cycles)."""
    import glob
    import re
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt"))):
        txt = open(f).read()
        for kn in ("dojo_step_kernel", "dojo_grad_kernel"):
            a = re.search(kn + r"\s+SQ_ACTIVE_INST_VALU\s+per-dispatch mean ([0-9.e+-]+)", txt)
            c = re.search(kn + r"\s+SQ_WAVE_CYCLES\s+per-dispatch mean ([0-9.e+-]+)", txt)
            if a and c:
                out[kn] = {"valu_active_frac": float(a.group(1)) / float(c.group(1)), "source": os.path.relpath(f, ROOT)}
    return out


def cpu_baseline(spec, grad):
    """The C++ oracle ("port": the reference itself needs Julia, which is not installed) timed on
    the host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import dojo_amd as d
    from oracle import Oracle
    cores = os.cpu_count() or 1
    o = Oracle(spec)
    nsample = 64 * max(1, min(cores, 16) // 4)
    Z, U = d.synthetic_inputs(spec, nsample)
    t0 = time.perf_counter()
    Zn, st, it, dz, du = o.step_batch(Z, U, with_grad=grad, grad_mode=0, nthreads=cores)
    el = time.perf_counter() - t0
    # keep the sample within ~10-30 s of CPU work
    rounds = 1
    while el * cores < 10.0 and rounds < 64:
        t1 = time.perf_counter()
        o.step_batch(Z, U, with_grad=grad, grad_mode=0, nthreads=cores)
        el += time.perf_counter() - t1
        rounds += 1
    return {"value": nsample * rounds / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d Ant env-steps (fwd%s) of the same synthetic inputs, C++ oracle (dense KKT, fp64), one env per thread" % (nsample * rounds, "+grad" if grad else "")}


if __name__ == "__main__":
    main()
