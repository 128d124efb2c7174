#!/usr/bin/env python3
"""bench.py -- differentiable env-steps/s of the batched contact-implicit step on MI355X.

One "step" = one pass of the hot path (forward Mehrotra solve + IFT gradients) over a batch of
B = 4096 Ant environments per GPU (BASELINE.json configs[2], the config the metric is quoted on),
closed loop: z_{k+1} = step(z_k, u_k) with synthetic controls, all buffers resident in HBM.
N > 1: one process per GPU (torch.distributed / RCCL), the batch is sharded (weak scaling: B per
GPU is fixed; --batch-total T: strong scaling, T / N per GPU -- the form BASELINE.json quotes
configs 4 and 5 in) with no data-path collective; the trajectory chunk [K, B, 13 Nb] of the timed rollout is all-gathered
once over RCCL inside the timed region (north_star: "RCCL all-gather of trajectories").

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
    ap.add_argument("--batch", type=int, default=4096, help="environments per GPU (weak scaling: fixed as N grows)")
    ap.add_argument("--batch-total", type=int, default=0, help="STRONG scaling: a fixed total batch sharded over the N GPUs in contiguous slices "
                    "(BASELINE.json configs[3]: --config 4 --batch-total 8192, configs[4]: --config 5 --batch-total 2048); overrides --batch")
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (3 = Ant)")
    ap.add_argument("--io-dtype", default="f32")
    ap.add_argument("--no-grad", action="store_true")
    ap.add_argument("--distribution", default="baseline", choices=["baseline", "standing"],
                    help="synthetic states: BASELINE.md section 3's perturbation (default, every config) or, for Atlas, states around the reference's standing pose (dojo_amd.coords)")
    ap.add_argument("--chunks", type=int, default=0, help="environment groups the library steps the per-GPU batch as (dojo_set_groups); 0 = the library's choice "
                    "(16 at batch 4096), 1 = one launch per kernel on the caller's stream")
    ap.add_argument("--no-parity", action="store_true", help="skip the grad-inf-err-vs-CPU leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=0, help="1: dojo_set_async(h, 2) -- the IFT kernel of a group's step runs next to the group's next step kernel "
                    "(two hand-off records in turn; the rollout's states and controls sit in per-step buffers, as the asynchronous contract wants); 0: plain asynchronous groups")
    ap.add_argument("--dispatch-order", type=int, default=1, help="dojo_set_dispatch_order: 1 (the library's default) = joined steps of a batch of more workgroups than the "
                    "GPU holds at once hand the previous step's longest solves out first (the sync_per_step and single_launch legs); 0 = batch order; 2 = always")
    ap.add_argument("--iter-cap", type=int, default=-1, help="dojo_set_iteration_cap: solves unfinished after this many Newton iterations go on in the continuation "
                    "kernel (line-search trials side by side; joined steps only: the sync_per_step leg); 0 / -1 = off (the library's default)")
    ap.add_argument("--timed-only", action="store_true", help="warmup + the timed region only (no joined-per-step leg, no roofline leg, no parity, no CPU baseline): "
                    "what tools/gpu_pmc.sh profiles, so that its counters cover exactly the timed region's step window")
    ap.add_argument("--refine", type=float, default=None, help="stiffness threshold of the solve refinement (dojo_set_refinement); default: the library's")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1: nccl (= RCCL, production) or gloo "
                    "(plumbing check of the N > 1 path on a box with fewer GPUs than ranks: ranks share devices, the gather goes through the host)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: spawn the N ranks here (one process per GPU; the same command line under
    # torch.distributed.run on 127.0.0.1), fail loudly if the node has fewer than N devices.  Under a launcher (WORLD_SIZE set)
    # --gpus must agree with it.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, args.backend))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    # one hardware queue per environment group: ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware
    # queues, and streams that share a queue serialize (measured: 4 groups on the default 4 queues run at 0.6x, not 1.1x)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(24, args.chunks + 8)))
    import numpy as np
    import torch
    import dojo_amd as d
    from dojo_amd import api

    from dojo_amd import distributed as D
    import torch.distributed as dist
    rank, world, local = D.init_from_env(backend=args.backend)    # "nccl" IS RCCL on ROCm
    if args.backend != "nccl":
        local = local % torch.cuda.device_count()                 # plumbing check only: ranks may share a device
    elif local >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d (local rank %d) has no GPU of its own: %d device(s) visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    spec = d.baseline_config(args.config)
    if os.environ.get("DOJO_BENCH_ONE_CONTACT_PER_BODY") == "1":     # kernel experiments only (tools/gpu_r2_i.sh): a different mechanism, never a reported configuration
        seen = set(); spec.contacts = [c for c in spec.contacts if not (c.body in seen or seen.add(c.body))]
    strong = args.batch_total > 0
    if strong and args.batch_total % world:
        sys.exit("bench.py: --batch-total %d is not a multiple of the %d ranks" % (args.batch_total, world))
    B, K, W = (args.batch_total // world if strong else args.batch), args.steps, args.warmup
    tdt = torch.float32 if args.io_dtype == "f32" else torch.float64
    w = 4 if args.io_dtype == "f32" else 8
    # synthetic inputs: B DISTINCT seeded environments per rank (perturbed nominal states built in minimal
    # coordinates, so the joints are closed), distinct seeds per rank; controls ~ 0.5 N(0,1) on the actuated inputs, fresh every step
    # (strong scaling: ONE batch of batch_total environments all ranks agree on, rank r steps its contiguous slice [r B, (r + 1) B); weak: B fresh ones per rank)
    if strong:
        Z0, U0 = d.synthetic_inputs(spec, B, seed=20241008, distribution=args.distribution, offset=rank * B)
    else:
        Z0, U0 = d.synthetic_inputs(spec, B, seed=20241008 + rank, distribution=args.distribution)
    reps = 1
    z = torch.tensor(Z0, dtype=tdt, device=dev).contiguous()
    rng = np.random.Generator(np.random.Philox(key=[20241008, 1000 + rank]))
    Uall = torch.tensor(0.5 * rng.standard_normal((K + W, B, spec.nu)) * (np.abs(U0) > 0), dtype=tdt, device=dev).contiguous()
    zn = torch.empty_like(z)
    traj = torch.empty((K,) + tuple(z.shape), dtype=tdt, device=dev)      # the states after each timed step: what the ranks exchange (and what the rollout is for)
    status = torch.empty(B, dtype=torch.int32, device=dev); iters = torch.empty(B, dtype=torch.int32, device=dev)
    grad = not args.no_grad
    dz = torch.empty((B, spec.nx, spec.nx), dtype=tdt, device=dev) if grad else None
    du = torch.empty((B, max(spec.nu, 1), spec.nx), dtype=tdt, device=dev) if grad else None

    # ONE handle for the whole per-GPU batch.  The library itself steps the batch as independent environment groups on
    # internal HIP streams (dojo_step_dev, include/dojo_hip.h): environments are independent, so while the rare
    # environment that runs into max_iter = 50 (5x the mean iteration count, one wavefront) finishes in one group, the other
    # groups' kernels fill the GPU.  Asynchronous mode: consecutive steps chain per group, one dojo_join at the end.
    lib = api.lib()
    sys.path.insert(0, ROOT)
    from __graft_entry__ import build_info
    gm = api.BatchedMechanism(spec, B, dtype=args.io_dtype, device=local)
    if args.refine is not None:
        gm.set_refinement(args.refine)
    if args.chunks > 0:
        gm.set_groups(args.chunks)
    gm.set_iteration_cap(args.iter_cap)
    gm.set_dispatch_order(args.dispatch_order)
    gm.set_async(2 if args.pipeline else True)
    NCH = args.chunks if args.chunks > 0 else min(16, max(1, B // 256))
    lib_gather, lib_stuck = False, False
    gather_pref = os.environ.get("DOJO_BENCH_GATHER", "library")       # "torch": never try the library's communicator; "library-force": try it under gloo too (plumbing check of the fall-back)
    if world > 1 and ((args.backend == "nccl" and gather_pref == "library") or gather_pref == "library-force"):
        # The library's own RCCL communicator (dojo_comm_init; the id travels over torch.distributed) and one probe gather,
        # before the timed region and under a watchdog: this path has only ever run with one rank (the boxes of this pool have
        # one GPU), so a failure or a hang must cost nothing but the fall-back to torch's all-gather (the same RCCL call).
        import threading
        res = {}

        def _connect():
            try:
                torch.cuda.set_device(dev)
                D.connect_handle(gm, rank, world)
                probe = torch.full((8,), float(rank), device=dev, dtype=tdt)
                got = D.all_gather_states_rccl(gm, probe, world)
                torch.cuda.synchronize()
                res["ok"] = bool((got.view(world, 8)[:, 0].cpu() == torch.arange(world, dtype=tdt)).all())
            except Exception as e:
                res["err"] = e
        th = threading.Thread(target=_connect, daemon=True)
        th.start(); th.join(float(os.environ.get("DOJO_BENCH_COMM_TIMEOUT", "120")))
        lib_stuck = th.is_alive()
        lib_gather = bool(res.get("ok", False)) and not lib_stuck
        if not lib_gather:
            print("rank %d: library RCCL communicator not usable (%s): gathering with torch.distributed" % (rank, "timeout" if lib_stuck else res.get("err", "probe mismatch")), file=sys.stderr)
        flag = torch.tensor([1 if lib_gather else 0], device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # all ranks or none
        lib_gather = bool(flag.item())

    def ptr(t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    def one_step(k, out=None):
        """z <- step(z, u_k); `out`: the buffer the new state is written to (a row of the trajectory), else the ping-pong buffer"""
        nonlocal z, zn
        dst = zn if out is None else out
        api._chk(lib.dojo_step_dev(gm.h, ptr(z), ptr(Uall[k]), ptr(dst), ptr(status), ptr(iters), ptr(dz) if grad else None, ptr(du) if grad else None,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if out is None:
            z, zn = zn, z
        else:
            z = out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()                     # inputs were produced on the default stream
    wtraj = torch.empty((max(W, 1),) + tuple(traj.shape[1:]), device=traj.device, dtype=traj.dtype)      # (per-step buffers for the warmup too: the inputs of un-joined steps stay untouched)
    for k in range(W):
        one_step(k, wtraj[k])
    gm.join(torch.cuda.current_stream().cuda_stream)
    barrier()
    gm.kernel_time_totals(reset=True)                      # (drains the warmup's event pairs)
    t0 = time.perf_counter()
    for k in range(W, W + K):
        one_step(k, traj[k - W])
    gm.join(torch.cuda.current_stream().cuda_stream)       # the environment groups -> torch's stream
    gathered_bytes = 0
    if world > 1:
        # the trajectory chunk [K, B, 13 Nb] of this rollout over RCCL/xGMI, once: dojo_allgather_dev (the library's communicator), else torch's
        if args.backend == "nccl" or lib_gather:
            traj_all = D.all_gather_states_rccl(gm, traj, world) if lib_gather else D.all_gather_states(traj, world)
        else:
            torch.cuda.synchronize(); traj_all = D.all_gather_states(traj.cpu(), world)
        gathered_bytes = traj_all.numel() * traj_all.element_size()
        # rank r's block of the gathered trajectory is what rank r computed (rank order = the order of the contiguous batch shards): every rank checks its own
        gather_ok = bool(torch.equal(traj_all.reshape((world,) + tuple(traj.shape))[rank].to(traj.device), traj))
    barrier()
    el = D.max_over_ranks(time.perf_counter() - t0, world, device=dev if args.backend == "nccl" else "cpu")
    # hipEvent durations of the timed region's launches, summed over the environment groups (every group's kernels on its own stream): against
    # ms_per_step this is how much the groups overlapped
    ksum_step, ksum_ift, klaunches = gm.kernel_time_totals(reset=True)
    ok_frac = float((status == 0).float().mean().item())          # of the timed region's last step
    mean_iters = float(iters.float().mean().item())
    z = traj[K - 1].clone()
    if args.timed_only:
        if rank == 0:
            print(json.dumps({"metric": "differentiable env-steps/sec (fwd+grad) at batch=4096; grad inf-err vs CPU", "value": world * B * K / el, "unit": "env-steps/s", "n_gpus": world,
                              "steps": K, "warmup": W, "ms_per_step": 1e3 * el / K, "note": "--timed-only: warmup + timed region (profiling runs)"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    rank_devices = None
    if world > 1:                                 # which device every rank ran on (rank 0 prints it)
        box = [None] * world
        dist.all_gather_object(box, "rank %d: cuda:%d %s" % (rank, local, torch.cuda.get_device_name(local)))
        rank_devices = box
        box2 = [None] * world
        dist.all_gather_object(box2, {"rank": rank, "env_range": [rank * B, (rank + 1) * B] if strong else [0, B], "own_block_of_the_gather_matches": gather_ok,
                                      "first_state_checksum": float(Z0[0].sum())})
        rank_slices = box2
    # the same closed loop with a join of the environment groups into the caller's stream after EVERY step (what a policy over the
    # whole batch needs: a barrier per step; the max_iter tail of a step is then not hidden behind the next step), outside the timed region
    gm.set_async(False)
    barrier()
    t1 = time.perf_counter()
    for k in range(W, W + K):
        one_step(k)
    barrier()
    el_sync = D.max_over_ranks(time.perf_counter() - t1, world, device=dev if args.backend == "nccl" else "cpu")

    # Kernel durations for the roofline, outside the timed region: the SAME closed loop once more from the same initial states with the
    # same controls -- the warmup steps, then the timed region's K steps -- as ONE launch of the whole batch per kernel (groups = 1), so that
    # the hipEvent durations (on the launch stream) are those of a kernel that has the GPU to itself: the figure the rocprofv3
    # --kernel-trace of `bench.py --chunks 1` shows, over the step window the PMC passes of tools/gpu_pmc.sh count (profiles/*_pmc_traffic.json).
    gm.set_async(False); gm.set_groups(1); gm.set_iteration_cap(0)                        # (whole solves inside dojo_step_kernel, as in the PMC passes)
    z = torch.tensor(Z0, dtype=tdt, device=dev).contiguous()
    torch.cuda.synchronize()
    for k in range(W):
        one_step(k)
    torch.cuda.synchronize()
    per_launch = []
    for k in range(W, W + K):
        one_step(k)
        per_launch.append(gm.last_kernel_times())        # (step kernel ms, IFT kernel ms) of this launch; waits for its end event
    step_ms = sum(a for a, _ in per_launch) / len(per_launch); ift_ms = sum(b for _, b in per_launch) / len(per_launch)
    step_ms_best = min(a for a, _ in per_launch); ift_ms_best = min(b for _, b in per_launch)

    if rank == 0:
        nb, nu = spec.Nb, spec.nu
        bytes_fwd = (26 * nb + nu) * w + 8
        bytes_grad = 12 * nb * (12 * nb + nu) * w if grad else 0
        # one step = two launches: dojo_step_kernel (Newton loop) and dojo_grad_kernel (IFT back-solves).  Algorithmic bytes per
        # launch: the step kernel reads z,u and writes z_next,status,iters; the IFT kernel writes dz,du.
        traffic = measured_traffic()
        util = measured_valu_utilization()
        binfo = build_info()

        def roof(kernel, ms, nbytes):
            """The bound of these kernels is the fp64 vector ALU (the KKT systems never leave registers / LDS):
            achieved = fp64 flops the kernel EXECUTES per launch (PMC pass SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 of the committed
            profile of this command, x64 lanes, FMA = 2) / its average duration measured here.  HBM figures ride along."""
            t = traffic.get(kernel)
            scale = B / t["envs_per_launch"] if t else 0.0
            fl = t["fp64_flops_per_launch"] * scale if (t and t.get("fp64_flops_per_launch")) else None
            ach_v = fl / (ms * 1e-3) / 1e12 if (fl and ms > 0) else None
            ach_h = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            r = {"bound": "valu_fp64", "kernel": kernel, "avg_kernel_ms": ms, "achieved": ach_v, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": (ach_v / FP64_VECTOR_PEAK_TFLOPS) if ach_v else None, "executed_fp64_flops_per_launch": fl,
                 "flops_source": t["source"] if t else None,
                 # the PMC pass belongs to THIS library (content hash of its sources, written by tools/gpu_pmc.sh) and covers the timed region's steps
                 "flops_source_digest_matches": (t.get("library_digest") == binfo.get("library_digest")) if (t and t.get("library_digest")) else None,
                 "flops_source_step_window": t.get("step_window") if t else None,
                 "traffic": t["bytes_per_launch"] * scale if t else None,     # HBM bytes per launch, PMC (2 x FETCH_SIZE + WRITE_SIZE), scaled to this launch size
                 "hbm": {"bound": "hbm", "achieved": ach_h, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_h / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}}
            if kernel in util:
                r["valu_active_frac"] = util[kernel]["valu_active_frac"]      # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the committed PMC pass
                r["valu_source"] = util[kernel]["source"]
            return r
        r_step = roof("dojo_step_kernel", step_ms, bytes_fwd * B)
        r_ift = roof("dojo_grad_kernel", ift_ms, bytes_grad * B) if grad else None
        # the same figure for the shortest of those launches: a launch whose environments all converge in ~10 iterations (a stalled
        # environment keeps one wavefront busy for 50 iterations with exhausted line searches, 5-10x the rest of the launch)
        for r, best in ((r_step, step_ms_best), (r_ift, ift_ms_best)):
            if r is not None and r.get("executed_fp64_flops_per_launch") and best > 0:
                a_ = r["executed_fp64_flops_per_launch"] / (best * 1e-3) / 1e12
                r["best_launch"] = {"kernel_ms": best, "achieved": a_, "frac": a_ / FP64_VECTOR_PEAK_TFLOPS}
        for r in (r_step, r_ift):
            if r is not None and r.get("flops_source_digest_matches") is False:
                r["warning"] = "STALE COUNTS: %s was taken with another build of the library (digest mismatch); re-run tools/gpu_pmc.sh" % r["flops_source"]
        # The roofline that belongs to `value`: what BOTH kernels execute per step of the timed region (the asynchronous rollout, environment
        # groups overlapping) / ms_per_step.  The one-launch-per-kernel replay (groups = 1: a kernel that has the GPU to itself, the figure a
        # rocprofv3 --kernel-trace of `bench.py --chunks 1` shows) rides along as `single_launch`, per kernel.
        ms_step = 1e3 * el / K
        parts = [r_ for r_ in (r_step, r_ift) if r_ is not None]
        have_fl = all(r_.get("executed_fp64_flops_per_launch") for r_ in parts)
        fl_step = sum(r_["executed_fp64_flops_per_launch"] for r_ in parts) if have_fl else None
        tr_step = sum(r_["traffic"] for r_ in parts) if all(r_.get("traffic") for r_ in parts) else None
        alg_bytes = (bytes_fwd + bytes_grad) * B
        a_ = fl_step / (1e-3 * ms_step) / 1e12 if fl_step else None
        roofline = {"bound": "valu_fp64", "kernel": "dojo_step_kernel + dojo_grad_kernel: one step of the timed rollout",
                    "achieved": a_, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": (a_ / FP64_VECTOR_PEAK_TFLOPS) if a_ else None,
                    "executed_fp64_flops_per_step": fl_step, "ms_per_step": ms_step,
                    "traffic": tr_step,        # HBM bytes per step, PMC (2 x FETCH_SIZE + WRITE_SIZE of both kernels), scaled to this batch
                    "hbm": {"bound": "hbm", "achieved": alg_bytes / (1e-3 * ms_step) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg_bytes / (1e-3 * ms_step) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg_bytes},
                    "kernel_ms_sum_per_step": {"dojo_step_kernel": ksum_step / K, "dojo_grad_kernel": ksum_ift / K, "launches_per_step": klaunches / K,
                                               "overlap_factor": (ksum_step + ksum_ift) / K / ms_step if ms_step > 0 else None,
                                               "note": "hipEvent durations of the timed region's launches summed over the %d environment groups, per step; / ms_per_step = how many kernels were in flight on average" % NCH},
                    "flops_source": r_step.get("flops_source"), "flops_source_digest_matches": r_step.get("flops_source_digest_matches"),
                    "single_launch": {r_["kernel"]: r_ for r_ in parts},
                    "note": ("fp64 vector-ALU-bound lane program. achieved = fp64 flops both kernels EXECUTE per step (PMC pass of this command, x64 lanes, FMA = 2) / ms_per_step "
                             "of the timed region; single_launch = each kernel as ONE launch of the whole batch (groups = 1), hipEvents on the launch stream, mean over the timed "
                             "region's K steps replayed after it (same states, same controls); the HBM roofline the contract asks for is the `hbm` member")}
        if any(r_.get("warning") for r_ in parts):
            roofline["warning"] = next(r_["warning"] for r_ in parts if r_.get("warning"))
        res = {
            "metric": "differentiable env-steps/sec (fwd+grad) at batch=4096; grad inf-err vs CPU" if grad else "env-steps/sec (fwd only)",
            "value": world * B * K / el, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * el / K, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (("BASELINE.json configs[2]: Ant (13 bodies as built by the reference, 8 revolute+limits, 4 fixed, 4 foot contacts), "
                                     if args.config == 3 else "BASELINE.json configs[%d]: %s (%d bodies, %d contacts), " % (args.config - 1, spec.name, spec.Nb, len(spec.contacts)))
                                    + ("total batch %d sharded over %d GPU(s) = %d per GPU (strong scaling), " % (world * B, world, B) if strong else "batch=%d per GPU, " % B)
                                    + "%s, closed-loop rollout with random controls" % ("fwd + IFT gradients" if grad else "forward only")),
                       "per_rank_batch": B, "total_batch": world * B,
                       "input_distribution": args.distribution + (" (BASELINE.md section 3: height U(0, 0.3), rotation N(0, 0.1), velocities N(0, 0.5), joints +-0.2)" if args.distribution == "baseline"
                                                                   else " (Atlas: around the reference's standing pose, dojo_amd.coords._SYNTH_STANDING; other mechanisms as baseline)"),
                       "io_dtype": args.io_dtype, "arithmetic": "fp64 state/residual/factorization, %s buffers at the ABI" % args.io_dtype,
                       "solver_options": "reference defaults (rtol 1e-6, btol 1e-4, max_iter 50, max_ls 10)",
                       "parallelism": "batch-sharded x%d, no data-path collective; per GPU ONE handle, dojo_step_dev steps its batch as %d environment groups on internal HIP streams (asynchronous, one join per rollout%s)" % (world, NCH, "; pipelined: a group's IFT kernel of step k next to its step kernel of step k + 1" if args.pipeline else ""),
                       "dispatch_order": {0: "batch order", 1: "library default: joined steps (the sync_per_step and single_launch legs) hand out the previous step's longest solves first; the asynchronous timed region runs in batch order", 2: "longest solves of the previous step first, in every leg"}[args.dispatch_order],
                       "iteration_cap": "off (library default; dojo_set_iteration_cap: measured gain only at per-GPU batches <= 2048, DESIGN.md section 6)" if args.iter_cap <= 0 else args.iter_cap,
                       "converged_fraction_last_step": ok_frac, "mean_newton_iters_last_step": mean_iters,
                       "sync_per_step_value": world * B * K / el_sync, "sync_per_step_ms": 1e3 * el_sync / K,
                       "sync_per_step_note": "%d more steps of the rollout (the same controls again, from the timed region's end state) with the environment groups joined into the caller's stream after every step (a barrier per step); `value` is the asynchronous rollout (one join at the end)" % K,
                       "build": build_info()},
            "roofline": roofline,
        }
        if grad and not args.no_parity and world == 1:
            res["grad_inf_err_vs_cpu"] = pv = parity_vs_cpu(spec, B, local, args.distribution)
            # the norm the metric's second half is claimed in, at the top level: RELATIVE per-environment inf-norm, the absolute one next to it
            res["grad_err_claim"] = {"norm": "per environment, over dz and du, maximum over every environment that converged on both sides to the same point: ABSOLUTE inf-norm |J_gpu - J_cpu|_inf "
                                             "(the contract's norm: `within_bounds`) and RELATIVE inf-norm |J_gpu - J_cpu|_inf / max(1, |J_cpu|_inf) (`within_relative_bound`)",
                                     "bound": {"f64": 1e-6, "f32": 1e-3},
                                     "statement": "the timed path (fp32 ABI) meets 1e-3 in both norms; the fp64 ABI meets 1e-6 in the relative norm on every such environment and in the "
                                                  "absolute norm on all but `n_grad_abs_err_above_bound` of them; `reference_like_direct_solve` is the oracle WITHOUT its refinement rounds "
                                                  "(the reference's own plain-fp64 direct solves) against the refined oracle on the same batch: on the environment with entries ~2e3 it is itself "
                                                  "1.8e-6 off, i.e. 1e-6 absolute there is below what the reference's arithmetic reproduces; with every linear solve refined "
                                                  "(dojo_set_refinement(h, 0)) the device meets the absolute norm as well",
                                     "timed_path_f32_abi": {k_: pv.get("f32", {}).get(k_) for k_ in ("grad_inf_err_max", "grad_abs_inf_err_max", "state_inf_err_max", "within_bounds", "within_relative_bound", "n_grad_abs_err_above_bound")},
                                     "f64_abi": {k_: pv.get("f64", {}).get(k_) for k_ in ("grad_inf_err_max", "grad_abs_inf_err_max", "state_inf_err_max", "within_bounds", "within_relative_bound", "n_grad_abs_err_above_bound", "jacobian_inf_norm_of_worst_abs")},
                                     "f64_abi_all_solves_refined": {k_: pv.get("f64_refined", {}).get(k_) for k_ in ("grad_inf_err_max", "grad_abs_inf_err_max", "state_inf_err_max", "within_bounds", "within_relative_bound", "n_grad_abs_err_above_bound")},
                                     "reference_like_direct_solve": pv.get("reference_like_direct_solve"),
                                     "jacobian_inf_norm_max": pv.get("f64", {}).get("jacobian_inf_norm_max")}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cb = cpu_baseline(spec, grad, mean_iters, (fl_step / B) if fl_step else None, args.distribution)
            # the USEFUL share: what a block-sparse direct method needs for this step (cpu_baseline.sparse_lu_flops) / ms_per_step, against the same peak
            u_ = cb.get("sparse_lu_flops", {}).get("per_env_step")
            if u_:
                a_ = u_ * B / (1e-3 * 1e3 * el / K) / 1e12
                rf_ = cb.get("sparse_lu_flops", {}).get("reference_formula_flops", {}).get("per_env_step")
                res["roofline"]["useful"] = {"flops_per_step": u_ * B, "achieved": a_, "unit": "TFLOP/s", "frac": a_ / FP64_VECTOR_PEAK_TFLOPS,
                                             "lane_efficiency": (u_ * B / fl_step) if fl_step else None,
                                             "reference_formula_flops_per_step": (rf_ * B) if rf_ else None,
                                             "algorithm_flops_over_executed": ((u_ + rf_) * B / fl_step) if (fl_step and rf_) else None,
                                             "note": "flops of a block-sparse no-pivot LU (one factorization + two solves per Newton iteration, one + a solve per Jacobian column) / ms_per_step; "
                                                     "lane_efficiency = these / the executed fp64 flops (the linear algebra alone: a lower bound of the useful share).  reference_formula_flops_per_step = "
                                                     "what the reference's own formulas execute OUTSIDE the solves (assembly, line-search residuals, violations, IFT data matrix), counted by running "
                                                     "the oracle on an operation-counting scalar (cpu_baseline.sparse_lu_flops.reference_formula_flops); algorithm_flops_over_executed = (LU + those) / "
                                                     "executed: the reference's generic small-matrix products cost several times the flops of the device's closed forms, so this ratio is an UPPER "
                                                     "estimate of the useful share (the device's redundancy -- four lanes evaluate every joint -- is in the denominator, its cheaper formulas are not in the numerator)"}
        if world > 1:
            res["config"]["trajectory_gather"] = {"what": "the timed rollout's states [K=%d, B=%d, 13 Nb=%d] of every rank, once, inside the timed region" % (K, B, 13 * spec.Nb),
                                                 "bytes_received_per_rank": gathered_bytes,
                                                 "through": "dojo_allgather_dev (the library's RCCL communicator)" if lib_gather else "torch.distributed all_gather (RCCL)"}
            res["config"]["rank_devices"] = rank_devices
            res["config"]["rank_slices"] = rank_slices        # (strong scaling: rank r = environments [r B, (r + 1) B) of the one batch_total-environment batch)
        print(json.dumps(res), flush=True)
    if world > 1:
        if lib_stuck:                            # a thread is still inside the library's communicator set-up: leave without the teardown that would wait for it
            sys.stdout.flush(); sys.stderr.flush(); os._exit(0)
        dist.destroy_process_group()


def spawn_ranks(n, backend):
    """Re-executes this command line as n ranks on this node (torch.distributed.run, rendezvous on 127.0.0.1, a free port).
    The device count is checked first -- through the library, not torch, so that the parent never creates a GPU context."""
    import socket
    import subprocess
    if backend == "nccl":                         # (--backend gloo: ranks may share devices, plumbing check)
        from dojo_amd import api
        have = api.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but this node has %d GPU(s)\n" % (n, have))
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def parity_vs_cpu(spec, B, device, distribution="baseline"):
    """The metric's second half: state and gradient inf-norm error of the device against the CPU oracle at the BASELINE batch
    with B DISTINCT seeded environments, reference-default solver options, after 8 closed-loop steps (so that feet are on the
    ground).  fp64 ABI against the north-star bound 1e-6, fp32 ABI (what the timed loop uses) against 1e-3.  Gradient error is
    relative: |dz_gpu - dz_cpu|_inf / max(1, |dz_cpu|_inf) per environment."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import dojo_amd as d
    from dojo_amd import api
    from oracle import Oracle
    cores = os.cpu_count() or 1
    Z, U = d.synthetic_inputs(spec, B, distribution=distribution)
    out = {"envs": B, "pre_steps": 8, "solver_options": "reference defaults", "oracle": "C++ restatement, dense pivoted LU + 2 rounds of long-double refinement"}
    o = Oracle(spec)
    g64 = api.BatchedMechanism(spec, B, dtype="f64", device=device)
    for _ in range(8):
        Z, st, it = g64.step(Z, U)
    # f64_refined: the same step with every linear solve of every environment refined against the uncondensed blocks
    # (dojo_set_refinement(h, 0): what a caller who needs the 1e-6 bound on EVERY environment switches on; slower, not the timed path)
    for name, dt in (("f64", np.float64), ("f64_refined", np.float64), ("f32", np.float32)):
        Zi, Ui = Z.astype(dt), U.astype(dt)
        gm = g64 if dt == np.float64 else api.BatchedMechanism(spec, B, dtype="f32", device=device)
        if name == "f64_refined":
            gm.set_refinement(0.0)
        zn, st, it = gm.step(Zi, Ui, with_gradient=True)
        dz, du = gm.gradients()
        if name != "f64_refined":                # (the refined leg is checked against the fp64 leg's oracle run: same inputs)
            Zo, st_o, it_o, dz_o, du_o = o.step_batch(d.fp32_abi_state(Zi) if name == "f32" else Zi, Ui.astype(np.float64), with_grad=True, nthreads=cores)   # fp32: the state the buffer stands for
        ok = (st == 0) & (st_o == 0)
        idx = np.nonzero(ok)[0]
        ez = np.abs(zn.astype(np.float64) - Zo).max(axis=1)[ok]
        ea = np.array([max(np.abs(dz[b] - dz_o[b]).max(), np.abs(du[b] - du_o[b]).max()) for b in idx])                  # absolute inf-norm
        eg = np.array([max(np.abs(dz[b] - dz_o[b]).max() / max(1.0, np.abs(dz_o[b]).max()), np.abs(du[b] - du_o[b]).max() / max(1.0, np.abs(du_o[b]).max())) for b in idx])
        # environments whose two solves ended at different points (both within the solver's tolerances, states more than the 1e-6
        # bound apart -- long solves that wander at mu ~ 1e-12): their Jacobians are Jacobians of different points; listed, not hidden
        tol_s = 1e-6 if dt == np.float64 else 1e-5
        apart = ez > tol_s
        same = ~apart
        if not same.any():
            out[name] = {"converged_both": int(ok.sum()), "error": "no environment converged on both sides to the same point"}
            continue
        # north_star: "state/gradient inf-norm <= 1e-6 fp64, <= 1e-3 fp32" -- the ABSOLUTE inf-norm decides `within_bounds`; the relative one
        # (|dJ|_inf / max(1, |J|_inf): what fp64 arithmetic on Jacobians with entries up to 2e3 can promise) rides along as `within_relative_bound`
        gb, ab = (1e-6, 1e-6) if dt == np.float64 else (1e-3, 1e-3)
        out[name] = {"converged_both": int(ok.sum()), "status_mismatch": int((st != st_o).sum()), "iters_mismatch": int((it[ok] != it_o[ok]).sum()),
                     "state_inf_err_max": float(ez[same].max()), "state_inf_err_max_unfiltered": float(ez.max()), "n_state_err_above_bound": int(apart.sum()), "state_bound": tol_s,
                     "grad_inf_err_max": float(eg[same].max()), "grad_abs_inf_err_max": float(ea[same].max()), "grad_inf_err_max_unfiltered": float(eg.max()),
                     "grad_bound_relative": gb, "grad_bound_absolute": ab, "within_bounds": bool(ea[same].max() <= ab and ez[same].max() <= tol_s),
                     "within_relative_bound": bool(eg[same].max() <= gb and ez[same].max() <= tol_s),
                     "n_grad_abs_err_above_bound": int((ea[same] > ab).sum()),
                     "jacobian_inf_norm_of_worst_abs": float(max(np.abs(dz_o[idx[i]]).max(), np.abs(du_o[idx[i]]).max()) if (i := int(np.argmax(np.where(same, ea, -1.0)))) >= 0 else 0.0),
                     "grad_inf_err_q99": float(np.quantile(eg, 0.99)),
                     "grad_inf_err_q50": float(np.quantile(eg, 0.5)), "n_grad_err_above_1e-6": int((eg[same] > 1e-6).sum()),
                     "jacobian_inf_norm_max": float(max(np.abs(dz_o[b]).max() for b in idx)),
                     "solves_that_ended_apart": [{"env": int(idx[i]), "iters": int(it[idx[i]]), "iters_cpu": int(it_o[idx[i]]), "state_err": float(ez[i]), "grad_err": float(eg[i])} for i in np.nonzero(apart)[0][:16]],
                     "note": "grad errors are per-environment inf-norms, relative = / max(1, |J_cpu|_inf), absolute next to it; maxima over every environment that converged on both sides and whose states agree within state_bound (the others are listed, and enter the *_unfiltered maxima)"}
        if name == "f64":
            # What the 1e-6 ABSOLUTE bound means on this batch: the oracle once more WITHOUT its refinement rounds -- a dense partial-pivot LU in plain
            # fp64, i.e. the reference's own `solmat \\ datamat` (src/gradients/state.jl:99) and LDU -- against the refined oracle, same inputs.
            o_plain = Oracle(spec); o_plain.set_refine_steps(0)
            Zp, st_p, it_p, dz_p, du_p = o_plain.step_batch(Zi, Ui.astype(np.float64), with_grad=True, nthreads=cores)
            okp = (st_p == 0) & (st_o == 0)
            eap = np.array([max(np.abs(dz_p[b] - dz_o[b]).max(), np.abs(du_p[b] - du_o[b]).max()) for b in np.nonzero(okp)[0]])
            wp = int(np.nonzero(okp)[0][int(np.argmax(eap))])
            out["reference_like_direct_solve"] = {"grad_abs_inf_err_max": float(eap.max()), "n_grad_abs_err_above_bound": int((eap > 1e-6).sum()), "env_of_max": wp,
                                                  "jacobian_inf_norm_there": float(max(np.abs(dz_o[wp]).max(), np.abs(du_o[wp]).max())), "device_abs_err_there": float(max(np.abs(dz[wp] - dz_o[wp]).max(), np.abs(du[wp] - du_o[wp]).max())),
                                                  "note": "the oracle without its two rounds of long-double refinement (dense partial-pivot LU in plain fp64 = the reference's direct solves) against the refined oracle "
                                                          "the device is checked with: where this exceeds 1e-6, the bound is below what the reference's own arithmetic reproduces (Jacobians with entries ~1e3 at "
                                                          "a contact about to switch: the error there comes from the last Newton steps' 1e-11 state difference, not from the IFT solve)"}
            del dz_p, du_p
        del dz, du
        if gm is not g64:
            gm.close()
    g64.close()
    return out


def measured_traffic():
    """HBM bytes per launch from the committed PMC passes of this same command (tools/gpu_pmc.sh ->
    profiles/*_pmc_traffic.json): 2 x FETCH_SIZE + WRITE_SIZE, in KB, corrected as the MI355X guide prescribes."""
    import glob
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            for k, v in json.load(open(f)).items():
                out[k] = {"bytes_per_launch": v["bytes_per_launch"], "envs_per_launch": v.get("envs_per_launch", 4096), "source": os.path.relpath(f, ROOT),
                          "fp64_flops_per_launch": v.get("fp64_flops_per_launch"), "library_digest": v.get("library_digest"), "step_window": v.get("step_window")}
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


def cpu_quota():
    """CPU-time quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(spec, grad, mean_iters=None, executed_flops_per_env=None, distribution="baseline"):
    """The C++ oracle ("port": the reference itself needs Julia, which is not installed) timed on the host's PHYSICAL cores: one
    persistent thread pinned to each core, every thread owns a copy of the mechanism (its workspaces and symbolic factorization
    are set up by one untimed step) and walks 128 environments of the same synthetic batch, once per solver variant; the clock
    starts when all threads are ready.  The same loop on ONE thread gives `single_thread` and the parallel efficiency.
    Linear solves without the checker's refinement rounds: one factorization and two solves per Newton iteration like
    src/solver/mehrotra.jl:36-49, one factorization + 170 right-hand sides for the IFT (src/gradients/state.jl:99).
      value  = the BLOCK-SPARSE variant SURVEY.md §8d asks for: sparse LU without pivoting in the elimination order of the
               mechanism graph (contacts and limits first, then the tree leaves -> root), the structure of the reference's LDU;
      dense  = the dense 206 x 206 partial-pivot LU (what the checker itself uses, without its refinement)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dojo_amd as d
    import oracle as orc
    from oracle import Oracle
    physical = orc.physical_cores() or (os.cpu_count() or 1)
    quota = cpu_quota()
    cores = max(1, min(physical, int(quota))) if quota else physical      # a container's CPU-time quota (cgroup) caps what threads can use: more threads only get throttled
    o = Oracle(spec, fast=True)                   # liboracle_fast.so: the same source compiled -O3 -march=native (the checker build is host-independent, oracle/Makefile)
    o.set_refine_steps(0)
    per_thread = 128
    nsample = per_thread * cores
    Z, U = d.synthetic_inputs(spec, nsample, distribution=distribution)
    out = {}
    for name, sparse in (("sparse", True), ("dense", False)):
        o.set_sparse_solver(sparse)
        el1 = o.time_batch(Z[:per_thread // 2], U[:per_thread // 2], with_grad=grad, nthreads=1, rounds=1)
        el = o.time_batch(Z, U, with_grad=grad, nthreads=cores, rounds=1)
        single = (per_thread // 2) / el1
        out[name] = {"value": nsample / el, "cpu_seconds": el * cores + el1, "single_thread": single, "parallel_efficiency": (nsample / el) / (cores * single)}
    # what a block-sparse direct method needs per env-step (flops of the sparse LU above, multiply-add = 2): one factorization
    # + two solves per Newton iteration, one factorization + one solve per Jacobian column -- the "useful" share of what the
    # kernels execute (roofline.executed_fp64_flops_per_launch also counts assembly, line searches, replicated and masked lanes)
    o.set_sparse_solver(True); o.step(Z[0], U[0])
    fF, fS = o.sparse_flops(), o.sparse_solve_flops()
    ncol = 12 * spec.Nb + spec.nu
    la = {"factor": fF, "solve_per_rhs": fS, "per_newton_iteration": fF + 2 * fS, "ift": (fF + ncol * fS) if grad else 0}
    if mean_iters:
        la["per_env_step"] = mean_iters * la["per_newton_iteration"] + la["ift"]
        if executed_flops_per_env:
            la["frac_of_executed_fp64"] = la["per_env_step"] / executed_flops_per_env
    # ... and what the reference's FORMULAS need outside the linear solves: set_entries!, every residual evaluation of the line searches, the
    # violations, the data matrix of the IFT -- the oracle on an operation-counting scalar (oracle/counted.hpp; + - * / sqrt sin cos atan = 1 each,
    # so a multiply-add = 2 as in the PMC figures), 16 environments of the same batch.  The oracle restates the reference's generic small-matrix
    # algebra (products with structurally zero blocks included), not the device's closed forms: an upper estimate of the useful assembly work.
    try:
        import numpy as np
        oc = Oracle(spec, dtype="count")
        oc.op_count()
        ops_step, ops_grad, its_ = [], [], []
        for i in range(min(16, len(Z))):
            _, info = oc.step(Z[i], U[i]); ops_step.append(oc.op_count())
            if grad:
                oc.gradients(0); ops_grad.append(oc.op_count())
            its_.append(info["iters"])
        sc_ = (mean_iters / float(np.mean(its_))) if (mean_iters and np.mean(its_) > 0) else 1.0      # (the sample's solves against the timed rollout's mean iteration count)
        la["reference_formula_flops"] = {"per_env_step": float(sc_ * np.mean(ops_step) + (np.mean(ops_grad) if ops_grad else 0.0)), "newton_loop_of_the_sample": float(np.mean(ops_step)),
                                         "ift_data_matrix": float(np.mean(ops_grad)) if ops_grad else 0.0, "mean_iters_of_the_sample": float(np.mean(its_)), "scaled_to_iters": mean_iters, "envs": len(its_)}
    except Exception as e:      # (an oracle library older than the counting scalar)
        la["reference_formula_flops"] = {"error": str(e)}
    return {"value": out["sparse"]["value"], "unit": "env-steps/s", "cores": cores, "threads": cores, "kind": "port",
            "single_thread": out["sparse"]["single_thread"], "parallel_efficiency": out["sparse"]["parallel_efficiency"],
            "cpu_seconds": out["sparse"]["cpu_seconds"] + out["dense"]["cpu_seconds"],
            "dense": out["dense"]["value"], "dense_single_thread": out["dense"]["single_thread"], "dense_parallel_efficiency": out["dense"]["parallel_efficiency"],
            "logical_cpus": os.cpu_count(), "physical_cores_of_the_host": physical, "cgroup_cpu_quota": quota, "sparse_lu_flops": la,
            "sample": "%d Ant env-steps (fwd%s) per variant: %d synthetic environments per thread, one pinned thread per core on %d physical cores (= the container's CPU quota where there is one; after one untimed "
                      "step per thread); C++ oracle, fp64; value = block-sparse no-pivot LU in the mechanism graph's elimination order, dense = 206x206 partial-pivot LU"
                      % (nsample, "+grad" if grad else "", per_thread, cores)}


if __name__ == "__main__":
    main()
