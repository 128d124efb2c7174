"""GPU tier of the multi-GPU path.  The boxes this suite runs on have ONE GPU, so:
 * two ranks share device 0 (gloo carries the gather through the host; RCCL refuses two ranks on one device): the sharded
   Quadruped step at the BASELINE batch, 8192 / 2 per rank, gathered == the unsharded batch, bit for bit;
 * the library's own RCCL path (dojo_comm_unique_id / dojo_comm_init / dojo_allgather_dev) runs with world = 1."""
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, sys
ROOT = %r
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import numpy as np, torch
import dojo_amd as d
from dojo_amd import api, distributed as D
rank, world, local = D.init_from_env(backend="gloo")
spec = d.baseline_config(4)                       # Quadruped, BASELINE configs[3]: batch 8192 sharded over the ranks
B = 8192
Z, U = d.synthetic_inputs(spec, B)
lo, hi = D.shard_slice(B, rank, world)
gm = api.BatchedMechanism(spec, hi - lo, dtype="f32", device=0)
def step(z, u):
    zn, st, it = gm.step(z.astype(np.float32), u.astype(np.float32), with_gradient=False)
    return zn, st, it
zg, sg, ig = D.sharded_step(step, Z, U, rank, world)
gm.close()
if rank == 0:
    full = api.BatchedMechanism(spec, B, dtype="f32", device=0)
    zf, sf, itf = full.step(Z.astype(np.float32), U.astype(np.float32))
    full.close()
    assert np.array_equal(zg.numpy(), zf) and np.array_equal(sg.numpy(), sf) and np.array_equal(ig.numpy(), itf), "sharded != unsharded"
    print("SHARD_OK", float((sf == 0).mean()))
""" % ROOT


def test_two_ranks_sharded_quadruped_equals_unsharded(tmp_path):
    f = tmp_path / "worker.py"
    f.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(f)], capture_output=True, text=True, timeout=900, env=env)
    assert "SHARD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_library_rccl_allgather_world_one():
    """dojo_comm_unique_id -> dojo_comm_init -> dojo_allgather_dev through RCCL itself (a communicator of one rank)."""
    import ctypes as C
    import torch
    import dojo_amd as d
    from dojo_amd import api
    spec = d.baseline_config(2)
    gm = api.BatchedMechanism(spec, 64, dtype="f64")
    uid = gm.comm_unique_id()
    assert len(uid) == 128
    gm.comm_init(0, 1, uid)
    Z, U = d.synthetic_inputs(spec, 64)
    z = torch.tensor(Z, device="cuda:0"); out = torch.empty_like(z)
    torch.cuda.synchronize()
    gm.allgather_dev(z.data_ptr(), out.data_ptr(), z.numel(), stream=torch.cuda.current_stream().cuda_stream)
    st = torch.arange(64, dtype=torch.int32, device="cuda:0"); st_out = torch.empty_like(st)
    gm.allgather_dev(st.data_ptr(), st_out.data_ptr(), st.numel(), as_int32=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, z) and torch.equal(st_out, st)
    gm.close()


def test_bench_two_ranks_plumbing():
    """bench.py's N > 1 path (torchrun, barrier + max-over-ranks timing, all-gather of the trajectory chunk, one JSON line from rank 0) on a
    one-GPU box: two ranks share device 0 and gloo carries the collectives (--backend gloo); with DOJO_BENCH_GATHER=library-force
    the library's RCCL communicator is tried first and must fall back cleanly (RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DOJO_BENCH_GATHER="library-force", DOJO_BENCH_COMM_TIMEOUT="60")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512",
                        "--backend", "gloo"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["value"] > 0 and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 512 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]       # whole-job aggregate over both ranks
    tg = res["config"]["trajectory_gather"]                # the trajectory chunk [K, B, 13 Nb] of both ranks, not only the final state
    assert "torch.distributed" in tg["through"] and tg["bytes_received_per_rank"] == 2 * 3 * 512 * 13 * 13 * 4


def test_bench_strong_scaling_line():
    """BASELINE configs[3] in the form it is quoted in -- a FIXED total batch sharded over the ranks: `bench.py --config 4 --batch-total T --gpus 2`
    gives every rank T / 2 environments (contiguous slices), reports "scaling": "strong", the per-rank and the total batch, and value = T K / time
    (two ranks on the one GPU of the box, gloo).  A total that the ranks do not divide is refused."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "4", "--batch-total", "512",
                        "--backend", "gloo", "--no-parity", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["per_rank_batch"] == 256 and res["config"]["total_batch"] == 512
    assert abs(res["value"] - 512 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--config", "4", "--batch-total", "511", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and "not a multiple" in (r.stderr + r.stdout)


def test_bench_eight_ranks_strong_scaling_plumbing():
    """The 8-GPU line of BASELINE configs[3] as the driver will launch it -- `bench.py --config 4 --batch-total 8192 --gpus 8` -- with its eight ranks on the ONE
    GPU of this box and gloo carrying the collectives: rendezvous, every rank's contiguous slice of the one 8192-environment batch ([r 1024, (r + 1) 1024)),
    the all-gather of the trajectory chunk [K, 1024, 13 Nb] of every rank (each rank finds its own block at its own place), "scaling": "strong", the
    whole-job value.  No scaling curve can be measured here; this is so that the first run on eight GPUs is not spent on plumbing."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    K = 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(K), "--warmup", "1", "--config", "4", "--batch-total", "8192",
                        "--backend", "gloo", "--no-parity", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 8 and res["scaling"] == "strong" and cfg["per_rank_batch"] == 1024 and cfg["total_batch"] == 8192
    assert abs(res["value"] - 8192 * K / (res["ms_per_step"] * K * 1e-3)) < 1e-6 * res["value"]
    nb = 13                                                  # Quadruped: trunk + 12 links
    assert cfg["trajectory_gather"]["bytes_received_per_rank"] == 8 * K * 1024 * 13 * nb * 4
    assert len(cfg["rank_devices"]) == 8
    sl = sorted(cfg["rank_slices"], key=lambda x: x["rank"])
    assert [x["env_range"] for x in sl] == [[1024 * i, 1024 * (i + 1)] for i in range(8)]
    assert all(x["own_block_of_the_gather_matches"] for x in sl)
    assert len({x["first_state_checksum"] for x in sl}) == 8                 # eight different shards, not eight copies of one


def test_library_communicator_failure_modes():
    """dojo_comm_init / dojo_allgather_dev refuse what cannot work with a return code and a message instead of hanging or crashing: a rank outside the world,
    an empty world, no id, a gather without a communicator; a communicator of one rank reports itself through dojo_comm_info."""
    import ctypes as C
    import dojo_amd as d
    from dojo_amd import api
    lib = api.lib()
    gm = api.BatchedMechanism(d.baseline_config(1), 8, dtype="f64")
    uid = gm.comm_unique_id()
    INVALID = -1                                             # DOJO_ERR_INVALID (include/dojo_hip.h)
    for rank, world, idp in ((1, 1, uid), (0, 0, uid), (-1, 2, uid), (2, 2, uid), (0, 1, None)):
        rc = lib.dojo_comm_init(gm.h, rank, world, C.c_char_p(idp) if idp is not None else None)
        assert rc == INVALID and b"dojo_comm_init" in lib.dojo_last_error(), (rank, world, rc, lib.dojo_last_error())
    buf = (C.c_double * 8)()
    rc = lib.dojo_allgather_dev(gm.h, C.cast(buf, C.c_void_p), C.cast(buf, C.c_void_p), C.c_int64(8), 0, C.c_void_p(0))
    assert rc == INVALID and b"no communicator" in lib.dojo_last_error()
    rk, wd = C.c_int32(-7), C.c_int32(-7)
    assert lib.dojo_comm_info(gm.h, C.byref(rk), C.byref(wd)) == 0 and wd.value == 1          # no communicator: a world of one
    gm.comm_init(0, 1, uid)
    assert lib.dojo_comm_info(gm.h, C.byref(rk), C.byref(wd)) == 0 and (rk.value, wd.value) == (0, 1)
    gm.close()


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py re-executes itself as two ranks (torch.distributed.run on 127.0.0.1)
    and rank 0 prints the one JSON line with n_gpus = 2 and the device of every rank (gloo: the two ranks share the one GPU)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and len(res["config"]["rank_devices"]) == 2


def test_bench_refuses_more_gpus_than_the_node_has():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "GPU(s)" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


RCCL_WORKER = r"""
import os, sys
ROOT = %r
sys.path.insert(0, os.path.join(ROOT, "dojo.jl_amd", "host"))
import numpy as np, torch
import dojo_amd as d
from dojo_amd import api, distributed as D
rank, world, local = D.init_from_env(backend="nccl")
torch.cuda.set_device(local)
spec = d.baseline_config(4)
B = 1024
Z, U = d.synthetic_inputs(spec, B)
lo, hi = D.shard_slice(B, rank, world)
gm = api.BatchedMechanism(spec, hi - lo, dtype="f32", device=local)
D.connect_handle(gm, rank, world)                      # dojo_comm_unique_id on rank 0 -> dojo_comm_init on every rank
z = torch.tensor(Z[lo:hi], dtype=torch.float32, device="cuda:%%d" %% local); u = torch.tensor(U[lo:hi], dtype=torch.float32, device=z.device)
zn = torch.empty_like(z); st = torch.empty(hi - lo, dtype=torch.int32, device=z.device); it = torch.empty_like(st)
import ctypes as C
api._chk(api.lib().dojo_step_dev(gm.h, C.c_void_p(z.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(zn.data_ptr()), C.c_void_p(st.data_ptr()), C.c_void_p(it.data_ptr()), None, None,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
zg = D.all_gather_states_rccl(gm, zn, world); sg = D.all_gather_states_rccl(gm, st, world)     # dojo_allgather_dev over RCCL / xGMI
torch.cuda.synchronize()
gm.close()
if rank == 0:
    full = api.BatchedMechanism(spec, B, dtype="f32", device=0)
    zf, sf, itf = full.step(Z.astype(np.float32), U.astype(np.float32)); full.close()
    assert np.array_equal(zg.cpu().numpy(), zf) and np.array_equal(sg.cpu().numpy(), sf), "gathered != unsharded"
    print("RCCL_OK", world)
""" % ROOT


def test_library_rccl_allgather_two_ranks(tmp_path):
    """dojo_allgather_dev with TWO ranks on two GPUs: the sharded Quadruped step, gathered through the library's RCCL communicator,
    equals the unsharded batch bit for bit.  Needs a node with >= 2 GPUs (the boxes of the development pool have one: skipped there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % torch.cuda.device_count())
    f = tmp_path / "rccl_worker.py"
    f.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", str(f)], capture_output=True, text=True, timeout=900, env=env)
    assert "RCCL_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
