#!/bin/bash
# N > 1 plumbing of bench.py on a one-GPU box: two ranks share device 0 (gloo carries the collectives); with
# DOJO_BENCH_GATHER=library-force the library's RCCL communicator is tried and must fall back (RCCL refuses two ranks on one device).
cd $GRAFT_REPO_ROOT
for pref in torch library-force; do
  echo "=== $pref"
  DOJO_BENCH_GATHER=$pref DOJO_BENCH_COMM_TIMEOUT=60 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 1 --batch 1024 --backend gloo 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*\|OMP_NUM" | tail -8 | cut -c1-600
done
