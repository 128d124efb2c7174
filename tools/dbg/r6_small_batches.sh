#!/bin/bash
# round 6, end: the 8-GPU strong-scaling per-rank batches (Quadruped B = 1024, Atlas B = 256) against the number of environment groups
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps 20 --warmup $3 --distribution $4 --chunks $5 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2 $4 chunks $5:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'sync %d' % r['config']['sync_per_step_value'])"; }
for ch in 0 4 8 16; do run 4 1024 2 baseline $ch; done 2>&1 | tee gpurun_out/r06_f_small_batches.txt
for ch in 0 2 4 8; do run 5 256 12 standing $ch; done 2>&1 | tee -a gpurun_out/r06_f_small_batches.txt
for ch in 0 16; do run 3 1024 3 baseline $ch; done 2>&1 | tee -a gpurun_out/r06_f_small_batches.txt
