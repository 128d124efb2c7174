#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps 20 --warmup $3 --distribution $4 --chunks $5 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2 $4 chunks $5:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'sync %d' % r['config']['sync_per_step_value'])"; }
for rep in 1 2; do for ch in 0 16; do run 5 2048 12 standing $ch; run 5 2048 5 baseline $ch; run 3 2048 3 baseline $ch; run 3 512 3 baseline $ch; done; done 2>&1 | tee -a gpurun_out/r06_f_small_batches.txt
