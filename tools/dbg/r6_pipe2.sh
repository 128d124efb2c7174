#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps 20 --warmup $3 --distribution $4 --pipeline $5 --chunks $6 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2 $4 pipeline $5 chunks $6:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'])"; }
for p in 0 1 0 1; do run 5 256 12 standing $p 2; run 5 256 12 standing $p 4; run 5 512 12 standing $p 4; done 2>&1 | tee -a gpurun_out/r06_g_pipeline.txt
