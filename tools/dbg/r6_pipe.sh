#!/bin/bash
# round 6, end: pipelined groups (dojo_set_async(h, 2): a group's IFT of step k next to its step kernel of step k + 1) on the small-batch lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
make -C oracle > /dev/null 2>&1
echo "=== test"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined" 2>&1 | grep -E "passed|failed|^E " | head -5
run() { python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps 20 --warmup $3 --distribution $4 --pipeline $5 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2 $4 pipeline $5:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'])"; }
for p in 0 1 0 1; do run 5 256 12 standing $p; run 4 1024 2 baseline $p; run 3 1024 3 baseline $p; run 3 4096 3 baseline $p; run 5 2048 12 standing $p; done 2>&1 | tee gpurun_out/r06_g_pipeline.txt
