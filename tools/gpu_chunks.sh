#!/bin/bash
# bench.py over the number of independent environment groups (same session, same box): gpurun_out/chunks.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
: > gpurun_out/chunks.txt
for rep in 1 2; do for c in ${CHUNKS:-8 16 24 32 64}; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --chunks $c 2>&1 | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('chunks $c', round(b['value']), b['ms_per_step'], b['roofline']['avg_kernel_ms'], b['roofline_second_kernel']['avg_kernel_ms'])" | tee -a gpurun_out/chunks.txt
done; done
