#!/bin/bash
# round 3, session c: the whole GPU tier + the bench line (parity and CPU baseline legs included)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" | tail -60
echo "=== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r3_c_bench_line.json | cut -c1-600
