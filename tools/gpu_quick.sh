#!/bin/bash
# quick GPU check: a pytest selection (K=...) and one bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== pytest -k '${K:-translational}'"; timeout 1200 python -m pytest tests -m gpu -q -k "${K:-translational}" 2>&1 | tail -${TAIL:-25}
if [ "${BENCH:-1}" = "1" ]; then echo "=== bench"; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_line_a.json | cut -c1-200; fi
