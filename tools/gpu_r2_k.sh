#!/bin/bash
# A/B inside one session: current build (lib) against the previous one (lib_old).
cd $GRAFT_REPO_ROOT
echo "=== Ant"; bash tools/gpu_ab.sh old
echo "=== Ant, one launch per kernel"; bash tools/gpu_ab.sh old --chunks 1
echo "=== Quadruped B=8192"; bash tools/gpu_ab.sh old --config 4 --batch 8192 --steps 10 --warmup 2
echo "=== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_batch or forward_parity or gradient_parity" 2>&1 | tail -3
