#!/bin/bash
# round 3, session e: contacts split over the quad (MAXC >= 4 builds): the other BASELINE configurations + the GPU tier
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== configs"; bash tools/gpu_r2_f.sh 2>&1
echo "=== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -15
