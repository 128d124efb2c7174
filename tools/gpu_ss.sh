#!/bin/bash
# body-body contact on the GPU, then the whole GPU tier and a short bench line (the device header changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== body-body contact"; timeout 900 python -m pytest tests/test_oracle_collisions.py tests/test_oracle_momentum.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15
echo "=== gpu tier"; timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_oracle_collisions.py --deselect tests/test_oracle_momentum.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "=== bench"; timeout 600 python bench.py --steps 20 --warmup 3 --no-parity --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
