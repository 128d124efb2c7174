#!/bin/bash
# kernel times of the five BASELINE configurations at their (per-GPU) batch sizes + the PMC passes of the bench: gpurun_out/configs.txt, gpurun_out/pmc/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
timeout 900 python tools/gpu_probe.py configs 2>&1 | tee gpurun_out/configs.txt | tail -20
if [ "${PMC:-1}" = "1" ]; then bash tools/gpu_pmc.sh 2>&1 | tail -30; fi
