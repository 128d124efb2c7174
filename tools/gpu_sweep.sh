#!/bin/bash
# random-mechanism sweep on the GPU (tools/random_sweep.py): gpurun_out/sweep.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
SWEEP_TRA=${SWEEP_TRA:-1} timeout 1500 python tools/random_sweep.py ${S0:-2000} ${CNT:-300} 2>&1 | tee gpurun_out/sweep.txt | tail -40
