#!/bin/bash
# Round-2 session h: the GPU test-suite on the final build.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02h/tests_full.txt 2>&1; grep -n "passed\|failed\|rror\|assert" gpurun_out/r02h/tests_full.txt | head -30
