#!/bin/bash
# The GPU test-suite (with the parity summaries of the BASELINE-batch tests) on the current build.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 1700 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02h/tests_full.txt 2>&1; grep -n "passed\|failed\|rror\|BASELINE cfg" gpurun_out/r02h/tests_full.txt | head -30
