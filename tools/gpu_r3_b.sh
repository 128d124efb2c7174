#!/bin/bash
# round 3, session b: per-phase cycles of the IFT kernel, explicit-inverse sweeps against LU-form sweeps (-DDJ_PROF build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export DOJO_HIP_LIB=$PWD/dojo.jl_amd/csrc/libdojo_hip_prof.so
for lw in none 0; do
  if [ $lw = none ]; then unset DOJO_IFT_LU_W; else export DOJO_IFT_LU_W=$lw; fi
  echo "=== DOJO_IFT_LU_W=$lw"; timeout 600 python tools/gpu_probe.py phases 2>&1 | grep -A12 "grad=1"
done
