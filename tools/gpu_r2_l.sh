#!/bin/bash
# fp32-ABI gradient error at the BASELINE batch: y parked in an fp64 buffer of its own (lib) against the fp32 output slots (lib_nopark)
cd $GRAFT_REPO_ROOT
for v in "" _nopark; do
  echo "=== lib$v"; DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/libdojo_hip$v.so timeout 600 python tools/probe_f32.py 4096 2>&1 | grep -v amdgpu
done
bash tools/gpu_ab.sh "nopark" 2>&1 | grep -v amdgpu
