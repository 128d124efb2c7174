#!/bin/bash
# round 6, second session: the speculative first trial (DJ_SPEC) -- per-phase cycles and A/B against the build without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
for v in $PROF_LIBS; do echo "=== phases $v"; DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/libdojo_hip_$v.so timeout 600 python tools/gpu_probe.py phases 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_spec_phases.txt
bash tools/dbg/r6_ab.sh "$AB_LIBS"
cp gpurun_out/r06_ab_last.txt gpurun_out/r06_spec_ab.txt
