#!/bin/bash
# round 6, GPU session 2: per-phase cycles of the step kernel with the quad / row factorization, A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
for v in rows0p rows1p; do echo "=== phases $v"; DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/libdojo_hip_$v.so timeout 600 python tools/gpu_probe.py phases 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_s2_phases.txt
echo "=== A/B"
ab() { DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/$1 python bench.py --no-cpu-baseline --no-parity "${@:3}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$2', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms']), 'sync', round(r['config'].get('sync_per_step_value') or 0))"; }
for rep in 1 2; do
  ab libdojo_hip_rows0.so "quad-only build      "
  ab libdojo_hip_rows1.so "rows-only build      "
  ab libdojo_hip.so       "both, rows at runtime"
done 2>&1 | tee gpurun_out/r06_s2_ab.txt
