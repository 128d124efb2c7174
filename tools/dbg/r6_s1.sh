#!/bin/bash
# round 6, GPU session 1: the row-layout ubench, parity of the row-layout factorization, A/B of the builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== ubench gj_rows"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/gj_rows.hip -o /tmp/gj_rows && /tmp/gj_rows | tee gpurun_out/r06_ubench_gj_rows.txt
echo "=== parity (default library: both layouts compiled, rows at run time)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_parity or gradient_parity or baseline_batch" 2>&1 | tail -6
echo "=== A/B"
ab() { DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/$1 python bench.py --no-cpu-baseline --no-parity "${@:3}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$2', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms']), 'sync', round(r['config'].get('sync_per_step_value') or 0))"; }
for rep in 1 2; do
  ab libdojo_hip_rows0.so "quad-only build      "
  ab libdojo_hip_rows1.so "rows-only build      "
  ab libdojo_hip.so       "both, rows at runtime"
  DOJO_ROWS=0 ab libdojo_hip.so "both, quad at runtime"
done 2>&1 | tee gpurun_out/r06_s1_ab.txt
echo "=== config 4 (Quadruped B=8192 / 1024)"
for lib in libdojo_hip_rows0.so libdojo_hip.so; do
  ab $lib "$lib cfg4 B8192" --config 4
  ab $lib "$lib cfg4 B1024" --config 4 --batch 1024
done 2>&1 | tee -a gpurun_out/r06_s1_ab.txt
