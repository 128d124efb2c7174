#!/bin/bash
# round 6: Atlas (two wavefronts per environment) with the factorization's level passes in the row layout against the quad layout (DOJO_ROWS=0), same library, same call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
run() { python bench.py --no-cpu-baseline --no-parity --config 5 --batch $2 --steps 20 --warmup $3 --distribution $4 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1 B $2 $4:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['single_launch']['dojo_step_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']['avg_kernel_ms']), 'sync %d' % r['config']['sync_per_step_value'], 'conv', r['config']['converged_fraction_last_step'], 'iters', r['config']['mean_newton_iters_last_step'])"; }
for rep in 1 2; do
  DOJO_ROWS=0 run "quad" 2048 5 baseline; run "rows" 2048 5 baseline
  DOJO_ROWS=0 run "quad" 2048 12 standing; run "rows" 2048 12 standing
done 2>&1 | tee gpurun_out/r06_e_atlas_rows.txt
DOJO_ROWS=0 run "quad" 256 12 standing | tee -a gpurun_out/r06_e_atlas_rows.txt; run "rows" 256 12 standing | tee -a gpurun_out/r06_e_atlas_rows.txt
echo "=== Atlas parity tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg5 or 5- or atlas or other_baseline" 2>&1 | grep -v amdgpu.ids | tail -4
