#!/bin/bash
# round 3, session a: LU-form IFT sweeps -- parity hunt at the BASELINE batch and kernel times against the explicit-inverse sweeps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
for lw in inf none; do
  if [ $lw = none ]; then unset DOJO_IFT_LU_W; else export DOJO_IFT_LU_W=$lw; fi
  echo "=== DOJO_IFT_LU_W=$lw bench f32"; timeout 600 python bench.py --steps 20 --warmup 3 --no-parity --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.readline()); print('value %.0f ms/step %.3f step_kernel %.3f ift_kernel %.3f (best %.3f / %.3f)' % (r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'], r['roofline']['best_launch']['kernel_ms'], r['roofline_second_kernel']['best_launch']['kernel_ms']))"
  echo "=== DOJO_IFT_LU_W=$lw bench f64"; timeout 600 python bench.py --steps 20 --warmup 3 --no-parity --no-cpu-baseline --io-dtype f64 2>&1 | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.readline()); print('value %.0f ms/step %.3f step_kernel %.3f ift_kernel %.3f (best %.3f / %.3f)' % (r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'], r['roofline']['best_launch']['kernel_ms'], r['roofline_second_kernel']['best_launch']['kernel_ms']))"
done
unset DOJO_IFT_LU_W
echo "=== hunt LU form, default tolerances, 4096 x 9 steps"; timeout 1500 python tools/hunt_parity.py 3 4096 3 default 2>&1 | tail -12
cp gpurun_out/hunt_cfg3_tol0.npz gpurun_out/hunt_lu_cfg3_tol0.npz 2>/dev/null
echo "=== gpu tests (gradient parity subset) with LU form"; timeout 1200 python -m pytest tests -m gpu -q -x -k "gradient or golden or baseline_batch or contact or minimal" 2>&1 | tail -8
