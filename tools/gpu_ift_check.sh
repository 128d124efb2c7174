#!/bin/bash
# round 3, session d: kernel times and HBM traffic of the IFT kernel after a change (bench lines f32 / f64, FETCH_SIZE / WRITE_SIZE passes, gradient tests)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
for dt in f32 f64; do
  echo "=== bench $dt"; timeout 600 python bench.py --steps 20 --warmup 3 --no-parity --no-cpu-baseline --io-dtype $dt 2>&1 | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.readline()); print('value %.0f sync %.0f ms/step %.3f step_kernel %.3f ift_kernel %.3f (best %.3f / %.3f)' % (r['value'], r['config']['sync_per_step_value'], r['ms_per_step'], r['roofline']['single_launch']['dojo_step_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_step_kernel']['best_launch']['kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']['best_launch']['kernel_ms']))"
done
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_d/$set -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --chunks 1 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_d/$set -name "*counter_collection.csv" | head -1)
  python3 - "$f" $set <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    kn = 'step' if 'dojo_step_kernel' in r['Kernel_Name'] else 'grad' if 'dojo_grad_kernel' in r['Kernel_Name'] else None
    if kn: acc[kn][r['Dispatch_Id']] += float(r['Counter_Value'])
for kn, d in acc.items(): print("  %s %s per-dispatch mean %.4g KB" % (kn, sys.argv[2], sum(d.values()) / len(d)))
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_d
cd $GRAFT_REPO_ROOT
echo "=== gpu tests (subset)"; timeout 1200 python -m pytest tests -m gpu -q -x -k "gradient or golden or baseline_batch or contact" 2>&1 | grep -v amdgpu.ids | tail -5
