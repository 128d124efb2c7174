#!/bin/bash
# round 6, end: joined steps with the iteration cap (continuation kernel) against without, on the round's kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do for cap in -1 12 16 20; do
  python bench.py --no-cpu-baseline --no-parity --iter-cap $cap 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cap $cap: async', round(r['value']), '| joined', round(r['config'].get('sync_per_step_value') or 0), 'ms/step %.3f' % r['config']['sync_per_step_ms'])"
done; done 2>&1 | tee gpurun_out/r06_f_cap.txt
