#!/bin/bash
# round 6: asynchronous and joined-per-step throughput against the number of environment groups (bench.py --chunks N = dojo_set_groups)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for n in 1 2 3 4 8 16; do
  python bench.py --no-cpu-baseline --no-parity --chunks $n 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('groups $n: async', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], '| joined', round(r['config'].get('sync_per_step_value') or 0), 'ms/step %.3f' % r['config']['sync_per_step_ms'])"
done; done 2>&1 | tee gpurun_out/r06_f_groups.txt
