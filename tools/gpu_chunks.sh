#!/bin/bash
# environment groups of dojo_step_dev against throughput: asynchronous rollout (`value`) and a join after every step (`sync_per_step_value`)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in 1 2 3 4 6 8 16 32; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-parity --no-cpu-baseline --chunks $c 2>&1 | tail -1 | python -c "
import sys, json; r = json.loads(sys.stdin.readline()); print('chunks $c: async %.0f  sync-per-step %.0f  (ms/step %.3f / %.3f)' % (r['value'], r['config']['sync_per_step_value'], r['ms_per_step'], r['config']['sync_per_step_ms']))"
done
