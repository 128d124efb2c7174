#!/bin/bash
# environment groups of dojo_step_dev at the BASELINE batch: throughput against the group count (same session)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for ch in 8 12 16 20 24; do
  python bench.py --no-cpu-baseline --no-parity --chunks $ch 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('chunks $ch', round(r['value']), 'ms/step %.3f' % r['ms_per_step'])"
done; done
