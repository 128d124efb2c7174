#!/bin/bash
# dispatch order: the test, the probe over group settings, bench.py A/B in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_dispatch.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dispatch_order or pipelined_groups or baseline_batch_distinct" 2>&1 | tail -3 >> $O
timeout 600 python tools/dbg/r6_dispatch.py 3 4096 >> $O 2>&1
timeout 300 python tools/dbg/r6_dispatch.py 4 8192 >> $O 2>&1
timeout 300 python tools/dbg/r6_dispatch.py 5 2048 standing >> $O 2>&1
for m in 0 1; do timeout 600 python bench.py --dispatch-order $m 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); sl = d['roofline']['single_launch']
print('bench --dispatch-order $m: value %.0f  sync_per_step %.0f (%.3f ms)  single launch step %.3f ms  ift %.3f ms' % (d['value'], d['config']['sync_per_step_value'], d['config']['sync_per_step_ms'], sl['dojo_step_kernel']['avg_kernel_ms'], sl['dojo_grad_kernel']['avg_kernel_ms']))" >> $O 2>&1; done
cat $O
