#!/bin/bash
# kernel-trace of one short bench run (--chunks 1): per-kernel mean durations and launch resources.  usage: gpu_trace.sh [bench flags]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --chunks 1 "$@" > /tmp/trace.log 2>&1)
t=$(find /tmp/trace_out -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list); res = {}
for r in rows:
    n = r['Kernel_Name']
    k = next((x for x in ('dojo_stepc_kernel', 'dojo_stepp_kernel', 'dojo_gradp_kernel', 'dojo_step_kernel', 'dojo_grad_kernel', 'dojo_sweep_kernel') if x in n), None)
    if k: acc[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6); res[k] = (r.get('LDS_Block_Size', '?'), r.get('Scratch_Size', '?'), r.get('VGPR_Count', '?'), r.get('Accum_VGPR_Count', '?'), r.get('Grid_Size', '?'))
for k, v in acc.items(): print("   %-18s n=%4d mean %.3f ms min %.3f max %.3f | LDS %s scratch %s VGPR %s AGPR %s grid %s" % ((k, len(v), sum(v) / len(v), min(v), max(v)) + res[k]))
PY
rm -rf /tmp/trace_out
