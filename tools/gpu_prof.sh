#!/bin/bash
# Profile session of a round (ROUND=r05_a ...): bench line, rocprofv3 kernel-trace stats (grouped default and --chunks 1) of `bench.py --timed-only`
# (warmup + the timed region of the driver's run: --steps 20 --warmup 5, nothing else), PMC passes over the same window (--chunks 1), the other configurations.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${ROUND:-r05}
export TMPDIR=/tmp
echo "=== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/${ROUND:-r05}/bench_line.json; cut -c1-300 gpurun_out/${ROUND:-r05}/bench_line.json
for mode in grouped chunks1; do
  extra=""; [ $mode = chunks1 ] && extra="--chunks 1"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r05}/prof_$mode -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --timed-only $extra > $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r05}/prof_$mode.log 2>&1)
  f=$(find gpurun_out/${ROUND:-r05}/prof_$mode -name "*kernel_stats*.csv" | head -1); cp $f gpurun_out/${ROUND:-r05}/kernel_stats_$mode.csv; echo "== $mode"; head -6 $f | cut -c1-200
  t=$(find gpurun_out/${ROUND:-r05}/prof_$mode -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list); res = {}
for r in rows:
    n = r['Kernel_Name']
    k = next((x for x in ('dojo_stepc_kernel', 'dojo_stepp_kernel', 'dojo_gradp_kernel', 'dojo_step_kernel', 'dojo_grad_kernel') if x in n), None)
    if k: acc[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6); res[k] = (r.get('LDS_Block_Size', '?'), r.get('Scratch_Size', '?'), r.get('VGPR_Count', '?'), r.get('Accum_VGPR_Count', '?'), r.get('Grid_Size', '?'))
for k, v in acc.items(): print("   %-18s n=%4d mean %.3f ms min %.3f max %.3f | LDS %s scratch %s VGPR %s AGPR %s grid %s" % ((k, len(v), sum(v) / len(v), min(v), max(v)) + res[k]))
PY
  rm -rf gpurun_out/${ROUND:-r05}/prof_$mode
done
echo "=== pmc"; bash tools/gpu_pmc.sh > gpurun_out/${ROUND:-r05}/pmc.log 2>&1; cp gpurun_out/pmc/pmc_summary.txt gpurun_out/pmc/pmc_traffic.json gpurun_out/${ROUND:-r05}/; tail -3 gpurun_out/${ROUND:-r05}/pmc.log | cut -c1-400
rm -rf gpurun_out/pmc/SQ_* gpurun_out/pmc/FETCH* gpurun_out/pmc/WRITE*
echo "=== configs"; bash tools/gpu_configs_bench.sh 2>&1 | tee gpurun_out/${ROUND:-r05}/configs.txt
