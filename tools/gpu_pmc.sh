#!/bin/bash
# PMC passes (separate from --kernel-trace/--stats, as the MI355X guide prescribes): instruction mix, stall
# breakdown and HBM traffic of the two kernels of the step.  Output: gpurun_out/pmc/*.csv, pmc_summary.txt and
# pmc_traffic.json (copy to profiles/<round>_pmc_traffic.json: bench.py reports it as roofline.traffic).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
# one launch = the whole batch of 4096 environments (--chunks 1); --timed-only: warmup + the timed region of the driver's bench run (--steps 20
# --warmup 5) and nothing else, and only the dispatches of the TIMED steps (the last PMC_STEPS per kernel) enter the means below
PMC_STEPS=${PMC_STEPS:-20}; export PMC_STEPS
ARGS="${BENCH_ARGS:---steps $PMC_STEPS --warmup 5 --timed-only --chunks 1}"; export BENCH_ARGS_USED="$ARGS"
cd /tmp
: > $GRAFT_REPO_ROOT/gpurun_out/pmc/pmc_summary.txt
IFS=';' read -ra EXTRA <<< "${PMC_EXTRA_SETS:-}"      # further passes, ';'-separated sets of counters
# PMC_EXTRA_SETS: further passes (e.g. the memory-stall counters of the IFT kernel, review item 4 of round 5): a pass whose counters this GPU does not have fails alone
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "FETCH_SIZE" "WRITE_SIZE" "${EXTRA[@]}"; do
  [ -z "$set" ] && continue
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  echo "== $set" | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc/pmc_summary.txt
  python3 - "$f" <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc/pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    kn = 'dojo_step_kernel' if 'dojo_step_kernel' in r['Kernel_Name'] else 'dojo_grad_kernel' if 'dojo_grad_kernel' in r['Kernel_Name'] else None
    if kn is None: continue
    acc[(kn, r['Counter_Name'])][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
import os
nst = int(os.environ.get('PMC_STEPS', '20'))
for (kn, c), d in sorted(acc.items()):
    v = [d[k] for k in sorted(d)][-nst:]          # the timed steps: the last PMC_STEPS dispatches of this kernel
    print("  %-18s %-24s per-dispatch mean %.6g  (n=%d)" % (kn, c, sum(v)/len(v), len(v)))
PY
done
python3 - <<'PY'
import re, json, os, sys
root = os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, root)
from __graft_entry__ import build_info
digest = build_info().get("library_digest")      # content hash of the sources the loaded library was built from (bench.py checks it)
nst = int(os.environ.get('PMC_STEPS', '20'))
txt = open(os.path.join(root, 'gpurun_out/pmc/pmc_summary.txt')).read()
out = {}
for kn in ('dojo_step_kernel', 'dojo_grad_kernel'):
    f = re.search(kn + r'\s+FETCH_SIZE\s+per-dispatch mean ([0-9.e+-]+)', txt); w = re.search(kn + r'\s+WRITE_SIZE\s+per-dispatch mean ([0-9.e+-]+)', txt)
    if f and w:
        fk, wk = float(f.group(1)), float(w.group(1))
        # rocprofv3 reports KB; gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated, taken as is
        out[kn] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "bytes_per_launch": (2 * fk + wk) * 1024.0, "envs_per_launch": 4096,
                   "library_digest": digest, "step_window": "the %d timed steps behind 5 warmup steps of the closed-loop rollout (the window bench.py times)" % nst,
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB, per launch, mean over the timed steps' dispatches of bench.py " + os.environ.get('BENCH_ARGS_USED', '')}
    c = {n: re.search(kn + r'\s+SQ_INSTS_VALU_' + n + r'_F64\s+per-dispatch mean ([0-9.e+-]+)', txt) for n in ('FMA', 'ADD', 'MUL', 'TRANS')}
    if kn in out and all(c.values()):
        v = {n: float(m.group(1)) for n, m in c.items()}
        # wave instructions x 64 lanes, FMA = 2 flops (the expression rocprofv3 itself uses for SQ_INSTS_VALU_FLOPS_FP64)
        out[kn].update({"fp64_wave_instructions": v, "fp64_flops_per_launch": (2 * v['FMA'] + v['ADD'] + v['MUL'] + v['TRANS']) * 64.0})
json.dump(out, open(os.path.join(root, 'gpurun_out/pmc/pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(out))
PY
