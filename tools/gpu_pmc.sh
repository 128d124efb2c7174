#!/bin/bash
# PMC passes (separate from --kernel-trace/--stats as the guide prescribes): instruction mix + HBM traffic of the step kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline}"
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  echo "== $set -> $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in rows:
    if 'dojo_step' not in r['Kernel_Name']: continue
    acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for c, d in acc.items():
    v = list(d.values()); print("  %-24s per-dispatch mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
done
