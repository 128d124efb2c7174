cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/icache
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_WAIT_INST|INST_LEVEL|SQC_.*INST|SQ_INSTS_VALU |SQ_BUSY_CU" | head -40 > gpurun_out/icache/avail.txt
cat gpurun_out/icache/avail.txt | cut -c1-200
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/icache/$name -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --timed-only --chunks 1 > $GRAFT_REPO_ROOT/gpurun_out/icache/$name.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/icache/$name -name "*counter_collection.csv" | head -1)
  echo "== $set"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    kn = 'dojo_step_kernel' if 'dojo_step_kernel' in r['Kernel_Name'] else 'dojo_grad_kernel' if 'dojo_grad_kernel' in r['Kernel_Name'] else None
    if kn is None: continue
    acc[(kn, r['Counter_Name'])][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
for (kn, c), d in sorted(acc.items()):
    v = [d[k] for k in sorted(d)]
    print("  %-18s %-28s per-dispatch mean %.6g  (n=%d)" % (kn, c, sum(v)/len(v), len(v)))
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/icache/$name
done
