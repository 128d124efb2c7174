cd $GRAFT_REPO_ROOT
echo "== bench (full line)"; python bench.py 2>&1 | tail -1 > gpurun_out/bench_r2_d.json; python - <<'PY'
import json
r=json.load(open('gpurun_out/bench_r2_d.json'))
print(r['value'], r['ms_per_step'], 'step', r['roofline']['avg_kernel_ms'], 'frac', r['roofline']['frac'], 'ift', r['roofline_second_kernel']['avg_kernel_ms'])
print(json.dumps(r.get('grad_inf_err_vs_cpu'), indent=1)); print(r.get('cpu_baseline'))
PY
echo "== bench chunks 1"; python bench.py --no-cpu-baseline --no-parity --chunks 1 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
echo "== bench again"; python bench.py --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
