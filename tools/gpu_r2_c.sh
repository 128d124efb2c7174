cd $GRAFT_REPO_ROOT
echo "== bench"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'])"
echo "== hunt 1e-8 (policy default)"; python tools/hunt_parity.py 3 2048 14 1e-8 2>&1 | tail -16
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
