cd $GRAFT_REPO_ROOT
echo "== bench default refine"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'])"
echo "== bench refine inf"; python bench.py --no-cpu-baseline --refine inf 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'])"
echo "== bench norefine build"; DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/libdojo_hip_norefine.so python bench.py --no-cpu-baseline --refine inf 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'])"
echo "== bench default refine again"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms'])"
echo "== hunt default tol"; python tools/hunt_parity.py 3 4096 6 default 2>&1 | tail -8
echo "== hunt 1e-8"; python tools/hunt_parity.py 3 2048 14 1e-8 2>&1 | tail -16
