#!/bin/bash
# End-of-round check on a fresh box: build stamp, smoke(), the GPU test-suite, the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | grep -v amdgpu | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/tests.txt 2>&1; grep -n "passed\|failed\|rror" gpurun_out/final/tests.txt | head -5
timeout 900 python bench.py 2>gpurun_out/final/bench_err.txt | tail -1 > gpurun_out/final/bench_line.json; python -c "
import json; r=json.load(open('gpurun_out/final/bench_line.json')); print('bench', round(r['value']), r['ms_per_step'], r['roofline']['frac'], r['grad_inf_err_vs_cpu']['f32']['grad_inf_err_max'], r['cpu_baseline']['value'], r['cpu_baseline'].get('sparse_lu_flops'))"
