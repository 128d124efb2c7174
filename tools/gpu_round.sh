#!/bin/bash
# One GPU session: probes, parity tests, smoke, bench, rocprof kernel trace.  Output under gpurun_out/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
if [ "${PROBE:-all}" != "none" ]; then echo "=== probe"; timeout 900 python tools/gpu_probe.py ${PROBE:-all} 2>&1 | tail -30; fi
if [ "${TESTS:-1}" = "1" ]; then echo "=== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15; fi
if [ "${PMC:-0}" = "1" ]; then echo "=== pmc"; bash tools/gpu_pmc.sh 2>&1 | tail -40; fi
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
echo "=== bench"; timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_line.json
echo "=== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find gpurun_out/prof -name "*kernel_stats*.csv" | head -3
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -4 $f | cut -c1-300; done
