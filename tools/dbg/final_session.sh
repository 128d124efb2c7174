cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
make -C oracle >/dev/null 2>&1
mkdir -p gpurun_out/r05_c
echo "=== tests"; timeout 200 python -m pytest tests/test_oracle_collisions.py -m gpu -q --tb=short -k "ball_on_atlas or ends_of_a_chain or world" 2>&1 | tail -5
echo "=== pmc"; timeout 400 bash tools/gpu_pmc.sh > gpurun_out/r05_c/pmc.log 2>&1; cp gpurun_out/pmc/pmc_summary.txt gpurun_out/pmc/pmc_traffic.json gpurun_out/r05_c/; tail -1 gpurun_out/r05_c/pmc.log | cut -c1-300
rm -rf gpurun_out/pmc/SQ_* gpurun_out/pmc/FETCH* gpurun_out/pmc/WRITE*
echo "=== bench (with the new counts in place)"; cp gpurun_out/r05_c/pmc_traffic.json profiles/r05_c_pmc_traffic.json; cp gpurun_out/r05_c/pmc_summary.txt profiles/r05_c_pmc_summary.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r05_c/bench_line.json; cut -c1-220 gpurun_out/r05_c/bench_line.json
