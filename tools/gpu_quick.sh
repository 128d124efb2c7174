cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== pytest new"; timeout 900 python -m pytest tests -m gpu -q -k "translational or random_tree_mechanisms_gpu or raiberthopper" 2>&1 | tail -25
echo "=== bench"; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_line_a.json
