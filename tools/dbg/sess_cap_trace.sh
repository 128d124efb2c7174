cd $GRAFT_REPO_ROOT
echo "== chunks 1 cap 16"; bash tools/gpu_trace.sh --iter-cap 16
echo "== chunks 1 cap 0"; bash tools/gpu_trace.sh --iter-cap 0
