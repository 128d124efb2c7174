cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh oldpiv
echo "== hunt default tol (new pivot order, no refinement)"; python tools/hunt_parity.py 3 4096 6 default 2>&1 | tail -7
echo "== hunt default tol (old)"; DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/libdojo_hip_oldpiv.so python tools/hunt_parity.py 3 4096 6 default 2>&1 | tail -7
