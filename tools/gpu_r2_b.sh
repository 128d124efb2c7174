cd $GRAFT_REPO_ROOT
for w in inf 1e8 1e7 1e6; do echo "== hunt default tol refine $w"; HUNT_REFINE_W=$w python tools/hunt_parity.py 3 4096 4 default 2>&1 | tail -5; done
