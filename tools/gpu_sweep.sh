# GPU regression sweep: random tree mechanisms against the oracle (tools/random_sweep.py), then the whole GPU tier.
cd $GRAFT_REPO_ROOT; make -C oracle >/dev/null 2>&1
echo "== sweep, solver tolerance 1e-9"; timeout 900 python tools/random_sweep.py 3000 200 2>&1 | tail -8
echo "== sweep, solver tolerance 1e-6"; SWEEP_TOL=1e-6 timeout 900 python tools/random_sweep.py 3000 200 2>&1 | tail -8
echo "== translational features, 1e-9"; SWEEP_TRA=1 timeout 900 python tools/random_sweep.py 4000 150 2>&1 | tail -8
echo "== gpu tier"; timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
