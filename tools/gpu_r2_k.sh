#!/bin/bash
# A/B inside one session: one ContactEval live at a time in evaluate() (new) against the previous build (old).
cd $GRAFT_REPO_ROOT
echo "=== Ant"; bash tools/gpu_ab.sh old
echo "=== Atlas B=2048"; bash tools/gpu_ab.sh old --config 5 --batch 2048 --steps 10 --warmup 2
echo "=== Block B=1024 fwd+grad"; bash tools/gpu_ab.sh old --config 2 --batch 1024 --steps 20 --warmup 3
