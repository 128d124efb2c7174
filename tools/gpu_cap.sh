#!/bin/bash
# iteration cap + continuation kernel: the bit-identity tests, then bench lines (asynchronous value and sync_per_step_value) against the cap
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
echo "=== pytest iteration cap"; timeout 1500 python -m pytest tests -m gpu -q -x -k "${K:-iteration_cap}" 2>&1 | tail -${TAIL:-15}
for cap in ${CAPS:-0 -1 20 24 0 -1}; do
  echo "=== bench --iter-cap $cap"
  timeout 600 python bench.py --steps ${STEPS:-20} --warmup 3 --iter-cap $cap --no-parity --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/cap_$cap.json
  python - <<PY
import json
try:
    r = json.load(open("gpurun_out/cap_$cap.json"))
    print("cap $cap: value %.0f  ms/step %.3f  sync %.0f (%.3f ms)  step_kernel %.3f ms (best %.3f)  ift %.3f" % (r["value"], r["ms_per_step"], r["config"]["sync_per_step_value"], r["config"]["sync_per_step_ms"], r['roofline']['single_launch']['dojo_step_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_step_kernel']['best_launch']['kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']["avg_kernel_ms"]))
except Exception as e:
    print("cap $cap: no line", e); print(open("gpurun_out/cap_$cap.json").read()[-600:])
PY
done
