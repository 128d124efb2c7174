set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -m2 gfx
make -C oracle > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python - <<'PY' 2>&1 | tail -20
import sys, time; sys.path.insert(0, "dojo.jl_amd/host")
import numpy as np, dojo_amd as d
from dojo_amd import api
for cfg, B, dt in ((3, 4096, "f32"), (3, 4096, "f64"), (2, 1024, "f64")):
    spec = d.baseline_config(cfg)
    Z, U = d.synthetic_inputs(spec, 64)
    Z = np.tile(Z, (B // 64, 1)); U = np.tile(U, (B // 64, 1))
    gm = api.BatchedMechanism(spec, B, dtype=dt)
    z = Z.astype(gm.np_dtype)
    for k in range(3):
        t = time.time(); zn, st, it = gm.step(z, U); el = time.time() - t
        print(spec.name, dt, "B", B, "step", k, "kernel ms %.3f" % gm.last_kernel_ms(), "wall %.3f" % el, "ok frac", (st == 0).mean(), "mean iters", it.mean(), flush=True)
        z = zn
    gm.close()
PY
