#!/bin/bash
# after an edit of dojo_hip.hip alone: recompile the host / C-ABI object, relink, and stamp the library with the sources' digest
cd "$(dirname "$0")/../dojo.jl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -c dojo_hip.hip -o build/host.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdojo_hip.so build/k_*.o build/host.o || exit 1
cd ../.. && python - <<'PY'
import glob, os, __graft_entry__ as g
C = g.CSRC
srcs = [os.path.join(C, "dojo_hip.hip"), os.path.join(C, "dojo_kernels.hip")] + sorted(glob.glob(os.path.join(C, "*.hpp"))) + [os.path.join(g.ROOT, "include", "dojo_hip.h")]
d = g._digest(srcs); open(os.path.join(C, "libdojo_hip.so.stamp"), "w").write(d + "\n"); g._build_log(os.path.join(C, "libdojo_hip.so"), "compiled", d); print("stamped", d[:16])
PY
