#!/bin/bash
# Experimental variant of the library: only the Ant-type kernels (fp32 ABI, MAXC = 1, quad mapping) are recompiled with the
# given -D flags, everything else is linked from the last full build.  usage: tools/build_variant.sh NAME "-DDJ_X=1 ..."
set -e
cd "$(dirname "$0")/../dojo.jl_amd/csrc"
name=$1; flags=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DDJ_TIO=float -DDJ_MAXC=1 -DDJ_QUAD=1 -DDJ_TSD=0 $flags -c dojo_kernels.hip -o build/v_$name.o
objs=$(ls build/k_*.o build/host.o | grep -v "k_float_1_1.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdojo_hip_$name.so $objs build/v_$name.o
echo built libdojo_hip_$name.so
