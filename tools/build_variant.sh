#!/bin/bash
# Experimental variant of the library: only ONE kernel object (default: the Ant-type kernels, fp32 ABI, MAXC = 1, quad mapping; the
# Atlas-type ones with VMAXC=4 VQUAD=2) is recompiled with the given -D flags, everything else is linked from the last full build.
# usage: [VTIO=float VMAXC=1 VQUAD=1] tools/build_variant.sh NAME "-DDJ_X=1 ..."
set -e
cd "$(dirname "$0")/../dojo.jl_amd/csrc"
name=$1; flags=$2
tio=${VTIO:-float}; mc=${VMAXC:-1}; qd=${VQUAD:-1}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DDJ_TIO=$tio -DDJ_MAXC=$mc -DDJ_QUAD=$qd -DDJ_TSD=0 $flags -c dojo_kernels.hip -o build/v_$name.o
objs=$(ls build/k_*.o build/host.o | grep -v "k_${tio}_${mc}_${qd}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdojo_hip_$name.so $objs build/v_$name.o
echo built libdojo_hip_$name.so
