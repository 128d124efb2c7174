#!/bin/bash
# scratch / spills of the step kernel alone after an edit of the lane program (one variant with nothing but dojo_step_kernel instantiated:
# under a minute instead of the five of the whole object).  usage: [VTIO=float VMAXC=1 VQUAD=1] tools/step_only.sh ["-DDJ_X=1 ..."]
cd "$(dirname "$0")/../dojo.jl_amd/csrc"
t=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DDJ_TIO=${VTIO:-float} -DDJ_MAXC=${VMAXC:-1} -DDJ_QUAD=${VQUAD:-1} -DDJ_TSD=0 -DDJ_ONLY_STEP=1 $1 \
    -c dojo_kernels.hip -o $t/k.o || exit 1
bash ../../tools/kernel_resources.sh $t/k.o
rm -rf $t
