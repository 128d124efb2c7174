#!/bin/bash
# Experimental library: the Ant kernel object (float ABI, 1 contact/body, quad mapping) rebuilt with extra
# flags, everything else linked from the regular build.  usage: build_variant.sh NAME [hipcc flags...]
# -> dojo.jl_amd/csrc/libdojo_hip_NAME.so  (select with DOJO_HIP_LIB=...)
set -e
name=$1; shift
cd "$(dirname "$0")/../dojo.jl_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DDJ_TIO=float -DDJ_MAXC=1 -DDJ_QUAD=1 "$@" -c dojo_kernels.hip -o build/v_${name}.o
objs=$(ls build/*.o | grep -v "k_float_1_1.o" | grep -v "/v_" | grep -v "_prof.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o libdojo_hip_${name}.so $objs build/v_${name}.o
echo built libdojo_hip_${name}.so
