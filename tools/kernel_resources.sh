#!/bin/bash
# scratch / spills / code size of the kernels in an object of dojo_kernels.hip: tools/kernel_resources.sh build/k_float_1_1.o
o=$1; t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $o $t/fb.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$t/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/k.co | grep -E "^ *\.name:|private_segment_fixed_size|vgpr_spill|group_segment" | sed 's/^ *//' | paste - - - - | sed 's/_ZN12_GLOBAL__N_1[0-9]*//; s/EEvN2dj10KernelArgsIT_T0_EE//' | cut -c1-200
/opt/rocm/lib/llvm/bin/llvm-objdump -d $t/k.co > $t/k.s
for k in dojo_step_kernel dojo_grad_kernel; do echo "$k instructions: $(awk -v k=$k '$0 ~ "<.*"k"I" {f=1; next} /^[0-9a-f]+ </{f=0} f' $t/k.s | wc -l)  scratch ops: $(awk -v k=$k '$0 ~ "<.*"k"I" {f=1; next} /^[0-9a-f]+ </{f=0} f' $t/k.s | grep -c scratch_)"; done
rm -rf $t
