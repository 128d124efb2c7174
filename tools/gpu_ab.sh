# A/B of kernel variants inside one gpurun call (boxes of the pool differ by up to 1.4x): usage gpu_ab.sh "NAME1 NAME2 ..." [bench args]
# compares dojo.jl_amd/csrc/libdojo_hip.so with libdojo_hip_NAME.so (tools/build_variant.sh), two rounds
cd $GRAFT_REPO_ROOT
names=$1; shift
for rep in 1 2; do
  for lib in "" $names; do
    [ -n "$lib" ] && lib="_$lib"
    l=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/libdojo_hip$lib.so
    DOJO_HIP_LIB=$l python bench.py --no-cpu-baseline --no-parity "$@" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('lib$lib', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['single_launch']['dojo_step_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']['avg_kernel_ms']))"
  done
done
