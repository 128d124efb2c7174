#!/bin/bash
# A/B of library variants inside one GPU call: usage r6_ab.sh "lib1 lib2 ..." [bench args]; prints headline, kernel means, joined-step value
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
libs=$1; shift
ab() { DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/$1 python bench.py --no-cpu-baseline --no-parity "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['single_launch']['dojo_step_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']['avg_kernel_ms']), 'sync', round(r['config'].get('sync_per_step_value') or 0))"; }
for rep in 1 2 3; do for l in $libs; do ab $l "$@"; done; done 2>&1 | tee gpurun_out/r06_ab_last.txt
