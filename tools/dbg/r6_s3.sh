#!/bin/bash
# round 6, GPU session 3: is the asynchronous headline throughput- or chain-bound?  batch sweep, quad-only vs rows-only build, clocks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
ab() { DOJO_HIP_LIB=$GRAFT_REPO_ROOT/dojo.jl_amd/csrc/$1 python bench.py --no-cpu-baseline --no-parity "${@:3}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$2', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms']), 'sync', round(r['config'].get('sync_per_step_value') or 0))"; }
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' '; echo; sleep 2; done) > gpurun_out/r06_s3_clocks.txt &
SMI=$!
for B in 4096 8192 16384; do
  for lib in rows0 rows1; do
    ab libdojo_hip_$lib.so "$lib B=$B        " --batch $B
  done
done 2>&1 | tee gpurun_out/r06_s3_ab.txt
for lib in rows0 rows1; do ab libdojo_hip_$lib.so "$lib B=4096 chunks=1" --batch 4096 --chunks 1; ab libdojo_hip_$lib.so "$lib B=4096 chunks=8" --batch 4096 --chunks 8; done 2>&1 | tee -a gpurun_out/r06_s3_ab.txt
for lib in rows0 rows1; do GPU_MAX_HW_QUEUES=8 ab libdojo_hip_$lib.so "$lib B=4096 hwq=8" --batch 4096; done 2>&1 | tee -a gpurun_out/r06_s3_ab.txt
kill $SMI
sort gpurun_out/r06_s3_clocks.txt | uniq -c | sort -rn | head -12
