#!/bin/bash
# Atlas: what do the four-contact register arrays cost?  Same mechanism with one contact per foot on the MAXC = 1 build.
cd $GRAFT_REPO_ROOT
for one in 0 1; do
  DOJO_BENCH_ONE_CONTACT_PER_BODY=$one python bench.py --no-cpu-baseline --no-parity --config 5 --batch 2048 --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('atlas one_contact=$one:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms']), 'conv', r['config']['converged_fraction_last_step'], 'iters', r['config']['mean_newton_iters_last_step'])"
done
