cd $GRAFT_REPO_ROOT
for cfg in "5 2048" "5 256" "4 8192" "4 1024" "2 1024"; do set -- $cfg   # (5 256 and 4 1024: the per-GPU batches of the 8-GPU strong-scaling lines of BASELINE configs 5 and 4)
  python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['avg_kernel_ms'], r['roofline_second_kernel']['avg_kernel_ms']), 'conv', r['config']['converged_fraction_last_step'], 'iters', r['config']['mean_newton_iters_last_step'])"
done
python bench.py --no-cpu-baseline --no-parity --config 2 --batch 1024 --no-grad --io-dtype f64 --steps 20 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('block fwd f64 B1024:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'])"
