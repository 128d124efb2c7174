cd $GRAFT_REPO_ROOT
# (5 256 and 4 1024: the per-GPU batches of the 8-GPU strong-scaling lines of BASELINE configs 5 and 4.)
# Atlas on BOTH input distributions (round 6): "baseline" = BASELINE.md section 3's perturbation, bench.py's default -- the robot is thrown onto its foot
# edges and 5-9 % of the reference's own solves stall; "standing" = around the reference's initialize_atlas! pose (dojo_amd.coords._SYNTH_STANDING), where warmup 12 =
# the landing on the eight coplanar foot contacts is over and "warmup 0, 10 steps" times exactly those landing steps.
for cfg in "5 2048 5 20 baseline" "5 256 5 20 baseline" "5 2048 12 20 standing" "5 256 12 20 standing" "5 2048 0 10 standing" "4 8192 2 10 baseline" "4 1024 2 10 baseline" "2 1024 2 10 baseline"; do set -- $cfg
  python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps $4 --warmup $3 --distribution $5 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2 $5 (warmup $3, $4 steps):', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'step %.3f ift %.3f' % (r['roofline']['single_launch']['dojo_step_kernel']['avg_kernel_ms'], r['roofline']['single_launch']['dojo_grad_kernel']['avg_kernel_ms']), 'sync %d' % r['config']['sync_per_step_value'], 'conv', r['config']['converged_fraction_last_step'], 'iters', r['config']['mean_newton_iters_last_step'])"
done
# Atlas at the small per-rank batches with pipelined groups (dojo_set_async(h, 2): a group's IFT kernel of step k next to its step kernel of step k + 1, two groups):
# where the batch leaves SIMDs idle the two kernels of a step overlap
for cfg in "5 256 12 20 standing 2" "5 512 12 20 standing 4" "5 256 5 20 baseline 2"; do set -- $cfg
  python bench.py --no-cpu-baseline --no-parity --config $1 --batch $2 --steps $4 --warmup $3 --distribution $5 --pipeline 1 --chunks $6 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('cfg $1 B $2 $5 PIPELINED, $6 groups (warmup $3, $4 steps):', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'conv', r['config']['converged_fraction_last_step'], 'iters', r['config']['mean_newton_iters_last_step'])"
done
python bench.py --no-cpu-baseline --no-parity --config 2 --batch 1024 --no-grad --io-dtype f64 --steps 20 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('block fwd f64 B1024:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'])"
