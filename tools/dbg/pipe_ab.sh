cd $GRAFT_REPO_ROOT
# pipelined groups (dojo_set_async(h, 2)) against plain asynchronous groups: group counts, shared IFT streams (DOJO_PIPE_STREAMS), hardware queues
run() { python bench.py --no-cpu-baseline --no-parity --pipeline $1 --chunks $2 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('pipeline $1 chunks $2 ift-streams ${DOJO_PIPE_STREAMS:-4} queues ${GPU_MAX_HW_QUEUES:-default}:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'], 'sync', round(r['config']['sync_per_step_value']))"; }
run 0 16
DOJO_PIPE_STREAMS=4 run 1 16
DOJO_PIPE_STREAMS=2 run 1 16
DOJO_PIPE_STREAMS=4 run 1 14
DOJO_PIPE_STREAMS=4 run 1 12
DOJO_PIPE_STREAMS=8 run 1 12
DOJO_PIPE_STREAMS=1 run 1 16
run 0 16
