cd $GRAFT_REPO_ROOT
for q in "24 16" "40 16" "40 22" "40 32" "72 32" "72 64" "24 16"; do set -- $q
  GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu-baseline --no-parity --chunks $2 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('queues $1 chunks $2:', round(r['value']), 'ms/step %.3f' % r['ms_per_step'])"
done
