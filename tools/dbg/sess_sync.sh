cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfgB in "3 1024" "3 512" "4 1024" "3 2048"; do
for a in "0 0" "-1 0" "0 0" "-1 0" "12 0" "20 0"; do echo "== cfg/B $cfgB cap/groups $a"; timeout 300 python tools/dbg/sync_probe.py $a 30 $cfgB 2>&1 | tail -1; done
done
