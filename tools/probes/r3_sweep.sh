# (streams, batch) sweep of the timed region only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O; cd $R
for cfg in "1 1" "1 2" "2 2" "2 3" "1 4" "3 2" "2 1"; do
  set -- $cfg
  python bench.py --streams $1 --batch $2 --steps 3 --warmup 1 --rollout-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $1 batch $2:', round(d['value'],2), 'fps', round(d['host_busy_cores'],2), 'cores', d.get('host_busy_cores_by_thread'))"
done
