# HIP / HSA runtime knobs vs the helper thread that keeps ~0.85 of a host core busy, and vs throughput (timed region only)
R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$1 $2: "; env $1 python bench.py --steps 3 --warmup 1 --rollout-only $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'fps', round(d['host_busy_cores'],2), 'cores', round(d['host_enqueue_ms_per_forward'],2), 'ms cpu/fwd', round(d['host_enqueue_wall_ms_per_forward'],2), 'ms wall/fwd', [(t['thread'][:28], t['busy_cores']) for t in d.get('host_busy_cores_by_thread', [])][:3])"; }
run X=1 ""
run ROC_SIGNAL_POOL_SIZE=128 ""
run ROC_SIGNAL_POOL_SIZE=256 ""
run ROC_SIGNAL_POOL_SIZE=1024 ""
run ROC_SIGNAL_POOL_SIZE=4096 ""
run ROC_SIGNAL_POOL_SIZE=1024 "--max-inflight 0"
run ROC_SIGNAL_POOL_SIZE=4096 "--max-inflight 0"
run ROC_SIGNAL_POOL_SIZE=1024 "--max-inflight 4"
run X=1 "--max-inflight 0"
