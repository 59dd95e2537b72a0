# sf_dit_forward_pair on / off, timed region only, alternating runs on one box
R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$*: "; python bench.py --steps 3 --warmup 1 --rollout-only $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'fps')"; }
for i in 1 2; do
  run --streams 1 --batch 1
  run --streams 1 --batch 1 --no-pair
  run --streams 2 --batch 2
  run --streams 2 --batch 2 --no-pair
  run --streams 1 --batch 2
  run --streams 1 --batch 2 --no-pair
done
