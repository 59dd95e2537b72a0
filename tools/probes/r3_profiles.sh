# Round-3 profile passes (run through gpurun from the repo root): bench, the clean 1-call x 2-prompt kernel trace,
# FETCH_SIZE / WRITE_SIZE passes of the attention microbenchmark at 1 and 2 samples per launch.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
WHAT=${1:-all}
python $R/bench.py --steps 3 --warmup 1 --no-vae --cfg-frames 0 --no-cpu-baseline > $O/bench4.json 2> $O/bench4.err
echo bench done
if [ "$WHAT" = all ] || [ "$WHAT" = trace ]; then
rm -rf $O/prof_1x2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1x2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --batch 2 --rollout-only > $O/prof_1x2.json 2> $O/prof_1x2.err
echo trace done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
LKS=4680,9360,14040,18720,23400,28080,32760
for spec in "f1:FETCH_SIZE:1" "w1:WRITE_SIZE:1" "f2:FETCH_SIZE:2" "w2:WRITE_SIZE:2"; do
  IFS=: read name ctr b <<< "$spec"
  rm -rf $O/pmc_$name
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -o run -- python3 $R/tools/microbench.py --what attn --lk $LKS --iters 2 --batch $b > $O/pmc_$name.log 2>&1
  echo pmc $name done
done
fi
find $O -name "*kernel_trace.csv" -delete
