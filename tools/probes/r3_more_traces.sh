# kernel traces of the one-prompt configuration (passes paired) and of the VAE decode, final kernels of round 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1x1 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --batch 1 --rollout-only > $O/prof_1x1.json 2> $O/prof_1x1.err; echo "1x1 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vae -o run -- python3 $R/tools/microbench.py --what vae --iters 5 > $O/prof_vae.log 2>&1; echo "vae rc=$?"
find $O -name "*kernel_trace.csv" -delete
