# Round-3 final evidence pass: GPU tests, default bench, clean 1 x 2 kernel trace, SQ counter pass (MFMA utilisation)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -3 $O/gputest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1x2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --batch 2 --rollout-only > $O/prof_1x2.json 2> $O/prof_1x2.err; echo "trace rc=$?"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_b1 -o run -- python3 $R/tools/microbench.py --what attn,gemm --lk 4680,18720,32760 --iters 2 > $O/pmc_sq_b1.log 2>&1; echo "sq b1 rc=$?"
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_b2 -o run -- python3 $R/tools/microbench.py --what attn,gemm --lk 4680,18720,32760 --iters 2 --batch 2 --n 4680 > $O/pmc_sq_b2.log 2>&1; echo "sq b2 rc=$?"
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_m9360 -o run -- python3 $R/tools/microbench.py --what gemm --n 9360 --iters 2 > $O/pmc_sq_m9360.log 2>&1; echo "sq m9360 rc=$?"
find $O -name "*kernel_trace.csv" -delete
