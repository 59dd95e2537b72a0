# BASELINE configs[3] (long context) and configs[4] (14B / 720p, per GPU) through bench.py; results -> gpurun_out/r3/
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
python bench.py --frames 42 --steps 2 --warmup 1 --no-vae --cfg-frames 0 --no-cpu-baseline > $O/long_global.json 2> $O/long_global.err; echo long-global done
python bench.py --frames 42 --local-attn-size 21 --sink-size 1 --steps 2 --warmup 1 --no-vae --cfg-frames 0 --no-cpu-baseline > $O/long_window.json 2> $O/long_window.err; echo long-window done
python bench.py --model Wan2.1-T2V-14B --latent-height 90 --latent-width 160 --streams 1 --batch 1 --steps 1 --warmup 1 --no-vae --cfg-frames 0 --no-cpu-baseline > $O/x14b.json 2> $O/x14b.err; echo 14b done
