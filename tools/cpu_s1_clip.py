#!/usr/bin/env python
"""The S1 CPU baseline as a kept one-off (">= 1 full 21-frame clip for the 1.3B config"; bench.py's
cpu_baseline leg is bounded to ~1 min and only extrapolates S1).  Runs the CPU oracle in bf16 mode -- the restatement of
the reference's CPU path, oracle/wan_oracle.py -- on BASELINE configs[1]: Wan-1.3B shape, latent [1, 21, 16, 60, 104],
3 frames per chunk, steps [1000, 750, 500, 250] warped at shift 5 + one context pass per chunk = 35 full forwards against a
cache that grows from 4680 to 32760 tokens.  TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/).

    python tools/cpu_s1_clip.py --mode full      --out profiles/r03_cpu_s1_clip.json   # the literal clip (tens of minutes)
    python tools/cpu_s1_clip.py --mode per-chunk --out ...                             # one forward per cache length, x 5

per-chunk: the five forwards of a chunk cost the same (same tokens, same cache length; the re-noise in between is
negligible), so ONE forward is timed at each of the 7 cache lengths -- with the earlier cache rows filled by running the
chunk's forward itself, i.e. the real K / V -- and the clip is 5 x their sum.  It fits one 20-minute box lease; `full`
does not on 16 host threads.
"""
import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import self_forcing_amd as sfa  # noqa: E402
from oracle import wan_oracle as wo  # noqa: E402


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def cpu_flags():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("flags"):
                fl = set(ln.split(":", 1)[1].split())
                return {k: (k in fl) for k in ("amx_bf16", "avx512_bf16", "avx512f", "avx2")}
    except OSError:
        pass
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["full", "per-chunk"], default="per-chunk")
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--frames", type=int, default=21)
    a = ap.parse_args()
    cores = a.threads or usable_cores()
    torch.set_num_threads(cores)
    shape = sfa.WAN_1_3B
    H, Wd, nfpb = 60, 104, 3
    fs = (H // 2) * (Wd // 2)
    t0 = time.time()
    sd = sfa.synth_state_dict(shape, seed=0)
    W = wo.prepare_weights(sd, torch.bfloat16)
    del sd
    cfg = wo.OracleConfig(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers, text_dim=shape.text_dim)
    print(f"[cpu_s1_clip] weights ready after {time.time() - t0:.0f} s; {cores} threads; mode {a.mode}", flush=True)
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(1, a.frames, 16, H, Wd, generator=g).to(torch.bfloat16)
    pe = torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16)
    pe[:, 77:] = 0
    n_chunks = a.frames // nfpb
    rec = {"workload": f"BASELINE configs[1] (S1): Wan-1.3B shape, latent [1, {a.frames}, 16, {H}, {Wd}], {nfpb} frames per chunk, 4 warped steps "
                       f"(shift 5) + 1 context pass per chunk = {5 * n_chunks} forwards of {nfpb * fs} tokens, cache 4680 .. {a.frames * fs} tokens",
           "implementation": "oracle/wan_oracle.py in bf16 mode (the CPU restatement of the reference's CPU path; random-init weights, seed 0)",
           "threads": cores, "cpu_model": cpu_model(), "cpu_flags": cpu_flags(), "host": platform.node(), "mode": a.mode,
           "decoded_frames": 1 + 4 * (a.frames - 1)}
    with torch.no_grad():
        if a.mode == "full":
            eps = [torch.randn(nfpb, 16, H, Wd, generator=g).to(torch.bfloat16) for _ in range(3 * n_chunks)]
            t1 = time.time()
            lat = wo.rollout(W, cfg, wo.RolloutArgs(num_frame_per_block=nfpb, timestep_shift=5.0), noise, pe, eps)
            dt = time.time() - t1
            assert torch.isfinite(lat.float()).all()
            rec.update(seconds=dt, frames_per_s=rec["decoded_frames"] / dt, measured="the whole clip, one run")
        else:
            sched = wo.FlowMatchTables(5.0)
            kv = wo.init_kv_cache(cfg, 1, a.frames * fs, torch.bfloat16)
            ca = wo.init_crossattn_cache(cfg, 1, torch.bfloat16)
            ts = torch.full((1, nfpb), 937.5)
            per = []
            for c in range(n_chunks):
                x = noise[:, c * nfpb:(c + 1) * nfpb]
                t1 = time.time()
                wo.wrapper_forward(W, cfg, sched, x, pe, ts, kv, ca, c * nfpb * fs)     # also leaves this chunk's K / V in the cache
                per.append(time.time() - t1)
                print(f"[cpu_s1_clip] chunk {c}: cache {(c + 1) * nfpb * fs} tokens, one forward {per[-1]:.1f} s", flush=True)
            dt = 5 * sum(per)
            rec.update(seconds=dt, frames_per_s=rec["decoded_frames"] / dt, forward_seconds_per_cache_length={str((c + 1) * nfpb * fs): round(t, 2) for c, t in enumerate(per)},
                       measured="ONE forward at each of the 7 cache lengths (real K / V of the earlier chunks in the cache); clip = 5 x their sum "
                                "(a chunk's 4 denoising passes and its context pass have the same shape and cache length)")
    print(json.dumps(rec), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
