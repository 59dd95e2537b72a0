#!/usr/bin/env python
"""Benchmark of the Self-Forcing hot path on MI355X: decoded frames / s / node for the
chunk-wise autoregressive 4-step rollout (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full rollout call `CausalInferencePipeline.inference` on `--batch` prompts (default 2; `--streams`
such calls in flight, default 2; `value_one_stream` is ONE prompt alone on the GPU):
Wan2.1-T2V-1.3B-shape random-init weights, 832x480 (latent 60x104), 21 latent = 81 decoded frames,
3 frames per chunk, 4 warped denoising steps + 1 context pass per chunk = 35 DiT forwards,
synthetic T5 embeddings resident in HBM (BASELINE.json configs[1], "S1").  With N > 1 (launched by
torch.distributed.run, one process per GPU) every rank rolls out its own prompts
(`rank, rank + N, ...`, the reference's DistributedSampler assignment, inference.py:96-100); RCCL
carries only a weight-checksum all-reduce, barriers and the timing reduction -> weak scaling.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (self-attention over the KV cache): algorithmic FLOPs per
                  launch / its average launch duration, measured here with events on the launch
                  stream, against the 2.5 PFLOP/s dense bf16 MFMA peak;
  cpu_baseline -- the CPU oracle (oracle/wan_oracle.py, bf16 mode = the reference's CPU path)
                  timed on this host's cores on a bounded sample (N = 1 only);
  vae_decode   -- SURVEY 8f-1, reported beside the metric (whose timed region is the DiT rollout,
                  SURVEY 8d): the Wan VAE decode of one clip's latents to 81 frames of 480x832 pixels,
                  and the literal rate of rollout + decode with the pixels left in HBM.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import self_forcing_amd as sfa  # noqa: E402
from self_forcing_amd import ops  # noqa: E402
from self_forcing_amd.sharding import shard_indices  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
LAT_H, LAT_W = 60, 104
DECODED_PER_LATENT = lambda f: 1 + 4 * (f - 1)  # noqa: E731  (Wan VAE temporal stride 4)


def log(msg):
    """progress on stderr (the JSON line on stdout stays alone)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class BoardSampler:
    """Socket power and reported shader clock of this rank's GPU during the timed region, sampled from `rocm-smi` in a
    background thread (one subprocess call per ~0.5 s: host-side only).  Reported beside the rates because the MFMA
    kernels of this path run at the board's power limit; None when rocm-smi is unavailable."""

    def __init__(self, device_index, period=0.5):
        import re
        import subprocess
        import threading
        self._re_clk = re.compile(r"sclk clock level[^\n]*\((\d+)Mhz\)")
        self._re_pw = re.compile(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([0-9.]+)")
        self._sp, self._stop, self.samples = subprocess, False, []
        self._cmd = ["rocm-smi", "--showclocks", "--showpower", "-d", str(device_index)]
        self._period = period
        self.limit_w = None
        try:
            cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)",
                            subprocess.run(["rocm-smi", "--showmaxpower", "-d", str(device_index)], capture_output=True, text=True, timeout=5).stdout)
            self.limit_w = float(cap.group(1)) if cap else None
        except Exception:   # noqa: BLE001
            pass
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop:
            try:
                out = self._sp.run(self._cmd, capture_output=True, text=True, timeout=5).stdout
            except Exception:   # noqa: BLE001  (no rocm-smi on this box: report nothing)
                return
            c, w = self._re_clk.search(out), self._re_pw.search(out)
            if c and w:
                self.samples.append((int(c.group(1)), float(w.group(1))))
            time.sleep(self._period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._th.join(timeout=10)

    def summary(self):
        if len(self.samples) < 3:
            return None
        clk, pw = sorted(c for c, _ in self.samples), sorted(w for _, w in self.samples)
        return {"samples": len(self.samples), "socket_power_w_median": pw[len(pw) // 2], "socket_power_w_max": pw[-1],
                "sclk_reported_mhz_median": clk[len(clk) // 2], "socket_power_limit_w": self.limit_w,
                "note": "rocm-smi during the timed region (every ~0.5 s; the limit from --showmaxpower); in-kernel cycle counters "
                        "see 1.5-1.7 GHz of effective clock under this load (DESIGN.md section 4)"}


def usable_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def rollout_flops(shape, frames, nfpb, n_steps, fs, window_frames=0, executed=False):
    """Algorithmic FLOPs of one rollout: linear + cross-attn per token, self-attn
    4*C*N*Lk per layer per forward; (n_steps + 1) forwards per chunk.  `window_frames` > 0: rolling-window mode,
    the keys a chunk attends to are capped at that many frames (causal_model.py:203-231).
    executed=True: what the HIP path really runs -- a chunk's context pass only updates the KV cache, so it returns
    after the LAST layer's K/V write (sf_forward_args.cache_only) and skips that layer's self-attention, output
    projection, cross-attention, FFN, and the head: nothing reads them (the reference computes and discards them)."""
    C, Fd, L = shape.dim, shape.ffn_dim, shape.num_layers
    per_tok = L * (12 * C * C + 4 * C * Fd + 4 * shape.text_len * C) + 2 * 64 * C + 2 * C * 64
    skipped_per_tok = (6 * C * C + 4 * C * Fd + 4 * shape.text_len * C) + 2 * C * 64   # o, cross q/o, cross-attn, ffn; head
    total = 0.0
    n = nfpb * fs
    for chunk in range(frames // nfpb):
        lk = (chunk + 1) * n
        if window_frames > 0:
            lk = min(lk, window_frames * fs)
        total += (n_steps + 1) * (per_tok * n + L * 4.0 * C * n * lk)
        if executed:
            total -= skipped_per_tok * n + 4.0 * C * n * lk
    return total


def time_graph(fn, iters):
    """Mean device time of `fn`'s kernels with the launches replayed from a HIP graph: for kernels of a few
    microseconds the Python call (allocation + dispatcher + ctypes) takes longer than the kernel, and timing eager
    launches measures the host.  The timed replays follow ~30 ms of the same replays WITHOUT a host synchronisation in
    between: after an idle gap the chip needs milliseconds to raise its clocks again (tools/probes/sustained.py: the
    ffn.0 GEMM takes 128-146 us right after 2 s of idle, 108-112 us under load), and a rollout never idles."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    e1.synchronize()
    t1 = max(e0.elapsed_time(e1), 1e-3)                 # ms per replay (cold clocks: an upper bound)
    n_warm, n_timed = max(1, int(30.0 / t1)), max(1, int(20.0 / t1))
    for _ in range(n_warm):
        graph.replay()
    e0.record()
    for _ in range(n_timed):
        graph.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * n_timed)


def roofline_leg(shape, dev, frames, nfpb, fs, window_frames=0):
    """Average launch duration of the self-attention kernel over the Lk values one rollout visits
    (each chunk index launches it equally often), and of the biggest GEMM: HIP events around launches
    replayed from a HIP graph on the launch stream (back-to-back device time, which is what rocprofv3's
    per-kernel average shows; eager launches through the operator layer leave gaps between kernels)."""
    H = shape.num_heads
    n = nfpb * fs
    g = torch.Generator(device="cpu").manual_seed(1)
    q = torch.randn(1, n, H, 128, generator=g).to(torch.bfloat16).to(dev)
    lk_max = (min(frames, window_frames) if window_frames > 0 else frames) * fs
    k = torch.randn(1, lk_max, H, 128, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(1, lk_max, H, 128, generator=g).to(torch.bfloat16).to(dev)
    durs, flops = [], []
    for chunk in range(frames // nfpb):
        lk = min((chunk + 1) * n, lk_max)
        ms = time_graph(lambda: ops.attention(q, k[:, :lk], v[:, :lk]), 10)
        durs.append(ms)
        flops.append(4.0 * shape.dim * n * lk)
    avg_ms = sum(durs) / len(durs)
    avg_flops = sum(flops) / len(flops)
    traffic = traffic_src = None
    for cand in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), same shapes -- a counter
            # run cannot share a process with the timed one (profiled passes run at another clock), so this field is
            # READ from the newest committed pass and labelled with its file
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                traffic = json.load(f)["attention_r64_kernel"]["hbm_bytes_per_launch"]
            traffic_src = f"profiles/{cand} (committed rocprofv3 --pmc pass, not measured in this run)"
            break
        except (OSError, KeyError, ValueError):
            pass
    att = {"bound": "mfma", "kernel": "attention_r64_kernel (self-attention over the KV cache)",
           "achieved": avg_flops / (avg_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": avg_flops / (avg_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": traffic,
           "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, mean over the same 7 cache lengths)",
           "traffic_source": traffic_src, "algorithmic_bytes_per_launch": 2 * (2 * n + 2 * sum(flops) / len(flops) / (4.0 * shape.dim * n)) * shape.dim,   # (Q + O + K + V) bf16
           "avg_launch_ms": avg_ms, "flops_per_launch": avg_flops,
           "per_lk_tflops": {str(min((i + 1) * n, lk_max)): flops[i] / (durs[i] * 1e-3) / 1e12 for i in range(len(durs))}}
    # GEMMs: ffn.0 (N = ffn_dim) and ffn.2 (K = ffn_dim) at M = n
    a = torch.randn(n, shape.dim, generator=g).to(torch.bfloat16).to(dev)
    w1 = (torch.randn(shape.ffn_dim, shape.dim, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b1 = torch.zeros(shape.ffn_dim, dtype=torch.bfloat16, device=dev)
    hbuf = torch.empty(n, shape.ffn_dim, dtype=torch.bfloat16, device=dev)
    ms1 = time_graph(lambda: ops.gemm(a, w1, b1, epilogue="gelu", out=hbuf), 20)
    w2 = (torch.randn(shape.dim, shape.ffn_dim, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    b2 = torch.zeros(shape.dim, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(n, shape.dim, dtype=torch.bfloat16, device=dev)
    ms2 = time_graph(lambda: ops.gemm(hbuf, w2, b2, out=o2), 20)
    fl = 2.0 * n * shape.dim * shape.ffn_dim
    gemm = {"ffn0_tflops": fl / (ms1 * 1e-3) / 1e12, "ffn2_tflops": fl / (ms2 * 1e-3) / 1e12,
            "ffn0_ms": ms1, "ffn2_ms": ms2, "M": n, "C": shape.dim, "ffn": shape.ffn_dim}
    return att, gemm


def hbm_bound_leg(shape, dev, nfpb, fs):
    """The HBM-bound kernels of a forward (reported as GB/s): algorithmic bytes per launch / mean launch
    duration (50 launches replayed from a HIP graph, events around the replay: back-to-back device time including the
    ~1.5 us kernel boundary), against the 8 TB/s HBM3E peak.  Shapes of one S1 forward."""
    HBM_PEAK = 8000.0
    n, C, H = nfpb * fs, shape.dim, shape.num_heads
    g = torch.Generator(device="cpu").manual_seed(2)
    rb = lambda *s_: torch.randn(*s_, generator=g).to(torch.bfloat16).to(dev)  # noqa: E731
    x, mod, e0 = rb(n, C), rb(6, C), rb(nfpb, 6 * C)
    res = {}

    def add(name, ms, nbytes, what):
        gbs = nbytes / (ms * 1e-3) / 1e9
        res[name] = {"GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK, "us": 1e3 * ms, "algorithmic_bytes": nbytes, "bytes": what}

    ms = time_graph(lambda: ops.layernorm_modulate(x, mod[0], mod[1], e0[:, :C], e0[:, C:2 * C], fs), 50)
    add("layernorm_kernel (LN + AdaLN modulate)", ms, 2 * n * C * 2, "x read + y written")
    qkv = rb(n, 3 * C)
    kc = torch.zeros(1, n, H, 128, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    nq, nk = rb(C), rb(C)
    from self_forcing_amd.model import rope_tables
    cos, sin = (t.to(dev) for t in rope_tables(128))
    ms = time_graph(lambda: ops.qkv_norm_rope_cache(qkv, nq, nk, kc, vc, cos, sin, (nfpb, LAT_H // 2, LAT_W // 2), 0, 0), 50)
    add("qkv_norm_rope_cache_kernel (QK-RMSNorm + RoPE + cache append)", ms, 6 * n * C * 2, "qkv read; q, K rows, V rows written")
    xs, w6, b6 = rb(nfpb, C), rb(6 * C, C), rb(6 * C)
    ms = time_graph(lambda: ops.small_linear(xs, w6, b6, act_in="silu"), 50)
    add("small_linear_kernel (time projection, M = 3)", ms, 6 * C * C * 2, "weights read once")
    return res


def cpu_baseline_leg(shape, sd, frames_sample, nfpb, n_steps, total_frames):
    """The CPU oracle in bf16 mode (the reference's CPU path: bf16 weights, SDPA-style attention)
    on this host's cores: ONE forward of `frames_sample` latent frames against an empty cache."""
    from oracle import wan_oracle as wo
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle forward on {cores} host threads")
    W = wo.prepare_weights(sd, torch.bfloat16)
    cfg = wo.OracleConfig(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers,
                          text_dim=shape.text_dim)
    fs = (LAT_H // 2) * (LAT_W // 2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, frames_sample, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
    pe = torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16)
    kv = wo.init_kv_cache(cfg, 1, frames_sample * fs, torch.bfloat16)
    ca = wo.init_crossattn_cache(cfg, 1, torch.bfloat16)
    sched = wo.FlowMatchTables(5.0)
    ts = torch.full((1, frames_sample), 937.5)
    with torch.no_grad():
        t0 = time.time()
        wo.wrapper_forward(W, cfg, sched, x, pe, ts, kv, ca, 0)
        dt = time.time() - t0
    # optimistic extrapolation: every one of the rollout's forwards costs what this empty-cache one does
    n_fwd = (total_frames // nfpb) * (n_steps + 1) * (nfpb / frames_sample)
    s1_fps = DECODED_PER_LATENT(total_frames) / (n_fwd * dt)
    # MEASURED: BASELINE configs[0] ("T": configs/tiny_test.yaml + few-step keys) as a whole rollout of the oracle:
    # independent first frame, 1 frame per block, shift 8, noise [1, 2, 16, 60, 104] = 2 chunks x (4 + 1) one-frame
    # forwards of 1560 tokens; 5 decoded frames
    Tn = torch.randn(1, 2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
    eps = [torch.randn(1, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(6)]
    targs = wo.RolloutArgs(num_frame_per_block=1, independent_first_frame=True, timestep_shift=8.0)
    log("cpu_baseline: measured config-T rollout (10 one-frame forwards)")
    with torch.no_grad():
        t0 = time.time()
        lat = wo.rollout(W, cfg, targs, Tn, pe, eps)
        dt_T = time.time() - t0
    assert torch.isfinite(lat.float()).all()
    return {"value": DECODED_PER_LATENT(2) / dt_T, "unit": "decoded frames/s", "cores": cores, "kind": "port",
            "sample": f"MEASURED: one whole config-T rollout of the CPU oracle (bf16, the reference's CPU path): 2 latent = 5 decoded "
                      f"frames, 2 chunks x (4 + 1) forwards of {fs} tokens, {dt_T:.1f} s",
            "rollout_seconds": dt_T,
            "s1_extrapolated": {"value": s1_fps, "unit": "decoded frames/s (upper bound, extrapolated)",
                                "sample": f"1 DiT forward, {frames_sample} latent frame(s) = {frames_sample * fs} tokens, empty KV cache, "
                                          f"bf16, {dt:.2f} s; an S1 rollout = {n_fwd:.0f} such forwards with growing cache "
                                          "(so the true CPU rate on S1 is lower)",
                                "forward_seconds": dt}}


def main():
    global LAT_H, LAT_W
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=21, help="latent frames per rollout")
    ap.add_argument("--model", default="Wan2.1-T2V-1.3B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE-decode leg")
    ap.add_argument("--profile", action="store_true", help="print the pipeline's per-chunk event timing")
    ap.add_argument("--latent-height", type=int, default=LAT_H, help="latent rows (60 = 480 px; 90 = 720 px)")
    ap.add_argument("--latent-width", type=int, default=LAT_W, help="latent columns (104 = 832 px; 160 = 1280 px)")
    ap.add_argument("--local-attn-size", type=int, default=-1,
                    help="rolling-window mode: KV cache of this many latent frames (long rollouts, BASELINE configs[3]); -1 = global")
    ap.add_argument("--sink-size", type=int, default=0, help="frames kept at the head of the rolling window")
    ap.add_argument("--cfg-frames", type=int, default=6,
                    help="latent frames of the 50-step CFG sampler leg (SURVEY 8f-4); 0 skips it, 21 = the whole clip (~20 s)")
    ap.add_argument("--streams", type=int, default=2, help="rollouts in flight per GPU (one HIP stream each, shared weights)")
    ap.add_argument("--batch", type=int, default=2, help="prompts per rollout call (batch dimension of every kernel)")
    a = ap.parse_args()
    LAT_H, LAT_W = a.latent_height, a.latent_width

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} (WORLD_SIZE={world})")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)   # RCCL over xGMI

    torch.set_num_threads(min(usable_cores(), 16))
    log(f"rank {rank}/{world} on {dev}; synthesising weights")
    shape = sfa.NAMED_SHAPES[a.model]
    nfpb, step_list, shift = 3, [1000, 750, 500, 250], 5.0      # configs/self_forcing_dmd.yaml:9-18,56,69-70
    fs = (LAT_H // 2) * (LAT_W // 2)
    sd = sfa.synth_state_dict(shape, seed=0)                    # identical on every rank
    args = SimpleNamespace(denoising_step_list=step_list, warp_denoising_step=True, independent_first_frame=False,
                           num_frame_per_block=nfpb, context_noise=0)
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd, timestep_shift=shift, is_causal=True, device=dev,
                                  local_attn_size=a.local_attn_size, sink_size=a.sink_size)
    window = a.local_attn_size if a.local_attn_size > 0 else 0
    enc = sfa.SyntheticTextEncoder(shape.text_len, shape.text_dim, device=dev)
    pool = sfa.RolloutPool(args, dev, gen, lambda: enc, sfa.IdentityVAE, streams=a.streams)

    if dist is not None:  # every rank must hold the same replica: compare a checksum over RCCL
        cs = torch.stack([t.float().sum() for t in gen.model._keep[:64]]).sum().reshape(1).double()
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert lo.item() == hi.item(), "weight replicas differ across ranks"

    total = a.warmup + a.steps
    B = a.batch
    idx = shard_indices(total * world * B, rank, world)         # rank r: r, r + W, r + 2W, ...
    prompts = [f"synthetic MovieGenVideoBench prompt #{i}" for i in idx]
    for i in range(total):
        enc(prompts[B * i:B * i + B])                           # embeddings resident in HBM before timing
    for p in prompts[:total]:
        enc([p])                                                # (the single-prompt legs)
    torch.manual_seed(0 + rank)                                 # set_seed(seed + rank), inference.py:45

    def one_step(pipe, i):
        noise = torch.randn([B, a.frames, 16, LAT_H, LAT_W], device=dev, dtype=torch.bfloat16)
        return pipe.inference(noise, prompts[B * i:B * i + B], return_latents=True, profile=a.profile and rank == 0)[1]

    log(f"model resident ({gen.model.param_bytes() / 1e9:.2f} GB); warmup x{a.warmup}, {a.streams} stream(s)")
    tw = time.perf_counter()
    # every stream's pipeline must have seen one rollout (cache / workspace allocation) before timing
    if a.warmup:
        pool.run_each(lambda pipe: one_step(pipe, 0))
        pool.run(list(range(min(a.streams, a.warmup), a.warmup)), one_step)
    torch.cuda.synchronize()
    log(f"warmup: {time.perf_counter() - tw:.2f} s")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    board = BoardSampler(local_rank) if rank == 0 else None
    if board is not None:
        board.__enter__()
    t0 = time.perf_counter()
    lats = pool.run(list(range(a.warmup, total)), one_step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if board is not None:
        board.__exit__()
    log(f"timed {a.steps} steps in {elapsed:.2f} s")
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    lat = lats[-1]
    assert torch.isfinite(lat.float()).all(), "non-finite latents"
    lat = lat[:1]                                               # the legs below work on ONE prompt's latents

    decoded = DECODED_PER_LATENT(a.frames)                      # per prompt
    fps = world * a.steps * B * decoded / elapsed
    flops = B * rollout_flops(shape, a.frames, nfpb, len(step_list), fs, window)
    flops_exec = B * rollout_flops(shape, a.frames, nfpb, len(step_list), fs, window, executed=True)
    # one rollout ALONE on the GPU (one stream): the same step, nothing in flight beside it
    one = None
    if rank == 0 and (a.streams > 1 or B > 1):
        def single(i):
            noise = torch.randn([1, a.frames, 16, LAT_H, LAT_W], device=dev, dtype=torch.bfloat16)
            return pool.pipes[0].inference(noise, [prompts[i]], return_latents=True)[1]
        single(0)                                               # (re)allocates this pipeline's caches for batch 1
        n_one = max(1, min(a.steps, 3))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_one):
            single(i)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t1) / n_one
    out = {
        "metric": "decoded frames/sec/node, Wan-1.3B 832x480 4-step AR rollout" if (a.model, LAT_H, LAT_W) == ("Wan2.1-T2V-1.3B", 60, 104)
                  else f"decoded frames/sec/node, {a.model} {8 * LAT_W}x{8 * LAT_H} 4-step AR rollout",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"S1: {a.model}-shape random-init weights, latent {a.frames}x16x{LAT_H}x{LAT_W} "
                               f"({decoded} decoded frames per prompt), {nfpb} frames/chunk, steps {step_list} warped (shift {shift}) "
                               f"+ 1 context pass per chunk, " + (f"rolling KV window of {window} frames (sink {a.sink_size}), " if window else "")
                               + f"{B} prompt(s) per rollout call, {a.streams} rollout call(s) in flight per GPU "
                               f"(one HIP stream each, shared weights), prompts sharded rank::N",
                   "forwards_per_step": (a.frames // nfpb) * (len(step_list) + 1), "tokens_per_chunk": nfpb * fs,
                   "parallelism": f"prompt-sharded x{world}", "streams_per_gpu": a.streams, "batch_per_rollout": B},
        "algorithmic_tflop_per_step": flops / 1e12,
        "executed_tflop_per_step": flops_exec / 1e12,
        "achieved_tflops_per_gpu": flops_exec * a.steps / elapsed / 1e12,
        "mfma_frac_end_to_end": flops_exec * a.steps / elapsed / 1e12 / MFMA_PEAK_TFLOPS,
        "flop_note": "algorithmic = the reference's (n_steps + 1) full forwards per chunk (SURVEY 8d); executed = minus what the "
                     "context passes skip behind the last layer's K/V write (cache_only); achieved / frac divide EXECUTED work",
    }
    if board is not None and board.summary() is not None:
        out["board"] = board.summary()
    if one is not None:
        out["value_one_stream"] = decoded / one
        out["ms_per_rollout_one_stream"] = 1e3 * one
        out["achieved_tflops_one_stream"] = flops_exec / B / one / 1e12
        out["one_stream_note"] = "ONE prompt (batch 1) rolled out alone on the GPU, one HIP stream: the latency configuration"
    if rank == 0 and not a.no_roofline:
        log("roofline leg")
        att, gemm = roofline_leg(shape, dev, a.frames, nfpb, fs, window)
        out["roofline"] = att
        out["gemm"] = gemm
        out["hbm_bound"] = hbm_bound_leg(shape, dev, nfpb, fs)
    if rank == 0 and not a.no_vae:
        # VAE decode: decode alone, then rollout + decode through the same pool
        log("vae decode leg")
        from self_forcing_amd import vae_weights as vw
        vsd = vw.synth_vae_state_dict(vw.WAN_VAE, seed=0)
        for pipe in pool.pipes:
            pipe.vae = sfa.WanVAEWrapper(vsd, device=dev)
        vae0 = pool.pipes[0].vae
        vae0.decode_to_pixel(lat)                                    # allocates state + scratch
        torch.cuda.synchronize()
        tv = time.perf_counter()
        for _ in range(2):
            pix = vae0.decode_to_pixel(lat)
        torch.cuda.synchronize()
        vae_s = (time.perf_counter() - tv) / 2
        assert torch.isfinite(pix).all() and pix.shape[1] == decoded
        vfl = vw.vae_decode_flops(vw.WAN_VAE, LAT_H, LAT_W, a.frames)

        def literal_step(pipe, i):
            noise = torch.randn([1, a.frames, 16, LAT_H, LAT_W], device=dev, dtype=torch.bfloat16)
            return pipe.inference(noise, [prompts[i]], return_latents=False)[0, -1, :, :2, :2].clone()

        pool.run_each(lambda pipe: literal_step(pipe, 0))
        torch.cuda.synchronize()
        tl = time.perf_counter()
        pool.run(list(range(a.warmup, total)), literal_step)
        torch.cuda.synchronize()
        lit_s = time.perf_counter() - tl
        out["vae_decode"] = {"ms_per_clip": 1e3 * vae_s, "frames_per_s": decoded / vae_s, "tflops": vfl / vae_s / 1e12,
                             "algorithmic_tflop_per_clip": vfl / 1e12, "pixels": f"{decoded}x3x{8 * LAT_H}x{8 * LAT_W} float32",
                             "rollout_plus_decode_frames_per_s": a.steps * decoded / lit_s,   # (batch 1 per call here)
                             "note": "Wan2.1 VAE decoder shape, random-init weights; not part of `value`, whose timed "
                                     "region is the DiT rollout (SURVEY 8d); the second rate is rollout + decode "
                                     "through the same streams, pixels left in HBM"}
    if rank == 0 and not a.no_roofline:
        # streaming boundary: chunk-at-a-time generation on one stream, wall time per chunk
        # (with the real VAE when the leg above installed it: the chunk's pixels are decoded before the yield)
        log("streaming leg")
        pipe0 = pool.pipes[0]
        noise = torch.randn([1, a.frames, 16, LAT_H, LAT_W], device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        ts, t_prev = [], time.perf_counter()
        for _idx, lat_chunk, _pix in pipe0.stream(noise, [prompts[0]]):
            torch.cuda.synchronize()
            now = time.perf_counter()
            ts.append(now - t_prev)
            t_prev = now
        steady = ts[1:] if len(ts) > 1 else ts
        out["streaming"] = {"first_chunk_ms": 1e3 * ts[0], "chunk_ms": [round(1e3 * t, 1) for t in ts],
                            "decoded_frames_per_chunk": 4 * nfpb, "steady_fps": 4 * nfpb * len(steady) / sum(steady),
                            "worst_chunk_fps": 4 * nfpb / max(steady), "realtime_playback_fps": 16,
                            "pixels_decoded": not a.no_vae,
                            "note": "one rollout alone on the GPU, each chunk denoised then decoded to pixels before it is yielded"}
        if not a.no_vae:   # decode of chunk k on a second HIP stream under the denoising of chunk k+1
            torch.cuda.synchronize()
            t_start = time.perf_counter()
            n_chunks = sum(1 for _ in pipe0.stream(noise, [prompts[0]], overlap_decode=True))
            torch.cuda.synchronize()
            out["streaming"]["overlapped_decode_clip_fps"] = decoded / (time.perf_counter() - t_start)
            out["streaming"]["serial_decode_clip_fps"] = decoded / sum(ts)
            assert n_chunks == len(ts)
    if rank == 0 and not a.no_roofline and a.cfg_frames > 0:
        # 50-step UniPC + classifier-free guidance over the same generator: 2 x 50 + 2 forwards per chunk
        log("cfg sampler leg")
        cargs = SimpleNamespace(num_train_timestep=1000, timestep_shift=shift, independent_first_frame=False,
                                num_frame_per_block=nfpb, negative_prompt="synthetic negative prompt", guidance_scale=3.0)
        cfg_frames = max(nfpb, a.cfg_frames // nfpb * nfpb)
        cfg_res = {}
        for overlap in (True, False):
            cpipe = sfa.CausalDiffusionInferencePipeline(cargs, dev, generator=gen, text_encoder=enc, vae=sfa.IdentityVAE(),
                                                         overlap_cfg=overlap)
            cnoise = torch.randn([1, cfg_frames, 16, LAT_H, LAT_W], device=dev, dtype=torch.bfloat16)
            cpipe.sampling_steps = 2
            cpipe.inference(cnoise, [prompts[0]], None, None, None)          # caches + workspaces
            cpipe.sampling_steps = 50
            torch.cuda.synchronize()
            tc = time.perf_counter()
            clat = cpipe.inference(cnoise, [prompts[0]], None, None, None, return_latents=True)[1]
            torch.cuda.synchronize()
            cfg_res[overlap] = time.perf_counter() - tc
            assert torch.isfinite(clat.float()).all()
            del cpipe
        n_fw = (cfg_frames // nfpb) * (2 * 50 + 2)
        cdec = DECODED_PER_LATENT(cfg_frames)
        out["cfg_sampler"] = {"latent_frames": cfg_frames, "decoded_frames": cdec, "forwards": n_fw, "sampling_steps": 50,
                              "guidance_scale": 3.0, "seconds": cfg_res[True], "seconds_one_stream": cfg_res[False],
                              "ms_per_forward": 1e3 * cfg_res[True] / n_fw, "frames_per_s": cdec / cfg_res[True],
                              "note": "CausalDiffusionInferencePipeline: UniPC (order 2) + guidance, prompt / negative-prompt "
                                      "passes on two HIP streams (`seconds_one_stream`: back to back); first "
                                      f"{cfg_frames} latent frames of a clip, so the cache is at most {cfg_frames * fs} tokens long"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(shape, sd, nfpb, nfpb, len(step_list), a.frames)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
