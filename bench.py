#!/usr/bin/env python
"""Benchmark of the Self-Forcing hot path on MI355X: decoded frames / s / node for the
chunk-wise autoregressive 4-step rollout (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full rollout call `CausalInferencePipeline.inference` on `--batch` prompts (default 2; `--streams`
such calls in flight, default 2; `value_one_stream` is ONE prompt alone on the GPU):
Wan2.1-T2V-1.3B-shape random-init weights, 832x480 (latent 60x104), 21 latent = 81 decoded frames,
3 frames per chunk, 4 warped denoising steps + 1 context pass per chunk = 35 DiT forwards,
synthetic T5 embeddings resident in HBM (BASELINE.json configs[1], "S1").  With N > 1 there is one process per GPU:
either the caller started them (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), or -- when
`--gpus N` is given and no torchrun environment is present -- this script starts them itself as child processes,
before it makes any GPU call, and passes rank 0's JSON line through.  Every rank rolls out its own prompts
(`rank, rank + N, ...`, the reference's DistributedSampler assignment, inference.py:96-100); RCCL carries only a
weight-checksum all-reduce, barriers, the timing reduction and one gather of per-rank figures -> weak scaling
(the protocol is `self_forcing_amd/distributed.py`, the code the world-size-2 gloo tests run; `--dist-selftest` runs
it on CPU without any GPU work).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel (self-attention over the KV cache) AT THE BATCH THE TIMED REGION RUNS: algorithmic
                  FLOPs per launch / its average launch duration, measured here with events on the launch stream,
                  against the 2.5 PFLOP/s dense bf16 MFMA peak and against the peak MEASURED on this box
                  (register-only MFMA loops under sustained clocks: `measured_peak`); the batch-1 figures beside it;
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
    """Socket power and reported shader clock of this rank's GPU during the timed region, sampled in a background thread.
    Read from the amdgpu sysfs files of the device (hwmon power1_average / power1_input, pp_dpm_sclk, power1_cap) -- plain
    file reads; `rocm-smi` is only the fallback: it is a Python program, and starting it every 0.5 s from a process with
    torch and a 300 GB GPU address space mapped cost 0.85 of a host core (measured, round 3).  Reported beside the rates
    because the MFMA kernels of this path run at the board's power limit; None when neither source
    is available."""

    def __init__(self, device_index, period=0.5):
        import glob
        import re
        import subprocess
        import threading
        self._stop, self.samples, self._period, self.limit_w = False, [], period, None
        self._sp, self._re = subprocess, re
        self._power_file = self._sclk_file = None
        self.source = None
        try:   # the PCI address of HIP device `device_index` -> its sysfs directory
            pr = torch.cuda.get_device_properties(device_index)
            dom = getattr(pr, "pci_domain_id", 0)
            pci = f"{dom:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            base = f"/sys/bus/pci/devices/{pci}"
            for name in ("power1_average", "power1_input"):
                hits = glob.glob(f"{base}/hwmon/hwmon*/{name}")
                if hits:
                    self._power_file = hits[0]
                    break
            if os.path.exists(f"{base}/pp_dpm_sclk"):
                self._sclk_file = f"{base}/pp_dpm_sclk"
            cap = glob.glob(f"{base}/hwmon/hwmon*/power1_cap")
            if cap:
                self.limit_w = int(open(cap[0]).read()) / 1e6
            if self._power_file and self._sclk_file:
                self._read_sysfs()            # (raises if unreadable)
                self.source = "amdgpu sysfs (hwmon power, pp_dpm_sclk)"
        except Exception:   # noqa: BLE001
            self.source = None
        if self.source is None:
            self._re_clk = re.compile(r"sclk clock level[^\n]*\((\d+)Mhz\)")
            self._re_pw = re.compile(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([0-9.]+)")
            self._cmd = ["rocm-smi", "--showclocks", "--showpower", "-d", str(device_index)]
            self._period = max(period, 2.0)
            try:
                cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)",
                                subprocess.run(["rocm-smi", "--showmaxpower", "-d", str(device_index)], capture_output=True, text=True, timeout=5).stdout)
                self.limit_w = float(cap.group(1)) if cap else None
                self.source = "rocm-smi (one process per sample)"
            except Exception:   # noqa: BLE001
                pass
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read_sysfs(self):
        watts = int(open(self._power_file).read()) / 1e6
        mhz = None
        for ln in open(self._sclk_file):
            if "*" in ln:
                m = self._re.search(r"(\d+)Mhz", ln)
                mhz = int(m.group(1)) if m else None
        return mhz, watts

    def _run(self):
        while not self._stop:
            try:
                if self._power_file and self._sclk_file and self.source and self.source.startswith("amdgpu"):
                    c, w = self._read_sysfs()
                    if c is not None:
                        self.samples.append((c, w))
                else:
                    out = self._sp.run(self._cmd, capture_output=True, text=True, timeout=5).stdout
                    c, w = self._re_clk.search(out), self._re_pw.search(out)
                    if c and w:
                        self.samples.append((int(c.group(1)), float(w.group(1))))
            except Exception:   # noqa: BLE001  (no source on this box: report nothing)
                return
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
                "sclk_reported_mhz_median": clk[len(clk) // 2], "socket_power_limit_w": self.limit_w, "source": self.source,
                "note": "sampled every ~0.5 s; in-kernel cycle counters see 1.5-1.7 GHz of effective clock under this load "
                        "(DESIGN.md section 4)"}


def thread_cpu_seconds():
    """{tid: (comm, user + system CPU seconds)} of every thread of this process (/proc/self/task): who uses the host."""
    import threading
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    py_threads = {t.native_id for t in threading.enumerate()}
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/stat") as f:
                    st = f.read()
                comm = st[st.index("(") + 1:st.rindex(")")]
                rest = st[st.rindex(")") + 2:].split()
                if int(tid) == os.getpid():
                    comm += " (main thread)"
                elif int(tid) not in py_threads:
                    comm += " (not a Python thread: HIP / HSA runtime helper)"
                out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / tick)
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out


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


def measured_peaks(dev):
    """Ceilings measured on THIS box: register-only bf16 MFMA loops on all 256 CUs (random operands, one
    wave per SIMD) and a 1 GiB streaming copy, both replayed from a HIP graph under sustained clocks."""
    g = torch.Generator(device="cpu").manual_seed(7)
    operands = torch.randn(4096, generator=g).to(torch.bfloat16).to(dev)
    sink = torch.empty(256 * 256, dtype=torch.float32, device=dev)
    out = {}
    for shp in ("32x32x16", "16x16x32"):
        fl = ops.probe_mfma(operands, sink, shape=shp, iters=2000)
        ms = time_graph(lambda: ops.probe_mfma(operands, sink, shape=shp, iters=2000), 4)
        out[f"mfma_{shp}_tflops"] = fl / (ms * 1e-3) / 1e12
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    ms = time_graph(lambda: ops.probe_copy(src, dst), 4)
    out["hbm_copy_GBps"] = 2.0 * src.numel() / (ms * 1e-3) / 1e9
    out["note"] = ("sf_probe_mfma: 256 workgroups x 4 waves, 4 independent accumulators, operands in registers, random bf16; "
                   "sf_probe_copy: 1 GiB read + 1 GiB written, 16 B per lane; graph replays that follow ~30 ms of the same replays")
    del src, dst
    return out


def committed_traffic(shape, n, lks, batch, window_frames):
    """HBM bytes per attention launch from the newest committed rocprofv3 --pmc passes under profiles/ -- only when
    they were taken at THIS shape (heads, queries, cache lengths, batch); else (None, why).  A counter run cannot share
    a process with the timed one (profiled passes run at another clock), so the field is READ, and labelled."""
    key = "attention_r64_kernel" if batch == 1 else f"attention_r64_kernel_batch{batch}"
    want = {"heads": shape.num_heads, "queries": n, "lk": [int(x) for x in lks], "batch": batch}
    if window_frames:
        return None, "rolling-window run: no committed counter pass at these cache lengths"
    for cand in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                rec = json.load(f)[key]
        except (OSError, KeyError, ValueError):
            continue
        have = rec.get("profiled_shape", {"heads": 12, "queries": 4680, "lk": [4680 * i for i in range(1, 8)], "batch": batch})
        if have != want:
            continue
        val = rec.get("hbm_bytes_per_launch")
        if val is None and "fetch_bytes_per_launch" in rec:     # (round-2 batch-2 record: fetch only; O is written once)
            val = rec["fetch_bytes_per_launch"] + 2 * batch * n * shape.dim
        return val, f"profiles/{cand} [{key}] (committed rocprofv3 --pmc pass at this shape, not measured in this run)"
    return None, "no committed counter pass at this shape / batch"


def roofline_leg(shape, dev, frames, nfpb, fs, window_frames=0, batch=1, peaks=None):
    """Average launch duration of the self-attention kernel over the Lk values one rollout visits
    (each chunk index launches it equally often), and of the FFN GEMMs, at `batch` prompts per launch -- the kernels
    the timed region runs -- and at batch 1: HIP events around launches replayed from a HIP graph on the launch stream
    (back-to-back device time, which is what rocprofv3's per-kernel average shows; eager launches through the operator
    layer leave gaps between kernels)."""
    H = shape.num_heads
    n = nfpb * fs
    g = torch.Generator(device="cpu").manual_seed(1)
    lk_max = (min(frames, window_frames) if window_frames > 0 else frames) * fs
    lks = [min((c + 1) * n, lk_max) for c in range(frames // nfpb)]
    meas_peak = max(peaks["mfma_32x32x16_tflops"], peaks["mfma_16x16x32_tflops"]) if peaks else None

    def attention_at(B):
        q = torch.randn(B, n, H, 128, generator=g).to(torch.bfloat16).to(dev)
        k = torch.randn(B, lk_max, H, 128, generator=g).to(torch.bfloat16).to(dev)
        v = torch.randn(B, lk_max, H, 128, generator=g).to(torch.bfloat16).to(dev)
        durs = [time_graph(lambda: ops.attention(q, k[:, :lk], v[:, :lk]), 10) for lk in lks]
        flops = [4.0 * shape.dim * n * lk * B for lk in lks]
        avg_ms, avg_flops = sum(durs) / len(durs), sum(flops) / len(flops)
        ach = avg_flops / (avg_ms * 1e-3) / 1e12
        traffic, src = committed_traffic(shape, n, lks, B, window_frames)
        rec = {"bound": "mfma", "kernel": "attention_r64_kernel (self-attention over the KV cache)", "batch": B,
               "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
               "traffic": traffic,
               "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, mean over the same cache lengths)",
               "traffic_source": src,
               "algorithmic_bytes_per_launch": 2 * B * (2 * n + 2 * sum(lks) / len(lks)) * shape.dim,   # (Q + O + K + V) bf16
               "avg_launch_ms": avg_ms, "flops_per_launch": avg_flops,
               "per_lk_tflops": {str(lk): f / (d * 1e-3) / 1e12 for lk, f, d in zip(lks, flops, durs)}}
        if meas_peak:
            rec["measured_peak"] = meas_peak
            rec["frac_of_measured_peak"] = ach / meas_peak
        return rec

    def gemms_at(B):
        M = B * n
        a = torch.randn(M, shape.dim, generator=g).to(torch.bfloat16).to(dev)
        w1 = (torch.randn(shape.ffn_dim, shape.dim, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        b1 = torch.zeros(shape.ffn_dim, dtype=torch.bfloat16, device=dev)
        hbuf = torch.empty(M, shape.ffn_dim, dtype=torch.bfloat16, device=dev)
        ms1 = time_graph(lambda: ops.gemm(a, w1, b1, epilogue="gelu", out=hbuf), 20)
        w2 = (torch.randn(shape.dim, shape.ffn_dim, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        b2 = torch.zeros(shape.dim, dtype=torch.bfloat16, device=dev)
        o2 = torch.empty(M, shape.dim, dtype=torch.bfloat16, device=dev)
        ms2 = time_graph(lambda: ops.gemm(hbuf, w2, b2, out=o2), 20)
        wq = (torch.randn(3 * shape.dim, shape.dim, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        bq = torch.zeros(3 * shape.dim, dtype=torch.bfloat16, device=dev)
        oq = torch.empty(M, 3 * shape.dim, dtype=torch.bfloat16, device=dev)
        ms3 = time_graph(lambda: ops.gemm(a, wq, bq, out=oq), 20)
        wo = (torch.randn(shape.dim, shape.dim, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        ms4 = time_graph(lambda: ops.gemm(a, wo, b2, epilogue="resid", resid=o2, out=torch.empty_like(o2)), 20)
        fl = 2.0 * M * shape.dim * shape.ffn_dim
        tf = lambda f, ms: f / (ms * 1e-3) / 1e12  # noqa: E731
        return {"M": M, "C": shape.dim, "ffn": shape.ffn_dim, "ffn0_tflops": tf(fl, ms1), "ffn2_tflops": tf(fl, ms2),
                "qkv_tflops": tf(2.0 * M * shape.dim * 3 * shape.dim, ms3), "proj_tflops": tf(2.0 * M * shape.dim * shape.dim, ms4),
                "ffn0_ms": ms1, "ffn2_ms": ms2, "qkv_ms": ms3, "proj_ms": ms4}

    att, gemm = attention_at(batch), gemms_at(batch)
    if batch != 1:
        att["at_batch_1"] = attention_at(1)
        gemm["at_batch_1"] = gemms_at(1)
    if peaks:
        att["measured_peaks"] = peaks
    return att, gemm


def hbm_bound_leg(shape, dev, nfpb, fs, batch=1, peaks=None):
    """The HBM-bound kernels of a forward (reported as GB/s): algorithmic bytes per launch / mean launch
    duration (50 launches replayed from a HIP graph, events around the replay: back-to-back device time including the
    ~1.5 us kernel boundary), against the 8 TB/s HBM3E peak and the streaming-copy rate measured on this box.  Shapes of
    one S1 forward at `batch` prompts per call."""
    HBM_PEAK = 8000.0
    n, C, H = batch * nfpb * fs, shape.dim, shape.num_heads
    groups = batch * nfpb
    g = torch.Generator(device="cpu").manual_seed(2)
    rb = lambda *s_: torch.randn(*s_, generator=g).to(torch.bfloat16).to(dev)  # noqa: E731
    x, mod, e0 = rb(n, C), rb(6, C), rb(groups, 6 * C)
    res = {"batch": batch}
    copy_peak = peaks.get("hbm_copy_GBps") if peaks else None
    if copy_peak:
        res["hbm_measured_peak_GBps"] = copy_peak

    def add(name, ms, nbytes, what):
        gbs = nbytes / (ms * 1e-3) / 1e9
        res[name] = {"GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK, "us": 1e3 * ms, "algorithmic_bytes": nbytes, "bytes": what}
        if copy_peak:
            res[name]["frac_of_measured_copy"] = gbs / copy_peak

    ms = time_graph(lambda: ops.layernorm_modulate(x, mod[0], mod[1], e0[:, :C], e0[:, C:2 * C], fs), 50)
    add("layernorm_kernel (LN + AdaLN modulate)", ms, 2 * n * C * 2, "x read + y written")
    qkv = rb(n, 3 * C)
    kc = torch.zeros(batch, nfpb * fs, H, 128, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    nq, nk = rb(C), rb(C)
    from self_forcing_amd.model import rope_tables
    cos, sin = (t.to(dev) for t in rope_tables(128))
    ms = time_graph(lambda: ops.qkv_norm_rope_cache(qkv, nq, nk, kc, vc, cos, sin, (nfpb, LAT_H // 2, LAT_W // 2), 0, 0), 50)
    add("qkv_norm_rope_cache_kernel (QK-RMSNorm + RoPE + cache append)", ms, 6 * n * C * 2, "qkv read; q, K rows, V rows written")
    xs, w6, b6 = rb(groups, C), rb(6 * C, C), rb(6 * C)
    ms = time_graph(lambda: ops.small_linear(xs, w6, b6, act_in="silu"), 50)
    add(f"small_linear_kernel (time projection, M = {groups})", ms, 6 * C * C * 2, "weights read once")
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
    s1_ref = None
    try:   # the kept one-off: the S1 clip itself on a GPU box's host cores (tools/cpu_s1_clip.py; too long for this leg)
        with open(os.path.join(ROOT, "profiles", "r03_cpu_s1_clip.json")) as f:
            r = json.load(f)
        s1_ref = {"value": r["frames_per_s"], "unit": "decoded frames/s", "seconds_per_clip": r["seconds"], "cores": r["threads"],
                  "cpu_model": r["cpu_model"], "measured": r["measured"], "source": "profiles/r03_cpu_s1_clip.json (committed; not measured in this run)"}
    except (OSError, KeyError, ValueError):
        pass
    return {"value": DECODED_PER_LATENT(2) / dt_T, "unit": "decoded frames/s", "cores": cores, "kind": "port", "s1_measured_ref": s1_ref,
            "sample": f"MEASURED: one whole config-T rollout of the CPU oracle (bf16, the reference's CPU path): 2 latent = 5 decoded "
                      f"frames, 2 chunks x (4 + 1) forwards of {fs} tokens, {dt_T:.1f} s",
            "rollout_seconds": dt_T,
            "s1_extrapolated": {"value": s1_fps, "unit": "decoded frames/s (upper bound, extrapolated)",
                                "sample": f"1 DiT forward, {frames_sample} latent frame(s) = {frames_sample * fs} tokens, empty KV cache, "
                                          f"bf16, {dt:.2f} s; an S1 rollout = {n_fwd:.0f} such forwards with growing cache "
                                          "(so the true CPU rate on S1 is lower)",
                                "forward_seconds": dt}}


def parse_args(argv=None):
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
    ap.add_argument("--rollout-only", action="store_true",
                    help="timed region only: no roofline / VAE / streaming / sampler / CPU legs (profiling runs)")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run the multi-rank protocol on CPU over gloo with no GPU work and print its report (tests)")
    ap.add_argument("--launch-timeout", type=float, default=None, help="seconds the self-launched rank processes may take")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl", help="collectives of the N > 1 protocol (nccl = RCCL over xGMI)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: every rank uses cuda:0 (needs --dist-backend gloo: RCCL "
                         "refuses two ranks on one device); the figures of such a run are not a scaling measurement")
    ap.add_argument("--no-pair", action="store_true",
                    help="one generator call per pass, as the reference (default: a chunk's context pass runs in ONE call with the "
                         "next chunk's first denoising pass -- same latents bit for bit, twice the rows per GEMM)")
    ap.add_argument("--max-inflight", type=int, default=None,
                    help="host pacing: passes a wrapper keeps enqueued before its thread sleeps (default: the wrapper's 2; 0 = unpaced)")
    return ap.parse_args(argv)


def main():
    global LAT_H, LAT_W
    a = parse_args()
    from self_forcing_amd import distributed as sfd
    if a.gpus > 1 and not sfd.launched_by_torchrun():
        # started the way the N = 1 bench is started: become the launcher.  The rank processes are fresh children of
        # torch.distributed.run; this process has made no GPU call (nothing above touches the device) and only waits.
        log(f"--gpus {a.gpus} without a torchrun environment: starting {a.gpus} rank processes")
        raise SystemExit(sfd.self_launch(os.path.abspath(__file__), sys.argv[1:], a.gpus, timeout=a.launch_timeout))
    # stdout carries ONE line, rank 0's JSON: libraries that write to file descriptor 1 on their own (RCCL prints a
    # five-line version banner there at init) are pointed at stderr for the rest of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    rank, local_rank, world = sfd.env_rank_world()
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or without torchrun)")
    if a.dist_selftest:
        rep = sfd.selftest(steps=a.steps, warmup=a.warmup)
        if rank == 0:
            emit(rep)
        return
    LAT_H, LAT_W = a.latent_height, a.latent_width
    if a.rollout_only:
        a.no_roofline = a.no_vae = a.no_cpu_baseline = True
        a.cfg_frames = 0

    if a.same_device and a.dist_backend == "nccl" and world > 1:
        raise SystemExit("--same-device needs --dist-backend gloo (RCCL refuses two ranks on one device)")
    dev = torch.device("cuda:0" if a.same_device else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    grp = sfd.RankGroup(backend=a.dist_backend, device=dev)     # RCCL over xGMI by default; no process group outside torchrun

    torch.set_num_threads(max(1, min(usable_cores() // max(1, world), 16)))
    log(f"rank {rank}/{world} on {dev}; synthesising weights")
    shape = sfa.NAMED_SHAPES[a.model]
    nfpb, step_list, shift = 3, [1000, 750, 500, 250], 5.0      # configs/self_forcing_dmd.yaml:9-18,56,69-70
    fs = (LAT_H // 2) * (LAT_W // 2)
    sd = sfa.synth_state_dict(shape, seed=0)                    # identical on every rank
    args = SimpleNamespace(denoising_step_list=step_list, warp_denoising_step=True, independent_first_frame=False,
                           num_frame_per_block=nfpb, context_noise=0)
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd, timestep_shift=shift, is_causal=True, device=dev,
                                  local_attn_size=a.local_attn_size, sink_size=a.sink_size)
    if a.max_inflight is not None:
        sfa.WanDiffusionWrapper.max_inflight_forwards = a.max_inflight
    window = a.local_attn_size if a.local_attn_size > 0 else 0
    enc = sfa.SyntheticTextEncoder(shape.text_len, shape.text_dim, device=dev)
    pool = sfa.RolloutPool(args, dev, gen, lambda: enc, sfa.IdentityVAE, streams=a.streams)
    for pipe in pool.pipes:
        pipe.pair_context_with_next = not a.no_pair

    # every rank must hold the same replica: a checksum of the device-resident weights, compared over RCCL
    grp.check_replicas(float(torch.stack([t.float().sum() for t in gen.model._keep[:64]]).sum().double().item()))

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
    # the board sampler starts a rocm-smi process every 0.5 s: inside the timed region only when this is the one rank
    # of the job; with several ranks it watches the warm-up instead (same kernels, no effect on the timed region)
    board = BoardSampler(dev.index or 0) if rank == 0 else None
    tw = time.perf_counter()
    if board is not None and world > 1:
        board.__enter__()
    # every stream's pipeline must have seen one rollout (cache / workspace allocation) before timing
    if a.warmup:
        pool.run_each(lambda pipe: one_step(pipe, 0))
        pool.run(list(range(min(a.streams, a.warmup), a.warmup)), one_step)
    torch.cuda.synchronize()
    if board is not None and world > 1:
        board.__exit__()
    log(f"warmup: {time.perf_counter() - tw:.2f} s")
    from self_forcing_amd.torch_ops import host_enqueue_stats
    host_enqueue_stats(reset=True)
    if board is not None and world == 1:
        board.__enter__()
    cpu0, thr0 = time.process_time(), thread_cpu_seconds()
    elapsed, local_elapsed, lats = grp.timed(lambda: pool.run(list(range(a.warmup, total)), one_step))
    proc_cpu = time.process_time() - cpu0                       # CPU seconds of ALL threads of this rank (incl. the sampler's)
    thr1 = thread_cpu_seconds()
    by_thread = sorted(((thr1[t][1] - thr0.get(t, (None, 0.0))[1], thr1[t][0]) for t in thr1), reverse=True)
    if board is not None and world == 1:
        board.__exit__()
    fw_calls, fw_secs, fw_cpu, fw_threads = host_enqueue_stats()
    log(f"timed {a.steps} steps in {local_elapsed:.2f} s (max over ranks {elapsed:.2f} s)")
    lat = lats[-1]
    assert torch.isfinite(lat.float()).all(), "non-finite latents"
    lat = lat[:1]                                               # the legs below work on ONE prompt's latents
    decoded = DECODED_PER_LATENT(a.frames)                      # per prompt
    per_rank = grp.gather([a.steps * B * decoded, local_elapsed, fw_calls, fw_cpu, proc_cpu])
    grp.finish()                                                # last collective; rank 0's legs below run alone
    if rank != 0:
        return

    fps = world * a.steps * B * decoded / elapsed
    flops = B * rollout_flops(shape, a.frames, nfpb, len(step_list), fs, window)
    flops_exec = B * rollout_flops(shape, a.frames, nfpb, len(step_list), fs, window, executed=True)
    # one rollout ALONE on the GPU (one stream): the same step, nothing in flight beside it
    one = None
    if (a.streams > 1 or B > 1) and not a.rollout_only:
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
                   "passes_paired": (not a.no_pair) and B * nfpb * fs <= pool.pipes[0].pair_max_rows,
                   "parallelism": f"prompt-sharded x{world}" + (" (REHEARSAL: all ranks on one GPU, gloo collectives)" if a.same_device and world > 1 else ""),
                   "streams_per_gpu": a.streams, "batch_per_rollout": B},
        "algorithmic_tflop_per_step": flops / 1e12,
        "executed_tflop_per_step": flops_exec / 1e12,
        "achieved_tflops_per_gpu": flops_exec * a.steps / elapsed / 1e12,
        "mfma_frac_end_to_end": flops_exec * a.steps / elapsed / 1e12 / MFMA_PEAK_TFLOPS,
        "flop_note": "algorithmic = the reference's (n_steps + 1) full forwards per chunk (SURVEY 8d); executed = minus what the "
                     "context passes skip behind the last layer's K/V write (cache_only); achieved / frac divide EXECUTED work",
        # host side of one rank (with no data-path collective only start-up skew and host contention can cost
        # scaling): wall time this rank's threads spent INSIDE sf_dit_forward (the C call that enqueues a pass's launches)
        "host_threads": a.streams,
        "host_enqueue_ms_per_forward": 1e3 * fw_cpu / max(1, fw_calls),
        "host_enqueue_wall_ms_per_forward": 1e3 * fw_secs / max(1, fw_calls),
        "host_busy_cores": proc_cpu / max(local_elapsed, 1e-9),
        "host_cores_available": usable_cores(),
        "host_busy_cores_by_thread": [{"thread": name, "busy_cores": round(sec / max(local_elapsed, 1e-9), 3)} for sec, name in by_thread[:6] if sec > 0],
        "host_note": f"{fw_calls} forwards enqueued by {fw_threads} Python thread(s) in the timed region on rank 0. "
                     "host_enqueue_ms_per_forward = CPU time of the calling thread inside sf_dit_forward (the C call that enqueues a "
                     "pass's ~430 launches); _wall_ = wall time inside it, which includes waiting for room in the stream's launch "
                     "queue because the host runs ahead of the GPU; host_busy_cores = CPU seconds of ALL threads of this rank's process "
                     "(Python, torch, HIP runtime, the rocm-smi sampler thread) / wall seconds of the timed region: what one rank asks "
                     "of the node's cores",
    }
    if world > 1:
        out["per_rank"] = {"frames": [r[0] for r in per_rank], "seconds": [r[1] for r in per_rank],
                           "host_enqueue_ms_per_forward": [1e3 * r[3] / max(1.0, r[2]) for r in per_rank],
                           "host_busy_cores": [r[4] / max(r[1], 1e-9) for r in per_rank]}
    if board is not None and board.summary() is not None:
        out["board"] = board.summary()
        out["board"]["sampled_during"] = "the timed region" if world == 1 else "the warm-up steps (several ranks: nothing but the rollout runs in the timed region)"
    if one is not None:
        out["value_one_stream"] = decoded / one
        out["ms_per_rollout_one_stream"] = 1e3 * one
        out["achieved_tflops_one_stream"] = flops_exec / B / one / 1e12
        out["one_stream_note"] = "ONE prompt (batch 1) rolled out alone on the GPU, one HIP stream: the latency configuration"
    if not a.no_roofline:
        log("measured ceilings + roofline leg")
        peaks = measured_peaks(dev)
        att, gemm = roofline_leg(shape, dev, a.frames, nfpb, fs, window, batch=B, peaks=peaks)
        out["roofline"] = att
        out["gemm"] = gemm
        out["mfma_frac_end_to_end_of_measured_peak"] = out["achieved_tflops_per_gpu"] / att["measured_peak"]
        out["hbm_measured_peak"] = {"value": peaks["hbm_copy_GBps"], "unit": "GB/s (read + write)", "kernel": "sf_probe_copy, 1 GiB, 4 x 16 B per lane in flight",
                                    "datasheet": 8000.0}
        out["hbm_bound"] = hbm_bound_leg(shape, dev, nfpb, fs, batch=B, peaks=peaks)
        if B != 1:
            out["hbm_bound"]["at_batch_1"] = hbm_bound_leg(shape, dev, nfpb, fs, batch=1, peaks=peaks)
    heavy = world == 1                  # the legs below describe one GPU; the scaling runs (N > 1) skip them
    if heavy and not a.no_vae:
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
    if heavy and not a.no_roofline:
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
                            "note": "one rollout alone on the GPU, each chunk denoised then decoded to pixels before it is yielded "
                                    "(the default; `overlap_decode=True` is an option that measured within noise of it)"}
        if not a.no_vae:   # decode of chunk k on a second HIP stream under the denoising of chunk k+1
            clip_s = {}
            for mode in (False, True, False, True):                  # interleaved A/B, best of two each
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                n_chunks = sum(1 for _ in pipe0.stream(noise, [prompts[0]], overlap_decode=mode))
                torch.cuda.synchronize()
                clip_s[mode] = min(clip_s.get(mode, 1e9), time.perf_counter() - t_start)
                assert n_chunks == len(ts)
            out["streaming"]["overlapped_decode_clip_fps"] = decoded / clip_s[True]
            out["streaming"]["serial_decode_clip_fps"] = decoded / clip_s[False]
    if heavy and not a.no_roofline and a.cfg_frames > 0:
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
    if heavy and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(shape, sd, nfpb, nfpb, len(step_list), a.frames)
    emit(out)


if __name__ == "__main__":
    main()
