"""CPU tests of the host logic: cache index arithmetic, scheduler tables, prompt sharding, weight
handling, the C-ABI library's symbols, and the N>1 bench protocol over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from self_forcing_amd.kvcache import plan_cache_update
from self_forcing_amd.sharding import shard_indices, shard
from oracle import wan_oracle as wo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------- cache plan
def test_plan_matches_oracle_on_random_call_sequences():
    """plan_cache_update (product) == kv_cache_plan (oracle restatement of causal_model.py:202-236)
    over random chunk sequences incl. re-runs and rolling-window eviction."""
    rng = np.random.default_rng(0)
    for trial in range(200):
        fs = int(rng.integers(1, 40))
        las = int(rng.choice([-1, 2, 3, 5, 8]))
        sink = int(rng.integers(0, max(1, las))) if las != -1 else 0
        nf = int(rng.integers(1, 4))
        frames = int(rng.integers(nf, 12))
        cap = (las if las != -1 else frames + nf) * fs
        window = cap if las == -1 else las * fs
        if las != -1 and (nf > las - sink or nf + sink > las):
            continue
        le = ge = 0
        start = 0
        for _ in range(frames // nf):
            for rerun in range(int(rng.integers(1, 4))):
                n = nf * fs
                ev, keep, nle, ws, as_ = wo.kv_cache_plan(le, ge, start * fs, n, cap, las, sink * fs, window)
                p = plan_cache_update(le, ge, start * fs, n, cap, las, sink * fs, window)
                assert (p.evict, p.keep, p.local_end, p.write_start, p.attn_start) == (ev, keep, nle, ws, as_)
                assert p.global_end == start * fs + n
                le, ge = p.local_end, p.global_end
            start += nf


def test_plan_rolling_window_golden_indices():
    """local_attn_size=3, sink=1, one-frame chunks, 24 tokens/frame: the index trace the reference
    produced (tests/golden/modules_reduced.npz: sa_local_end / sa_global_end)."""
    mods = np.load(os.path.join(GOLD, "modules_reduced.npz"))
    fs, le, ge = 24, 0, 0
    for st, want_le, want_ge in zip(mods["sa_starts"], mods["sa_local_end"], mods["sa_global_end"]):
        p = plan_cache_update(le, ge, int(st) * fs, fs, 3 * fs, 3, 1 * fs, 3 * fs)
        le, ge = p.local_end, p.global_end
        assert (le, ge) == (int(want_le), int(want_ge))


def test_plan_overflow_in_global_mode_raises():
    with pytest.raises(RuntimeError, match="overflow"):
        plan_cache_update(local_end=48, global_end=48, current_start=48, num_new=24, capacity=48,
                          local_attn_size=-1, sink_tokens=0, max_attention_size=48)


def test_plan_rerun_overwrites_in_place():
    p1 = plan_cache_update(0, 0, 0, 10, 100, -1, 0, 100)
    p2 = plan_cache_update(p1.local_end, p1.global_end, 0, 10, 100, -1, 0, 100)
    assert (p1.write_start, p1.local_end) == (p2.write_start, p2.local_end) == (0, 10)


# -------------------------------------------------------------------------------- scheduler
@pytest.mark.parametrize("shift", [5.0, 8.0])
def test_scheduler_tables_match_reference_golden(shift):
    g = np.load(os.path.join(GOLD, "ops.npz"))
    s = sfa.FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
    s.set_timesteps(1000, training=True)
    tag = str(int(shift))
    assert np.array_equal(s.sigmas.numpy(), g[f"sched{tag}_sigmas"])
    assert np.array_equal(s.timesteps.numpy(), g[f"sched{tag}_timesteps"])
    table = torch.cat((s.timesteps, torch.tensor([0], dtype=torch.float32)))
    warped = table[1000 - torch.tensor([1000, 750, 500, 250])]
    assert np.array_equal(warped.numpy(), g[f"sched{tag}_warped"])


# --------------------------------------------------------------------------------- sharding
def test_shard_indices_distributed_sampler_semantics():
    """DistributedSampler(shuffle=False, drop_last=True): first W*floor(P/W) prompts, rank::W."""
    from torch.utils.data import DistributedSampler
    for P, W in [(64, 8), (10, 4), (7, 2), (3, 4), (1003, 8)]:
        data = list(range(P))
        used = set()
        for r in range(W):
            ref = list(DistributedSampler(data, num_replicas=W, rank=r, shuffle=False, drop_last=True))
            assert shard_indices(P, r, W) == ref
            used.update(ref)
        assert used == set(range(W * (P // W)))
    assert shard(["a", "b", "c", "d", "e"], 1, 2) == ["b", "d"]
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


# ---------------------------------------------------------------------------------- weights
def test_synth_weights_deterministic_and_complete():
    a = sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0)
    b = sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0)
    c = sfa.synth_state_dict(sfa.WAN_REDUCED, seed=1)
    assert set(a) == set(sfa.param_shapes(sfa.WAN_REDUCED))
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["head.head.weight"], c["head.head.weight"])
    assert a["head.head.weight"].abs().sum() > 0          # the reference zero-inits it (vacuous output)
    n13 = sum(int(np.prod(s)) for s in sfa.param_shapes(sfa.WAN_1_3B).values())
    assert abs(n13 - 1.419e9) < 2e6                        # 1.419 B without pose_proj


def test_merge_lora_and_prefix_strip():
    g = torch.Generator().manual_seed(0)
    W, A, B = torch.randn(8, 6, generator=g), torch.randn(2, 6, generator=g), torch.randn(8, 2, generator=g)
    sd = {"model.blocks.0.self_attn.q.base.weight": W, "model.blocks.0.self_attn.q.base.bias": torch.zeros(8),
          "model.blocks.0.self_attn.q.lora_A.weight": A, "model.blocks.0.self_attn.q.lora_B.weight": B,
          "model.head.head.weight": torch.ones(4, 6)}
    out = sfa.merge_lora(sfa.strip_prefix(sd), alpha=4.0, rank=2)
    assert set(out) == {"blocks.0.self_attn.q.weight", "blocks.0.self_attn.q.bias", "head.head.weight"}
    x = torch.randn(3, 6, generator=g)
    ref = x @ W.t() + (x @ A.t() @ B.t()) * (4.0 / 2)      # utils/lora.py:47-50
    assert torch.allclose(x @ out["blocks.0.self_attn.q.weight"].t(), ref, atol=1e-5)


# -------------------------------------------------------------------------------- the C-ABI
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    """The built C-ABI library loads and exports exactly what include/sf_hip.h declares (no
    compute call: there is no GPU here)."""
    lib_path = sfa._lib.LIB_PATH
    if not os.path.exists(lib_path):
        sfa._lib.build()
    handle = ctypes.CDLL(lib_path)
    declared = _header_symbols()
    assert declared, "no symbols parsed from the header"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in sf_hip.h but not exported"
    assert set(declared) == set(sfa._lib.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    handle.sf_abi_version.restype = ctypes.c_int
    assert handle.sf_abi_version() == sfa._lib.ABI_VERSION


def test_entry_points_reject_bad_arguments_without_touching_the_gpu():
    lib = sfa._lib.lib()
    assert lib.sf_gemm_bf16(None, None) != 0
    assert b"null" in lib.sf_last_error()
    g = sfa._lib.GemmArgs()
    g.M, g.N, g.K = 4, 8, 100
    assert lib.sf_gemm_bf16(g, None) != 0 and b"K=100" in lib.sf_last_error()
    assert lib.sf_attention(None, None, None, None, 1, 1, 1, 1, 128, 128, 128, 128, 128, 128, None) != 0
    # the two-pass call: null arguments, and two passes that do not name the same workspace
    assert lib.sf_dit_forward_pair(None, None, None, None) != 0 and b"null" in lib.sf_last_error()
    m, a0, a1 = sfa._lib.Model(), sfa._lib.ForwardArgs(), sfa._lib.ForwardArgs()
    a0.workspace, a1.workspace = 16, 32
    assert lib.sf_dit_forward_pair(m, a0, a1, None) != 0 and b"same workspace" in lib.sf_last_error()
    assert lib.sf_probe_copy(None, None, 64, None) != 0 and lib.sf_probe_mfma(0, 0, 0, None, None, None, None) != 0


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the op wrappers reject host tensors instead of computing something."""
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="CUDA"):
        sfa.ops.gemm(x, x)
    with pytest.raises(ValueError, match="CUDA"):
        sfa.ops.attention(torch.zeros(1, 4, 1, 128, dtype=torch.bfloat16), torch.zeros(1, 4, 1, 128, dtype=torch.bfloat16),
                          torch.zeros(1, 4, 1, 128, dtype=torch.bfloat16))


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(sfa._lib, "_lib", None)
    monkeypatch.setattr(sfa._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(sfa._lib.SfHipError, match="no CPU/eager fallback"):
        sfa._lib.lib()


# ------------------------------------------------------------ N > 1 protocol over gloo (CPU)
# These run the code bench.py runs (self_forcing_amd/distributed.py: RankGroup, self_launch, selftest), not a copy.
def _json_line(stdout):
    import json
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line on stdout, got {len(lines)}: {stdout[-800:]}"
    return json.loads(lines[0])


def _check_selftest_report(out, steps):
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["backend"] == "gloo"
    assert out["prompt_indices_per_rank"] == [[0, 2, 4, 6][:steps + 1], [1, 3, 5, 7][:steps + 1]]   # rank::W, no overlap
    assert out["units_per_rank"] == [steps, steps]
    # barrier + MAX over ranks: rank 1 "works" 0.1 s per step, rank 0 half of that
    assert out["elapsed_s"] >= 0.1 * steps - 1e-3 and out["seconds_per_rank"][1] >= 0.1 * steps - 1e-3
    assert out["seconds_per_rank"][0] < out["seconds_per_rank"][1]
    assert abs(out["ms_per_step"] - 1e3 * out["elapsed_s"] / steps) < 1e-6


def test_bench_gpus2_self_launch_runs_the_protocol(tmp_path):
    """`python bench.py --gpus 2 ...` started the way the N = 1 bench is started (no torchrun environment) launches its
    own two rank processes and prints ONE line; --dist-selftest = the same protocol over gloo with no GPU work."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-selftest", "--steps", "3",
                          "--launch-timeout", "200"], env=env, capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    _check_selftest_report(_json_line(res.stdout), 3)


def test_bench_under_an_existing_torchrun(tmp_path):
    """The driver's way: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`; and a rank count
    that does not match --gpus is refused."""
    port = str(sfa.distributed.free_port())
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
            "--master-port", port, os.path.join(ROOT, "bench.py")]
    res = subprocess.run(base + ["--gpus", "2", "--dist-selftest", "--steps", "2"], capture_output=True, text=True, timeout=240,
                         cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    _check_selftest_report(_json_line(res.stdout), 2)
    res = subprocess.run(base + ["--gpus", "4", "--dist-selftest"], capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


_WORKER = r"""
import os, sys, json, time
import torch
sys.path.insert(0, os.environ["SF_ROOT"])
from self_forcing_amd.distributed import RankGroup
grp = RankGroup(backend="gloo", timeout_s=120)
assert grp.world == 2 and grp.dist is not None
assert grp.check_replicas(1234.5) == 1234.5
try:
    grp.check_replicas(1.0 + grp.rank)           # replicas that differ must be caught on EVERY rank
    caught = False
except RuntimeError as e:
    caught = "replicas differ" in str(e)
elapsed, local, res = grp.timed(lambda: (time.sleep(0.05 * (grp.rank + 1)), grp.rank * 10)[1])
rows = grp.gather([float(grp.rank), local, 7.0])
grp.finish()
assert grp.dist is None
if grp.rank == 0:
    print(json.dumps({"caught": caught, "elapsed": elapsed, "local": local, "res": res, "rows": rows}))
else:
    assert caught and res == 10
"""


def test_rank_group_two_ranks_gloo(tmp_path):
    """RankGroup under world size 2: replica check (equal passes, unequal raises everywhere), barrier-bracketed timing
    with the MAX over ranks, gather, finish."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(sfa.distributed.free_port())
    env = dict(os.environ, SF_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    out = _json_line(res.stdout)
    assert out["caught"] and out["res"] == 0
    assert out["elapsed"] >= 0.1 - 1e-3 > out["local"]            # rank 0 slept 0.05 s, the job took rank 1's 0.1 s
    assert [r[0] for r in out["rows"]] == [0.0, 1.0] and out["rows"][1][1] >= 0.1 - 1e-3 and out["rows"][0][2] == 7.0


def test_rank_group_single_process_is_the_identity():
    from self_forcing_amd.distributed import RankGroup, launched_by_torchrun
    assert not launched_by_torchrun() or "RANK" in os.environ
    grp = RankGroup(backend="gloo")
    if grp.world == 1 and not launched_by_torchrun():
        assert grp.dist is None and grp.check_replicas(3.0) == 3.0
        el, loc, res = grp.timed(lambda: 5)
        assert res == 5 and el >= loc >= 0 and grp.gather([1, 2]) == [[1.0, 2.0]]
        grp.finish()


# ----------------------------------------------------------------------------------- config
def test_config_merge_and_pipeline_selection(tmp_path):
    from self_forcing_amd.config import load_config, is_few_step
    d = tmp_path / "default.yaml"
    d.write_text("independent_first_frame: false\nwarp_denoising_step: false\ncontext_noise: 0\nmodel_kwargs:\n  timestep_shift: 8.0\n  sink_size: 0\n")
    r = tmp_path / "run.yaml"
    r.write_text("denoising_step_list: [1000, 750, 500, 250]\nwarp_denoising_step: true\nnum_frame_per_block: 3\nmodel_kwargs:\n  timestep_shift: 5.0\n")
    cfg = load_config(str(r), str(d))
    assert cfg.warp_denoising_step is True and cfg.independent_first_frame is False and cfg.num_frame_per_block == 3
    assert cfg.model_kwargs == {"timestep_shift": 5.0, "sink_size": 0}          # nested merge, run config wins
    assert is_few_step(cfg) and hasattr(cfg, "denoising_step_list") and getattr(cfg, "nope", 7) == 7
    assert not is_few_step(load_config(str(d)))
    hot = load_config(os.path.join(ROOT, "configs", "self_forcing_dmd_hotpath.yaml"))
    assert hot.denoising_step_list == [1000, 750, 500, 250] and hot.model_kwargs["timestep_shift"] == 5.0


class _StubModel:
    num_layers, local_attn_size, sink_size, num_frame_per_block = 2, -1, 0, 1
    shape = sfa.WAN_REDUCED


class _StubGenerator:
    """Records the calls the pipeline makes (the reference's own test style: injected stand-ins,
    test_lazy_load.py:84-92)."""

    def __init__(self):
        self.model = _StubModel()
        self.scheduler = sfa.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.scheduler.add_noise = lambda x0, eps, t: x0 + 0 * eps          # keep it on the CPU
        self.calls = []

    def get_scheduler(self):
        return self.scheduler

    def forward(self, noisy_image_or_video, conditional_dict, timestep, kv_cache, crossattn_cache, current_start, cache_only=False):
        self.calls.append((tuple(noisy_image_or_video.shape), timestep.flatten().tolist(), current_start, bool(cache_only)))
        return noisy_image_or_video, noisy_image_or_video * 0.5

    __call__ = forward


@pytest.mark.parametrize("iff,nfpb,frames,initial", [(False, 3, 6, 0), (True, 3, 4, 0), (False, 1, 2, 0), (True, 3, 3, 1), (False, 3, 3, 3)])
def test_pipeline_call_sequence_on_cpu(iff, nfpb, frames, initial):
    """The rollout loop's bookkeeping (chunking, warped timesteps, current_start, context pass,
    warm-up passes) without a GPU: causal_inference.py:72-244."""
    from types import SimpleNamespace
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, independent_first_frame=iff,
                           num_frame_per_block=nfpb, context_noise=0)
    gen = _StubGenerator()
    pipe = sfa.CausalInferencePipeline(args, "cpu", generator=gen, text_encoder=lambda text_prompts: {"prompt_embeds": None},
                                       vae=sfa.IdentityVAE())
    assert [round(float(v), 2) for v in pipe.denoising_step_list] == [1000.0, 937.5, 833.33, 625.0]
    noise = torch.randn(1, frames, 16, 8, 12)
    init = torch.randn(1, initial, 16, 8, 12) if initial else None
    video, lat = pipe.inference(noise, ["p"], initial_latent=init, return_latents=True)
    fs = 24
    assert pipe.frame_seq_length == fs and lat.shape[1] == frames + initial
    chunks = ([1] if (iff and not initial) else []) + [nfpb] * ((frames - (1 if iff and not initial else 0)) // nfpb)
    warm = []
    if initial:
        if iff:
            warm.append(1)
        warm += [nfpb] * ((initial - (1 if iff else 0)) // nfpb)
    assert len(gen.calls) == len(warm) + 5 * len(chunks)
    start, k = 0, 0
    for f in warm:                                  # warm-up: t = 0, ONE timestep group, cache only
        shp, ts, cs, co = gen.calls[k]
        assert shp[1] == f and ts == [0] and cs == start * fs and co
        start += f
        k += 1
    for f in chunks:
        for i, want in enumerate([1000.0, 937.5, 833.3333129882812, 625.0, 0.0]):
            shp, ts, cs, co = gen.calls[k]
            assert shp[1] == f and cs == start * fs and len(ts) == f and all(abs(t - want) < 1e-3 for t in ts)
            assert co == (i == 4)                   # only the context pass may skip its outputs
            k += 1
        start += f
    if initial:
        assert torch.equal(lat[:, :initial], init)


def test_pipeline_calls_foreign_generator_like_the_reference():
    """A generator without the `cache_only` extension is called with the reference's keywords only."""
    from types import SimpleNamespace

    class Plain(_StubGenerator):
        def forward(self, noisy_image_or_video, conditional_dict, timestep, kv_cache, crossattn_cache, current_start):
            self.calls.append(current_start)
            return noisy_image_or_video, noisy_image_or_video
        __call__ = forward

    args = SimpleNamespace(denoising_step_list=[1000, 500], warp_denoising_step=False, independent_first_frame=False,
                           num_frame_per_block=1, context_noise=0)
    gen = Plain()
    pipe = sfa.CausalInferencePipeline(args, "cpu", generator=gen, text_encoder=lambda text_prompts: {}, vae=sfa.IdentityVAE())
    assert pipe.denoising_step_list.dtype == torch.long          # un-warped list stays integer
    pipe.inference(torch.randn(1, 2, 16, 8, 12), ["p"])
    assert gen.calls == [0, 0, 0, 24, 24, 24]


# ------------------------------------------------------------------------------------------ VAE host logic
def test_vae_repack_conv_matches_convolution_definition():
    """`repack_conv` lays weights out as [Cout][tap*Cin_pad + ci] with tap = (dt*kh + dh)*kw + dw: an explicit
    gather-and-matmul with that layout must reproduce F.conv3d."""
    import torch.nn.functional as F
    from self_forcing_amd.vae import repack_conv
    g = torch.Generator().manual_seed(0)
    cin, cout, T, H, W = 5, 4, 2, 3, 4
    x = torch.randn(cin, T + 2, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g)
    rp = repack_conv(w)
    assert rp.shape == (cout, 27 * 32 + 32) and rp.shape[1] % 64 == 0         # Cin 5 -> 32, K 864 -> 896
    xp = F.pad(x, (1, 1, 1, 1, 0, 0))
    cols = torch.zeros(T, H, W, rp.shape[1])
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                tap = (dt * 3 + dh) * 3 + dw
                cols[..., tap * 32:tap * 32 + cin] = xp[:, dt:dt + T, dh:dh + H, dw:dw + W].permute(1, 2, 3, 0)
    ref = F.conv3d(xp[None], w)[0].permute(1, 2, 3, 0)
    assert torch.allclose(cols @ rp.t(), ref, atol=1e-4)
    # Conv2d weights are treated as kt = 1
    assert repack_conv(torch.randn(8, 64, 3, 3)).shape == (8, 9 * 64)


def test_vae_decode_flops_and_frame_counts():
    from self_forcing_amd import vae_weights as vw
    one = vw.vae_decode_flops(vw.WAN_VAE, 60, 104, 1)
    two = vw.vae_decode_flops(vw.WAN_VAE, 60, 104, 2)
    assert 13.0e12 < two - one < 14.0e12          # ~13.6 TFLOP per steady latent frame at 832x480
    assert one < (two - one) / 3                  # the first latent frame yields 1 pixel frame, not 4
    assert vw.WAN_VAE.temporal_factor == 4 and vw.WAN_VAE.spatial_factor == 8
    mid, ups = vw.decoder_layout(vw.WAN_VAE)
    assert [u.mode for u in ups if isinstance(u, vw.ResampleSpec)] == ["upsample3d", "upsample3d", "upsample2d"]
    assert len([u for u in ups if isinstance(u, vw.ResBlockSpec)]) == 12


def test_vae_wrapper_has_no_cpu_fallback():
    from self_forcing_amd import vae_weights as vw
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises((RuntimeError, AssertionError)):
        sfa.WanVAEWrapper(vw.synth_vae_state_dict(vw.VAE_REDUCED, seed=0), device="cuda:0", shape=vw.VAE_REDUCED)


# -------------------------------------------------------------------------------- torch custom ops (row b')
def test_torch_custom_ops_are_registered_with_mutation_schemas():
    """torch.ops.sf_hip.* exist, take Tensors, and declare which arguments they write ("registered ... so
    torch.ops.<ns>.* take Tensors"; the caches / workspace / outputs are the mutated ones)."""
    for name in sfa.torch_ops.OPS:
        assert hasattr(torch.ops.sf_hip, name), name
    sch = {n: str(getattr(torch.ops.sf_hip, n).default._schema) for n in sfa.torch_ops.OPS}
    assert "Tensor(a5!)[] k_cache" in sch["dit_forward"] and "Tensor(a6!)[] v_cache" in sch["dit_forward"]
    assert "Tensor(a7!)[] ck_cache" in sch["dit_forward"] and "-> (Tensor, Tensor)" in sch["dit_forward"]
    assert "Tensor(a1!) state" in sch["vae_decode_frames"] and "Tensor(a4!) out" in sch["vae_decode_frames"]
    assert "Tensor(a0!) out" in sch["gemm_out"] and "Tensor(a0!) out" in sch["lincomb_out"]
    assert "!" not in sch["attention"] and "!" not in sch["gemm"] and "!" not in sch["add_noise"]


def test_torch_custom_ops_have_fake_implementations():
    """Shape / dtype propagation without touching a GPU (what torch.compile's tracer runs, demo.py:340)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    class _Stub:                      # stands in for a device-resident model: the fake kernels only read its shape
        shape = SimpleNamespace(out_dim=16, dim=4096)
    stub = _Stub()
    h = sfa.torch_ops.register_model(stub)
    with FakeTensorMode():
        e = lambda *s, dt=torch.bfloat16: torch.empty(*s, device="cuda", dtype=dt)  # noqa: E731
        o = torch.ops.sf_hip.attention(e(2, 70, 3, 128), e(2, 300, 3, 128), e(2, 300, 3, 128))
        assert tuple(o.shape) == (2, 70, 3, 128) and o.dtype == torch.bfloat16
        y = torch.ops.sf_hip.gemm(e(100, 64), e(256, 64), e(256), 0, None, None, None, 1, 0)
        assert tuple(y.shape) == (100, 256) and y.dtype == torch.bfloat16
        y32 = torch.ops.sf_hip.gemm(e(100, 64), e(256, 64), None, 4, None, None, None, 1, 0)
        assert y32.dtype == torch.float32
        z = torch.ops.sf_hip.lincomb([e(3, 5), e(3, 5)], [0.5, 2.0])
        assert tuple(z.shape) == (3, 5)
        n = torch.ops.sf_hip.add_noise(e(3, 16, 8, 8), e(3, 16, 8, 8), e(3, dt=torch.float32), e(1000, dt=torch.float32), e(1000, dt=torch.float32))
        assert tuple(n.shape) == (3, 16, 8, 8)
        caches = [[e(1, 48, 4, 128) for _ in range(2)] for _ in range(2)] + [[e(1, 512, 4, 128) for _ in range(2)] for _ in range(2)]
        flow, x0 = torch.ops.sf_hip.dit_forward(h, e(1, 2, 16, 8, 12), e(1, 2, dt=torch.float32), None, None, *caches,
                                                e(1024, dt=torch.uint8), None, False, False, 0, 0, 0, 0, 0, 48, 0, None, 0)
        assert tuple(flow.shape) == (1, 2, 16, 8, 12) and tuple(x0.shape) == (1, 2, 16, 8, 12)
        flow, x0 = torch.ops.sf_hip.dit_forward(h, e(1, 2, 16, 8, 12), e(1, 2, dt=torch.float32), None, None, *caches,
                                                e(1024, dt=torch.uint8), None, False, True, 0, 0, 0, 0, 0, 48, 0, e(2, 2, dt=torch.int64), 24)
        assert flow.numel() == 0
        t = torch.ops.sf_hip.t5_encode(h, e(2, 512, dt=torch.int64), e(2, 512, dt=torch.int64), e(1023, dt=torch.int32), e(64, dt=torch.uint8))
        assert tuple(t.shape) == (2, 512, 4096) and t.dtype == torch.bfloat16


def test_torch_custom_ops_refuse_cpu_tensors():
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="CUDA"):
        torch.ops.sf_hip.gemm(x, x, None, 0, None, None, None, 1, 0)
    with pytest.raises(ValueError, match="CUDA"):
        torch.ops.sf_hip.lincomb([x], [1.0])
    with pytest.raises(RuntimeError, match="not registered"):
        torch.ops.sf_hip.t5_encode(12345, torch.zeros(1, 4, dtype=torch.int64), torch.zeros(1, 4, dtype=torch.int64),
                                   torch.zeros(7, dtype=torch.int32), torch.zeros(8, dtype=torch.uint8))


def test_default_components_need_the_reference_checkpoints(tmp_path, monkeypatch):
    """`WanTextEncoder()` / `WanVAEWrapper()` -- what the pipelines build when nothing is injected
    (causal_inference.py:19-23) -- load the reference's default local checkpoints and say so when they are absent."""
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="umT5 checkpoint"):
        sfa.WanTextEncoder(device="cpu")
    with pytest.raises(FileNotFoundError, match="VAE checkpoint"):
        sfa.WanVAEWrapper(device="cpu")
    assert sfa.text_encoder.HuggingfaceTokenizer.clean("  a &amp;amp; b \n\t c ") == "a & b c"


def test_bench_reads_counter_traffic_only_at_the_profiled_shape():
    """bench.py's roofline.traffic comes from the committed rocprofv3 --pmc pass ONLY when that pass was taken at the
    shape and batch being reported; anything else is null (round 2 reported the 1.3B figure beside a 14B run)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lks = [4680 * i for i in range(1, 8)]
    t1, src1 = bench.committed_traffic(sfa.WAN_1_3B, 4680, lks, 1, 0)
    t2, src2 = bench.committed_traffic(sfa.WAN_1_3B, 4680, lks, 2, 0)
    assert t1 and t2 and 1.9 < t2 / t1 < 2.1 and "r03_pmc_traffic.json" in src1 and "batch2" in src2
    assert bench.committed_traffic(sfa.WAN_14B, 10800, [10800 * i for i in range(1, 8)], 1, 0)[0] is None      # another model
    assert bench.committed_traffic(sfa.WAN_1_3B, 4680, lks[:5], 1, 0)[0] is None                                  # other cache lengths
    assert bench.committed_traffic(sfa.WAN_1_3B, 4680, lks, 3, 0)[0] is None                                      # unprofiled batch
    assert bench.committed_traffic(sfa.WAN_1_3B, 4680, lks, 1, 21)[0] is None                                     # rolling window
    # algorithmic FLOPs of the S1 rollout (990 TFLOP) and what the context passes skip
    full = bench.rollout_flops(sfa.WAN_1_3B, 21, 3, 4, 1560)
    assert abs(full / 1e12 - 990.3) < 1.0 and bench.rollout_flops(sfa.WAN_1_3B, 21, 3, 4, 1560, executed=True) < full
