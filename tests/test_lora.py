"""LoRA: the reference evaluates base(x) + B(A(x)) * alpha / rank on every call (utils/lora.py:47-50);
this build folds the adapters into the base matrices at load time.  Both against the reference's own run of
`apply_lora` + `load_lora_weights` on its CausalWanModel with NON-ZERO adapters on q, k, v, o of both attentions and
ffn.0 / ffn.2 (oracle/make_golden_lora.py -> tests/golden/lora_reduced.npz)."""
import os

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from oracle import wan_oracle as wo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LAT_H, LAT_W = 8, 12
FS = (LAT_H // 2) * (LAT_W // 2)
TARGETS = ["q", "k", "v", "o", "ffn.0", "ffn.2"]


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLD, "lora_reduced.npz"))


def _adapters(G, **kw):
    return sfa.synth_lora_state_dict(sfa.WAN_REDUCED, int(G["rank"]), seed=int(G["lora_seed"]), targets=TARGETS,
                                     b_std=float(G["b_std"]), **kw)


def test_fixture_is_not_vacuous(G):
    assert float(G["lora_effect_f32"]) > 0.05          # the adapters move the output by ~9 %, 15x the bf16 noise floor
    assert rel(T(G["y2_bf16"]), T(G["y2_f32"])) < 1e-2


def test_merged_weights_reproduce_the_references_unmerged_lora_fp32(G):
    """CPU: the oracle on MERGED weights (apply_lora_file, fp32 so the merge itself adds no rounding) against the
    reference's unmerged LoRALinear run in fp32: the merge is the same function (<= 1e-5)."""
    shape = sfa.WAN_REDUCED
    sd = {k: v.float() for k, v in sfa.synth_state_dict(shape, seed=int(G["weights_seed"])).items()}
    merged, loaded, skipped = sfa.apply_lora_file(sd, {"diffusion_model." + k: v for k, v in _adapters(G).items()}, shape,
                                                  int(G["rank"]), float(G["alpha"]), TARGETS)
    assert loaded == 10 * shape.num_layers and skipped == 0
    W = {k: v.float() for k, v in merged.items()}      # (prepare_weights would round the merged matrices to bf16)
    cfg = wo.OracleConfig(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers,
                          text_dim=shape.text_dim)
    kv, ca = wo.init_kv_cache(cfg, 1, 5 * FS, torch.float32), wo.init_crossattn_cache(cfg, 1, torch.float32)
    y1 = wo.forward_inference(W, cfg, T(G["x1"]), T(G["t1"]), T(G["pe"]), kv, ca, 0)
    y2 = wo.forward_inference(W, cfg, T(G["x2"]), T(G["t2"]), T(G["pe"]), kv, ca, 2 * FS)
    assert rel(y1, T(G["y1_f32"])) < 1e-5 and rel(y2, T(G["y2_f32"])) < 1e-5
    assert rel(kv[0]["k"], T(G["k0_f32"])) < 1e-5 and rel(kv[1]["v"], T(G["v1_f32"])) < 1e-5
    # default targets (q, k, v, o): the file's ffn adapters are skipped, as load_lora_weights skips them
    _, loaded, skipped = sfa.apply_lora_file(sd, _adapters(G), shape, int(G["rank"]), float(G["alpha"]))
    assert loaded == 8 * shape.num_layers and skipped == 2 * shape.num_layers
    # kohya-style names are accepted too (utils/lora.py:64-76)
    alt = {k.replace("lora_B", "lora_up").replace("lora_A", "lora_down"): v for k, v in _adapters(G, prefix="pipe.dit.").items()}
    m2, loaded, _ = sfa.apply_lora_file(sd, alt, shape, int(G["rank"]), float(G["alpha"]), TARGETS)
    assert loaded == 10 * shape.num_layers and torch.equal(m2["blocks.1.ffn.2.weight"], merged["blocks.1.ffn.2.weight"])


def test_files_without_usable_adapters_are_not_silent(G):
    """A file with no (up, down) pair raises like the reference's load_lora_weights (utils/lora.py:150-152); a file whose
    pairs all miss the model (wrong prefix / foreign modules) warns instead of silently running the base model."""
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0)
    with pytest.raises(ValueError, match="No LoRA pairs found"):
        sfa.apply_lora_file(sd, {"blocks.0.self_attn.q.weight": torch.zeros(2, 2)}, shape, 8, 4.0, TARGETS)
    with pytest.raises(ValueError, match="No LoRA pairs found"):       # an up matrix without its down matrix is no pair
        sfa.apply_lora_file(sd, {"blocks.0.self_attn.q.lora_B.weight": torch.zeros(2, 2)}, shape, 8, 4.0, TARGETS)
    foreign = {"unet." + k: v for k, v in _adapters(G).items()}
    with pytest.warns(UserWarning, match="NONE maps to a wrapped Linear"):
        out, loaded, skipped = sfa.apply_lora_file(sd, foreign, shape, 8, 4.0, TARGETS)
    assert loaded == 0 and skipped == 10 * shape.num_layers and all(torch.equal(out[k], sd[k]) for k in sd)


def test_small_adapters_survive_the_merge_rounding(G):
    """Adapters that move the weights by a few bf16 ulps (b_std 0.05: the output by 0.8 %; fixture keys `*_small`, the
    reference's own unmerged run).  fp32 merge: the same function (<= 1e-5).  bf16 merge, as the device path stores it
    (W + scale * B A rounded ONCE to bf16), evaluated in fp32 arithmetic: its distance to the reference's unmerged fp32
    run is the rounding of the weights only -- no larger than what rounding the BASE weights costs the base model --
    and far below the adapters' effect, i.e. the merged matrices still carry them."""
    shape = sfa.WAN_REDUCED
    eff = float(G["lora_effect_f32_small"])
    assert 4e-3 < eff < 2e-2
    small = sfa.synth_lora_state_dict(shape, int(G["rank"]), seed=int(G["lora_seed"]), targets=TARGETS, b_std=float(G["b_std_small"]))
    sd16 = sfa.synth_state_dict(shape, seed=int(G["weights_seed"]))
    cfg = wo.OracleConfig(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers,
                          text_dim=shape.text_dim)

    def run(W):
        kv, ca = wo.init_kv_cache(cfg, 1, 5 * FS, torch.float32), wo.init_crossattn_cache(cfg, 1, torch.float32)
        return wo.forward_inference({k: v.float() for k, v in W.items()}, cfg, T(G["x1"]), T(G["t1"]), T(G["pe"]), kv, ca, 0)

    merged32, loaded, _ = sfa.apply_lora_file({k: v.float() for k, v in sd16.items()}, small, shape, int(G["rank"]), float(G["alpha"]), TARGETS)
    assert loaded == 10 * shape.num_layers and rel(run(merged32), T(G["y1_f32_small"])) < 1e-5
    merged16, _, _ = sfa.apply_lora_file(sd16, small, shape, int(G["rank"]), float(G["alpha"]), TARGETS)
    assert merged16["blocks.0.self_attn.q.weight"].dtype == torch.bfloat16
    e_merge = rel(run(merged16), T(G["y1_f32_small"]))
    e_base = rel(run(sd16), T(G["y1_f32_small"]))            # the base model: what ignoring the adapters would give
    assert e_merge < 0.25 * eff and e_merge < 0.25 * e_base, (e_merge, e_base, eff)


def test_merge_lora_in_state_dict_layout_equals_file_merge(G):
    """The layout `apply_lora` leaves in a checkpoint (`<linear>.base.weight` + lora_A / lora_B) merges to the same
    matrices as the file route."""
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0)
    lora = _adapters(G)
    wrapped = set(sfa.lora_target_linears(shape, TARGETS))
    ck = {}
    for k, v in sd.items():
        stem, leaf = k.rsplit(".", 1)
        ck[("model." + stem + ".base." + leaf) if stem in wrapped else ("model." + k)] = v
    ck.update({"model." + k: v for k, v in lora.items()})
    a = sfa.merge_lora(sfa.strip_prefix(ck), alpha=float(G["alpha"]), rank=int(G["rank"]))
    b, _, _ = sfa.apply_lora_file(sd, lora, shape, int(G["rank"]), float(G["alpha"]), TARGETS)
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["lora_path", "state_dict"])
def test_hip_forward_with_lora_vs_reference(G, route, tmp_path):
    """-m gpu: WanDiffusionWrapper(lora_rank=, lora_alpha=, lora_targets=, lora_path=) -- and, second route, adapters
    inside the state dict -- through the HIP path, against the reference's fp32 LoRALinear run: <= 2e-2 relative
    Frobenius (the reference's own bf16 unmerged run is at 5.9e-3)."""
    from safetensors.torch import save_file
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0)
    lora = _adapters(G)
    kw = dict(shape=shape, timestep_shift=5.0, is_causal=True, device="cuda:0", lora_rank=int(G["rank"]), lora_alpha=float(G["alpha"]),
              lora_targets=TARGETS)
    if route == "lora_path":
        path = str(tmp_path / "adapters.safetensors")
        save_file({"diffusion_model." + k: v.contiguous() for k, v in lora.items()}, path)
        gen = sfa.WanDiffusionWrapper(state_dict=sd, lora_path=path, **kw)
        assert gen.lora_loaded == 10 * shape.num_layers and gen.lora_skipped == 0
    else:
        wrapped = set(sfa.lora_target_linears(shape, TARGETS))
        ck = {}
        for k, v in sd.items():
            stem, leaf = k.rsplit(".", 1)
            ck[("model." + stem + ".base." + leaf) if stem in wrapped else ("model." + k)] = v
        ck.update({"model." + k: v for k, v in lora.items()})
        gen = sfa.WanDiffusionWrapper(state_dict=ck, **kw)
    from types import SimpleNamespace
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, independent_first_frame=False,
                           num_frame_per_block=1, context_noise=0)
    pe = T(G["pe"], torch.bfloat16).cuda()
    pipe = sfa.CausalInferencePipeline(args, "cuda:0", generator=gen, text_encoder=sfa.FixedTextEncoder(pe), vae=sfa.IdentityVAE())
    pipe.frame_seq_length = FS
    pipe._initialize_kv_cache(1, torch.bfloat16, "cuda:0", cache_tokens=5 * FS)
    pipe._initialize_crossattn_cache(1, torch.bfloat16, "cuda:0")
    x1 = T(G["x1"], torch.bfloat16).permute(0, 2, 1, 3, 4).contiguous().cuda()     # wrapper layout [B, F, C, H, W]
    x2 = T(G["x2"], torch.bfloat16).permute(0, 2, 1, 3, 4).contiguous().cuda()
    f1, _ = gen(x1, {"prompt_embeds": pe}, T(G["t1"]).cuda(), pipe.kv_cache1, pipe.crossattn_cache, 0)
    f2, _ = gen(x2, {"prompt_embeds": pe}, T(G["t2"]).cuda(), pipe.kv_cache1, pipe.crossattn_cache, 2 * FS)
    torch.cuda.synchronize()
    y1, y2 = f1.permute(0, 2, 1, 3, 4), f2.permute(0, 2, 1, 3, 4)                  # model layout [B, C, F, H, W]
    assert rel(y1, T(G["y1_f32"])) < 2e-2 and rel(y2, T(G["y2_f32"])) < 2e-2, (rel(y1, T(G["y1_f32"])), rel(y2, T(G["y2_f32"])))
    assert rel(pipe.kv_cache1[0]["k"], T(G["k0_f32"])) < 2e-2 and rel(pipe.kv_cache1[1]["v"], T(G["v1_f32"])) < 2e-2
    assert rel(pipe.crossattn_cache[1]["k"][:, :80], T(G["ck1_f32"])) < 2e-2
    # and it IS the LoRA model: without adapters the same call is ~9 % away
    base = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd, timestep_shift=5.0, is_causal=True, device="cuda:0")
    pipe._initialize_kv_cache(1, torch.bfloat16, "cuda:0", cache_tokens=5 * FS)
    pipe._initialize_crossattn_cache(1, torch.bfloat16, "cuda:0")
    b1, _ = base(x1, {"prompt_embeds": pe}, T(G["t1"]).cuda(), pipe.kv_cache1, pipe.crossattn_cache, 0)
    assert rel(b1.permute(0, 2, 1, 3, 4), T(G["y1_f32"])) > 0.05
    # small adapters (weights moved by a few bf16 ulps, output by 0.8 %): still honoured through the bf16 merge --
    # within the tolerance of the reference's LoRA run AND closer to it than the adapter-free model is
    small = sfa.synth_lora_state_dict(shape, int(G["rank"]), seed=int(G["lora_seed"]), targets=TARGETS, b_std=float(G["b_std_small"]))
    if route == "lora_path":
        path = str(tmp_path / "small.safetensors")
        save_file({"diffusion_model." + k: v.contiguous() for k, v in small.items()}, path)
        gs = sfa.WanDiffusionWrapper(state_dict=sd, lora_path=path, **kw)
        pipe._initialize_kv_cache(1, torch.bfloat16, "cuda:0", cache_tokens=5 * FS)
        pipe._initialize_crossattn_cache(1, torch.bfloat16, "cuda:0")
        s1, _ = gs(x1, {"prompt_embeds": pe}, T(G["t1"]).cuda(), pipe.kv_cache1, pipe.crossattn_cache, 0)
        e_lora, e_base = rel(s1.permute(0, 2, 1, 3, 4), T(G["y1_f32_small"])), rel(b1.permute(0, 2, 1, 3, 4), T(G["y1_f32_small"]))
        assert e_lora < 2e-2 and e_lora < e_base, (e_lora, e_base, float(G["lora_effect_f32_small"]))
