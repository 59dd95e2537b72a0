"""GPU parity tests of the umT5 text encoder through the C-ABI: against the golden vectors
recorded from the reference's T5Encoder and against the CPU oracle.  Run with `-m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from self_forcing_amd import _lib, t5_weights as tw

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import t5_oracle as to  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
TOL = 2e-2


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_encoder_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "t5_reduced.npz"))
    enc = sfa.WanTextEncoder(tw.synth_t5_state_dict(tw.T5_REDUCED, seed=int(g["seed"])), device=DEV, shape=tw.T5_REDUCED)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    out = enc.encode_ids(ids, mask)["prompt_embeds"]
    gold = torch.from_numpy(g["context_f32"])
    assert out.shape == gold.shape and out.dtype == torch.bfloat16
    err = rel(out.float(), gold)
    assert err < TOL, f"rel err {err:.4f} (reference bf16 itself: {float(g['ref_bf16_rel_err']):.4f})"
    lens = g["mask"].sum(1)
    for i, n in enumerate(lens):          # rows past the prompt are exactly zero, each prompt on its own within tolerance
        assert float(out[i, n:].float().abs().max()) == 0.0 if n < out.shape[1] else True
        assert rel(out[i, :n].float(), gold[i, :n]) < TOL
    assert torch.equal(out, enc.encode_ids(ids, mask)["prompt_embeds"])


def test_encoder_matches_oracle_other_length_and_tokenizer_hook():
    shape = tw.T5_REDUCED
    sd = tw.synth_t5_state_dict(shape, seed=2)
    g = torch.Generator().manual_seed(3)
    L = 64
    ids = torch.randint(1, shape.vocab_size, (2, L), generator=g)
    mask = torch.ones(2, L, dtype=torch.long)
    mask[1, 20:] = 0
    ids[1, 20:] = 0
    cfg = to.T5OracleConfig(dim=shape.dim, dim_attn=shape.dim_attn, dim_ffn=shape.dim_ffn, num_heads=shape.num_heads,
                            num_layers=shape.num_layers)
    ref = to.text_encoder_forward(cfg, to.prepare_weights(sd, torch.float32), ids, mask)
    enc = sfa.WanTextEncoder(sd, tokenizer=lambda texts: (ids[:len(texts)], mask[:len(texts)]), device=DEV, shape=shape)
    out = enc(text_prompts=["a", "b"])["prompt_embeds"]
    assert rel(out.float(), ref) < TOL
    with pytest.raises(NotImplementedError, match="tokenizer"):
        sfa.WanTextEncoder(sd, device=DEV, shape=shape)(text_prompts=["a"])
    with pytest.raises(ValueError, match="token id"):
        enc.encode_ids(torch.full((1, L), shape.vocab_size), torch.ones(1, L, dtype=torch.long))


def test_softmax_bias_kernel_against_torch():
    g = torch.Generator().manual_seed(5)
    H, L, ld = 8, 100, 128
    s = torch.randn(H, L, ld, generator=g) * 3
    emb = torch.randn(32, H, generator=g).to(torch.bfloat16)
    mask = torch.ones(L, dtype=torch.long)
    mask[70:] = 0
    rb = sfa.relative_position_buckets(L)
    idx = torch.arange(L)[None, :] - torch.arange(L)[:, None] + L - 1
    bias = emb.float()[rb.long()[idx]].permute(2, 0, 1)                    # [H, L, L]
    logits = (s[:, :, :L] + bias).masked_fill(mask[None, None, :] == 0, float("-inf"))
    ref = torch.softmax(logits, -1)
    p = torch.empty(H, L, ld, dtype=torch.bfloat16, device=DEV)
    sd_, ed_, rd_, md_ = s.to(DEV), emb.to(DEV), rb.to(DEV), mask.to(DEV)       # keep the device copies alive across the call
    _lib.check(_lib.lib().sf_t5_softmax_bias(sd_.data_ptr(), p.data_ptr(), ed_.data_ptr(), rd_.data_ptr(), md_.data_ptr(), H, L, ld,
                                             torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert float(p[:, :, L:].float().abs().max()) == 0.0 and float(p[:, :, 70:L].float().abs().max()) == 0.0
    assert rel(p[:, :, :L].float(), ref) < 4e-3


def test_pipeline_with_real_text_encoder(tmp_path):
    """The rollout consumes the encoder's embeddings through the `text_encoder=` injection point."""
    from types import SimpleNamespace
    t5s = tw.T5Shape(vocab_size=512, dim=512, dim_attn=512, dim_ffn=1024, num_heads=8, num_layers=1)
    dit = sfa.WAN_REDUCED.replace(text_dim=512)
    ids = torch.randint(1, 512, (1, 512), generator=torch.Generator().manual_seed(1))
    mask = torch.zeros(1, 512, dtype=torch.long)
    mask[:, :33] = 1
    enc = sfa.WanTextEncoder(tw.synth_t5_state_dict(t5s, seed=0), tokenizer=lambda t: (ids, mask), device=DEV, shape=t5s)
    gen = sfa.WanDiffusionWrapper(shape=dit, state_dict=sfa.synth_state_dict(dit, seed=0), timestep_shift=5.0, is_causal=True, device=DEV)
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, independent_first_frame=False,
                           num_frame_per_block=1, context_noise=0)
    pipe = sfa.CausalInferencePipeline(args, DEV, generator=gen, text_encoder=enc, vae=sfa.IdentityVAE())
    noise = torch.randn(1, 2, 16, 8, 12, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).to(DEV)
    _, lat = pipe.inference(noise, ["a red fox"], return_latents=True)
    assert lat.shape == (1, 2, 16, 8, 12) and torch.isfinite(lat.float()).all()


def test_encoder_at_xxl_layer_geometry_vs_reference_golden():
    """Every encoder kernel at umT5-XXL's real channel counts (dim 4096, 64 heads of 64, ffn 10240; 2 layers) against
    the reference's fp32 run (oracle/make_golden_t5.py::xxl_geometry), prompts of 192 and 45 tokens."""
    g = np.load(os.path.join(GOLD, "t5_xxl_geometry.npz"))
    shape = tw.T5Shape(vocab_size=512, num_layers=2)
    enc = sfa.WanTextEncoder(tw.synth_t5_state_dict(shape, seed=int(g["seed"])), device=DEV, shape=shape)
    out = enc.encode_ids(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))["prompt_embeds"]
    gold = torch.from_numpy(g["context_f32_sub"])
    err = rel(out[:, ::4, ::4].float(), gold)
    assert err < TOL, f"rel err {err:.4f} (reference bf16 itself: {float(g['ref_bf16_rel_err']):.4f})"
    for i, n in enumerate(g["mask"].sum(1)):
        if n < out.shape[1]:
            assert float(out[i, n:].float().abs().max()) == 0.0
    sums = out.double().sum(dim=(1, 2)).cpu().numpy()
    assert np.all(np.abs(sums - g["sums"]) < 2e-3 * g["abs_sums"])
