"""The umT5-encoder oracle against the golden vectors recorded from the reference (CPU only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import t5_oracle as to  # noqa: E402
from self_forcing_amd import t5_weights as tw  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def load():
    g = np.load(os.path.join(GOLD, "t5_reduced.npz"))
    s = tw.T5_REDUCED
    cfg = to.T5OracleConfig(dim=s.dim, dim_attn=s.dim_attn, dim_ffn=s.dim_ffn, num_heads=s.num_heads, num_layers=s.num_layers,
                            num_buckets=s.num_buckets)
    return g, cfg, tw.synth_t5_state_dict(s, seed=int(g["seed"]))


def test_encoder_fp32_matches_reference():
    g, cfg, sd = load()
    out = to.text_encoder_forward(cfg, to.prepare_weights(sd, torch.float32), torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    assert out.shape == g["context_f32"].shape
    assert rel(out.numpy(), g["context_f32"]) < 1e-5
    lens = g["mask"].sum(1)
    assert all(float(np.abs(out[i, n:].numpy()).max()) == 0.0 for i, n in enumerate(lens) if n < out.shape[1])


def test_encoder_bf16_mode_within_reference_noise():
    g, cfg, sd = load()
    out = to.text_encoder_forward(cfg, to.prepare_weights(sd, torch.bfloat16), torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    assert rel(out.float().numpy(), g["context_f32"]) < 1.5 * float(g["ref_bf16_rel_err"])


def test_relative_position_buckets():
    """Bidirectional T5 bucketing: 16 buckets per sign, exact up to 7, logarithmic to 128, clamped."""
    rel_pos = torch.tensor([-200, -128, -20, -8, -7, -1, 0, 1, 7, 8, 20, 127, 128, 500])
    b = to.relative_position_bucket(rel_pos, 32, 128)
    assert b.tolist() == [15, 15, 10, 8, 7, 1, 0, 17, 23, 24, 26, 31, 31, 31]


def test_t5_param_shapes():
    ps = tw.t5_param_shapes(tw.UMT5_XXL)
    assert len(ps) == 2 + 24 * 10 and ps["blocks.23.pos_embedding.embedding.weight"] == (32, 64)
    assert sum(int(np.prod(v)) for v in ps.values()) > 5.6e9


def test_encoder_fp32_matches_reference_at_xxl_layer_geometry():
    """dim 4096, 64 heads, gated-GELU ffn 10240 (2 layers, 512-row vocabulary): oracle/make_golden_t5.py::xxl_geometry."""
    g = np.load(os.path.join(GOLD, "t5_xxl_geometry.npz"))
    s = tw.T5Shape(vocab_size=512, num_layers=2)
    cfg = to.T5OracleConfig(dim=s.dim, dim_attn=s.dim_attn, dim_ffn=s.dim_ffn, num_heads=s.num_heads, num_layers=s.num_layers,
                            num_buckets=s.num_buckets)
    sd = tw.synth_t5_state_dict(s, seed=int(g["seed"]))
    out = to.text_encoder_forward(cfg, to.prepare_weights(sd, torch.float32), torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    assert rel(out[:, ::4, ::4].numpy(), g["context_f32_sub"]) < 1e-5
    assert np.allclose(out.double().sum(dim=(1, 2)).numpy(), g["sums"], rtol=0, atol=1e-4 * g["abs_sums"].max())
