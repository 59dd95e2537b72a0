"""The VAE-decode oracle against the golden vectors recorded from the reference (CPU only)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vae_oracle as vo  # noqa: E402
from self_forcing_amd import vae_weights as vw  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {"reduced": vw.VAE_REDUCED, "full": vw.WAN_VAE}


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def load(name):
    g = np.load(os.path.join(GOLD, f"vae_{name}.npz"))
    shape = CASES[name]
    assert int(g["shape_dim"]) == shape.dim
    sd = vw.synth_vae_state_dict(shape, seed=int(g["seed"]))
    cfg = vo.VaeOracleConfig(dim=shape.dim, z_dim=shape.z_dim, dim_mult=shape.dim_mult,
                             num_res_blocks=shape.num_res_blocks, temperal_upsample=shape.temperal_upsample)
    return g, cfg, sd


@pytest.mark.parametrize("name", list(CASES))
def test_decode_fp32_matches_reference(name):
    g, cfg, sd = load(name)
    W = vo.prepare_weights(sd, torch.float32)
    lat = torch.from_numpy(g["latent"])
    out, _ = vo.decode_to_pixel(cfg, W, lat, torch.tensor(vw.LATENT_MEAN), torch.tensor(vw.LATENT_STD))
    assert out.shape == g["pixels_f32"].shape and out.dtype == torch.float32
    assert out.shape[1] == 1 + 4 * (lat.shape[1] - 1) and out.shape[-1] == 8 * lat.shape[-1]
    assert rel(out.numpy(), g["pixels_f32"]) < 1e-5


@pytest.mark.parametrize("name", list(CASES))
def test_streaming_state_matches_cached_decode(name):
    """decode([0:1]) then decode([1:]) with the carried state = the reference's two cached_decode calls
    (= its one-shot decode, which the fixture script asserts)."""
    g, cfg, sd = load(name)
    W = vo.prepare_weights(sd, torch.float32)
    lat = torch.from_numpy(g["latent"])
    mean, std = torch.tensor(vw.LATENT_MEAN), torch.tensor(vw.LATENT_STD)
    st = vo.DecoderState()
    a, st = vo.decode_to_pixel(cfg, W, lat[:, :1], mean, std, st)
    b, st = vo.decode_to_pixel(cfg, W, lat[:, 1:], mean, std, st)
    assert a.shape[1] == 1 and b.shape[1] == 4 * (lat.shape[1] - 1)
    assert rel(torch.cat([a, b], 1).numpy(), g["stream_f32"]) < 1e-5


def test_bf16_mode_within_reference_noise():
    g, cfg, sd = load("reduced")
    W = vo.prepare_weights(sd, torch.bfloat16)
    out, _ = vo.decode_to_pixel(cfg, W, torch.from_numpy(g["latent"]), torch.tensor(vw.LATENT_MEAN), torch.tensor(vw.LATENT_STD))
    # same op sequence as the reference in bf16; CPU bf16 convolutions are not bit-reproducible across
    # call patterns, so the check is "as far from the fp32 truth as the reference's own bf16 run"
    assert rel(out.numpy(), g["pixels_f32"]) < 1.5 * float(g["ref_bf16_rel_err"])
    assert rel(out.numpy(), g["pixels_bf16"]) < 2.0 * float(g["ref_bf16_rel_err"])


def test_vae_param_shapes_cover_decoder():
    ps = vw.vae_param_shapes(vw.WAN_VAE)
    assert len(ps) == 108 and ps["decoder.head.2.weight"] == (3, 96, 3, 3, 3)
    assert ps["decoder.upsamples.3.time_conv.weight"] == (768, 384, 3, 1, 1)
    assert ps["decoder.upsamples.11.resample.1.weight"] == (96, 192, 3, 3)
    assert "decoder.upsamples.11.time_conv.weight" not in ps       # upsample2d has no time conv
    assert vw.WAN_VAE.dims == [384, 384, 384, 192, 96]
