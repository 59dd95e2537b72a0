"""GPU parity tests of the VAE decode path: every HIP kernel through the C-ABI against
a plain fp32 torch/CPU statement of the same op, and the whole decode against the golden vectors
recorded from the reference (`oracle/make_golden_vae.py`) and the CPU oracle.  Run with `-m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import self_forcing_amd as sfa
from self_forcing_amd import ops, vae_weights as vw
from self_forcing_amd.vae import repack_conv

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import vae_oracle as vo  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
TOL = 2e-2   # relative Frobenius error of the decoded video against the fp32 reference (the reference's
             # own bf16 run is 1.3-1.6e-2 away from it, see ref_bf16_rel_err in the fixtures)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def cl(x):     # [C, T, H, W] -> channels-last [T, H, W, C]
    return x.permute(1, 2, 3, 0).contiguous()


# ------------------------------------------------------------------------------------- convolution
@pytest.mark.parametrize("cin,cout,T,H,W", [(96, 96, 2, 9, 14), (32, 64, 1, 6, 8), (192, 384, 3, 5, 7), (384, 192, 1, 8, 4),
                                            (128, 128, 4, 12, 10)])
def test_conv3d_causal_with_history(cin, cout, T, H, W):
    g = torch.Generator().manual_seed(cin + cout + T)
    x = bf((cin, T + 2, H, W), g)                      # two history frames in front
    w, b = bf((cout, cin, 3, 3, 3), g, (27 * cin) ** -0.5), bf((cout,), g, 0.1)
    ref = F.conv3d(F.pad(x.float()[None], (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0]      # [cout, T, H, W]
    out = ops.conv_igemm(cl(x).to(DEV), repack_conv(w).to(DEV), b.to(DEV), (3, 3, 3), T)
    assert out.shape == (T, H, W, cout)
    assert rel(out.float().permute(3, 0, 1, 2), ref) < 4e-3


@pytest.mark.parametrize("cin,cout,T,H,W,resid", [(96, 96, 2, 32, 48, False), (96, 96, 1, 23, 37, True), (192, 192, 2, 16, 16, True),
                                                  (192, 384, 1, 20, 33, False), (384, 192, 3, 17, 16, False), (32, 96, 4, 40, 24, True)])
def test_conv3d_halo_structure(cin, cout, T, H, W, resid):
    """The halo-tile kernel (16 x 16 output patches, the 18 x 18 input halo staged once per channel slice and frame, the
    taps as shifted fragment reads): full and ragged patches (H, W not multiples of 16), both channel configurations
    (Cout % 192 == 0: one tap per cluster; Cout = 96: a kernel row per cluster), 1 .. 12 channel slices, the residual
    epilogue; against fp32 conv3d and against the gather-per-tap kernel (different summation order: close, not equal)."""
    g = torch.Generator().manual_seed(cin + cout + T + H)
    x = bf((cin, T + 2, H, W), g)
    w, b = bf((cout, cin, 3, 3, 3), g, (27 * cin) ** -0.5), bf((cout,), g, 0.1)
    r = bf((T, H, W, cout), g) if resid else None
    ref = F.conv3d(F.pad(x.float()[None], (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
    if resid:
        ref = ref + r.float()
    xd, wd, bd, rd = cl(x).to(DEV), repack_conv(w).to(DEV), b.to(DEV), (r.to(DEV) if resid else None)
    out = ops.conv_igemm(xd, wd, bd, (3, 3, 3), T, resid=rd, structure="halo")
    assert out.shape == (T, H, W, cout)
    assert rel(out.float(), ref) < 4e-3
    old = ops.conv_igemm(xd, wd, bd, (3, 3, 3), T, resid=rd, structure="igemm")
    assert rel(out.float(), old.float()) < 3e-3
    assert torch.equal(out, ops.conv_igemm(xd, wd, bd, (3, 3, 3), T, resid=rd))          # the automatic choice
    for _ in range(3):                                                                    # counted waits: repeatable
        assert torch.equal(out, ops.conv_igemm(xd, wd, bd, (3, 3, 3), T, resid=rd, structure="halo"))


@pytest.mark.parametrize("cin,cout,T,h,w_", [(192, 96, 2, 16, 24), (384, 192, 1, 9, 13), (96, 96, 3, 8, 8)])
def test_conv2d_upsample_halo_structure(cin, cout, T, h, w_):
    """Per-frame 3 x 3 convolution behind the fused nearest 2x upsampling in the halo kernel (10 x 10 input halo)."""
    g = torch.Generator().manual_seed(cin + cout + h)
    x = bf((T, cin, h, w_), g)
    w, b = bf((cout, cin, 3, 3), g, (9 * cin) ** -0.5), bf((cout,), g, 0.1)
    up = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest")
    ref = F.conv2d(up, w.float(), b.float(), padding=1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = ops.conv_igemm(xd, repack_conv(w).to(DEV), b.to(DEV), (1, 3, 3), T, upsample=True, structure="halo")
    assert out.shape == (T, 2 * h, 2 * w_, cout)
    assert rel(out.float().permute(0, 3, 1, 2), ref) < 4e-3


def test_conv_halo_rejects_what_it_cannot_do():
    x = torch.zeros(3, 20, 20, 64, dtype=torch.bfloat16, device=DEV)
    w = repack_conv(torch.zeros(64, 64, 3, 3, 3, dtype=torch.bfloat16)).to(DEV)
    with pytest.raises(sfa._lib.SfHipError, match="halo"):
        ops.conv_igemm(x, w, torch.zeros(64, dtype=torch.bfloat16, device=DEV), (3, 3, 3), 1, structure="halo")     # Cout = 64


def test_conv3d_residual_and_1x1():
    g = torch.Generator().manual_seed(5)
    cin, cout, T, H, W = 64, 96, 2, 7, 9
    x = bf((cin, T, H, W), g)
    w, b = bf((cout, cin, 1, 1, 1), g, cin ** -0.5), bf((cout,), g, 0.1)
    r = bf((T, H, W, cout), g)
    ref = F.conv3d(x.float()[None], w.float(), b.float())[0].permute(1, 2, 3, 0) + r.float()
    out = ops.conv_igemm(cl(x).to(DEV), repack_conv(w).to(DEV), b.to(DEV), (1, 1, 1), T, resid=r.to(DEV))
    assert rel(out.float(), ref) < 4e-3


def test_time_conv_interleaves_frames():
    """(3,1,1) causal conv C -> 2C, channel halves become frames 2t / 2t+1 (vae.py:134-137)."""
    g = torch.Generator().manual_seed(6)
    c, T, H, W = 64, 2, 5, 6
    x = bf((c, T + 2, H, W), g)
    w, b = bf((2 * c, c, 3, 1, 1), g, (3 * c) ** -0.5), bf((2 * c,), g, 0.1)
    y = F.conv3d(x.float()[None], w.float(), b.float())          # [1, 2c, T, H, W]
    y = y.reshape(1, 2, c, T, H, W)
    ref = torch.stack((y[:, 0], y[:, 1]), 3).reshape(c, 2 * T, H, W)
    out = ops.conv_igemm(cl(x).to(DEV), repack_conv(w).to(DEV), b.to(DEV), (3, 1, 1), T, interleave=True)
    assert out.shape == (2 * T, H, W, c)
    assert rel(out.float().permute(3, 0, 1, 2), ref) < 4e-3


def test_conv2d_with_fused_nearest_upsample():
    g = torch.Generator().manual_seed(7)
    cin, cout, T, h, w_ = 128, 64, 3, 5, 7
    x = bf((T, cin, h, w_), g)
    w, b = bf((cout, cin, 3, 3), g, (9 * cin) ** -0.5), bf((cout,), g, 0.1)
    up = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest")
    ref = F.conv2d(up, w.float(), b.float(), padding=1)          # [T, cout, 2h, 2w]
    out = ops.conv_igemm(x.permute(0, 2, 3, 1).contiguous().to(DEV), repack_conv(w).to(DEV), b.to(DEV), (1, 3, 3), T, upsample=True)
    assert out.shape == (T, 2 * h, 2 * w_, cout)
    assert rel(out.float().permute(0, 3, 1, 2), ref) < 4e-3


def test_head_conv_float_clamped_planar():
    g = torch.Generator().manual_seed(8)
    cin, T, H, W = 96, 2, 10, 12
    x = bf((cin, T + 2, H, W), g)
    w, b = bf((3, cin, 3, 3, 3), g, 3.0 * (27 * cin) ** -0.5), bf((3,), g, 0.1)
    ref = F.conv3d(F.pad(x.float()[None], (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].clamp(-1, 1).permute(1, 0, 2, 3)
    out = ops.conv_igemm(cl(x).to(DEV), repack_conv(w).to(DEV), b.to(DEV), (3, 3, 3), T, clamp_f32=True)
    assert out.shape == (T, 3, H, W) and out.dtype == torch.float32
    assert float((ref.abs() >= 1).float().mean()) > 0.02          # the clamp is exercised
    assert (out.cpu() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("H,W", [(32, 48), (19, 37)])
def test_head_conv_halo_structure(H, W):
    """The 3-channel head (float, clamped, planar) in the halo kernel: 32-column tile of which 3 are real."""
    g = torch.Generator().manual_seed(H + W)
    cin, T = 96, 2
    x = bf((cin, T + 2, H, W), g)
    w, b = bf((3, cin, 3, 3, 3), g, 3.0 * (27 * cin) ** -0.5), bf((3,), g, 0.1)
    ref = F.conv3d(F.pad(x.float()[None], (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].clamp(-1, 1).permute(1, 0, 2, 3)
    xd, wd, bd = cl(x).to(DEV), repack_conv(w).to(DEV), b.to(DEV)
    out = ops.conv_igemm(xd, wd, bd, (3, 3, 3), T, clamp_f32=True, structure="halo")
    assert out.shape == (T, 3, H, W) and out.dtype == torch.float32
    assert float((ref.abs() >= 1).float().mean()) > 0.02
    assert (out.cpu() - ref).abs().max().item() < 2e-2
    old = ops.conv_igemm(xd, wd, bd, (3, 3, 3), T, clamp_f32=True, structure="igemm")
    assert (out - old).abs().max().item() < 2e-2


def test_conv_padded_input_channels():
    """decoder.conv1: 16 latent channels zero-padded to 32."""
    g = torch.Generator().manual_seed(9)
    cin, cout, H, W = 16, 128, 6, 8
    x = bf((cin, 3, H, W), g)
    w, b = bf((cout, cin, 3, 3, 3), g, (27 * cin) ** -0.5), bf((cout,), g, 0.1)
    ref = F.conv3d(F.pad(x.float()[None], (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0]
    xp = torch.zeros(3, H, W, 32, dtype=torch.bfloat16)
    xp[..., :cin] = cl(x)
    out = ops.conv_igemm(xp.to(DEV), repack_conv(w).to(DEV), b.to(DEV), (3, 3, 3), 1)
    assert rel(out.float().permute(3, 0, 1, 2), ref) < 4e-3


def test_conv_rejects_bad_arguments():
    x = torch.zeros(3, 4, 4, 48, dtype=torch.bfloat16, device=DEV)        # 48 channels: not a multiple of 32
    w = torch.zeros(32, 27 * 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(sfa._lib.SfHipError, match="multiple of 32"):
        ops.conv_igemm(x, w, torch.zeros(32, dtype=torch.bfloat16, device=DEV), (3, 3, 3), 1)
    with pytest.raises(ValueError, match="do not cover"):
        ops.conv_igemm(torch.zeros(2, 4, 4, 32, dtype=torch.bfloat16, device=DEV), w, torch.zeros(32, dtype=torch.bfloat16, device=DEV),
                       (3, 3, 3), 1)


# ------------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("C", [32, 96, 192, 384])
@pytest.mark.parametrize("silu", [True, False])
def test_rmsnorm_silu(C, silu):
    g = torch.Generator().manual_seed(C)
    x, gamma = bf((3, 5, 7, C), g, 2.0), (1 + 0.1 * torch.randn(C, generator=g)).to(torch.bfloat16)
    ref = F.normalize(x.float(), dim=-1) * C ** 0.5 * gamma.float()
    if silu:
        ref = F.silu(ref)
    out = ops.rmsnorm_silu_cl(x.to(DEV), gamma.to(DEV), silu)
    assert rel(out.float(), ref) < 4e-3


def test_softmax_rows_with_padding():
    g = torch.Generator().manual_seed(3)
    s = torch.randn(50, 48, generator=g) * 30
    out = ops.softmax_rows(s.to(DEV), 0.25, cols_padded=64)
    ref = torch.softmax(s * 0.25, dim=-1)
    assert out.shape == (50, 64) and float(out[:, 48:].abs().max()) == 0.0
    assert rel(out[:, :48].float(), ref) < 4e-3


def test_gemm_f32_epilogue():
    g = torch.Generator().manual_seed(4)
    a, w = bf((100, 128), g), bf((52, 128), g)
    out = ops.gemm(a.to(DEV), w.to(DEV), None, "f32")
    assert out.dtype == torch.float32
    assert rel(out, a.float() @ w.float().t()) < 1e-5


# ------------------------------------------------------------------------------------ whole decode
def load_case(name):
    g = np.load(os.path.join(GOLD, f"vae_{name}.npz"))
    shape = {"reduced": vw.VAE_REDUCED, "full": vw.WAN_VAE}[name]
    sd = vw.synth_vae_state_dict(shape, seed=int(g["seed"]))
    return g, shape, sd


@pytest.mark.parametrize("name", ["reduced", "full"])
def test_decode_matches_reference_golden(name):
    g, shape, sd = load_case(name)
    vae = sfa.WanVAEWrapper(sd, device=DEV, shape=shape)
    lat = torch.from_numpy(g["latent"]).to(torch.bfloat16).to(DEV)
    out = vae.decode_to_pixel(lat, use_cache=False)
    gold = torch.from_numpy(g["pixels_f32"])
    assert out.shape == gold.shape and out.dtype == torch.float32
    assert float(out.abs().max()) <= 1.0
    err = rel(out, gold)
    assert err < TOL, f"{name}: rel err {err:.4f} (reference bf16 itself: {float(g['ref_bf16_rel_err']):.4f})"
    # decoding twice gives the same bits (clear_cache really clears)
    assert torch.equal(out, vae.decode_to_pixel(lat, use_cache=False))


def test_streaming_decode_equals_one_shot():
    g, shape, sd = load_case("reduced")
    vae = sfa.WanVAEWrapper(sd, device=DEV, shape=shape)
    lat = torch.from_numpy(g["latent"]).to(torch.bfloat16).to(DEV)
    whole = vae.decode_to_pixel(lat, use_cache=False)
    vae.model.clear_cache()
    a = vae.decode_to_pixel(lat[:, :1], use_cache=True)
    b = vae.decode_to_pixel(lat[:, 1:], use_cache=True)
    assert a.shape[1] == 1 and b.shape[1] == 4 * (lat.shape[1] - 1)
    assert torch.equal(torch.cat([a, b], 1), whole)
    # decode_chunk = the pipeline's streaming hook
    c0 = vae.decode_chunk(lat[:, :2], 0)
    c1 = vae.decode_chunk(lat[:, 2:], 1)
    assert torch.equal(torch.cat([c0, c1], 1), whole)


def test_streams_of_two_latent_sizes_do_not_share_their_position():
    """One decoder, two latent sizes streamed chunk by chunk IN TURN: each size keeps its own history state and its own
    position in it (fresh / frames decoded / window slot).  Sharing the position -- a second size resetting the first
    one's counters -- would read stale history frames without any error; the pixels must equal each size decoded alone."""
    shape = vw.VAE_REDUCED
    sd = vw.synth_vae_state_dict(shape, seed=5)
    g = torch.Generator().manual_seed(77)
    la = bf((7, shape.z_dim, 8, 12), g).to(DEV)
    lb = bf((6, shape.z_dim, 6, 10), g).to(DEV)
    alone = sfa.WanVAEDecoder(shape, sd, DEV)
    ref_a, ref_b = alone.decode(la), alone.decode(lb)
    dec = sfa.WanVAEDecoder(shape, sd, DEV)
    out_a, out_b = [], []
    cuts_a, cuts_b = [(0, 1), (1, 4), (4, 7)], [(0, 2), (2, 3), (3, 6)]
    for (a0, a1), (b0, b1) in zip(cuts_a, cuts_b):
        assert dec.frames_out(a1 - a0, 8, 12) == ((1 + 4 * (a1 - a0 - 1)) if a0 == 0 else 4 * (a1 - a0))
        out_a.append(dec.cached_decode(la[a0:a1]))
        out_b.append(dec.cached_decode(lb[b0:b1]))
    assert torch.equal(torch.cat(out_a), ref_a) and torch.equal(torch.cat(out_b), ref_b)
    with pytest.raises(ValueError, match="several latent sizes"):
        dec.frames_out(1)
    dec.clear_cache()                                   # resets every size's state AND position
    assert torch.equal(dec.cached_decode(lb), ref_b) and torch.equal(dec.cached_decode(la[:1]), ref_a[:1])


@pytest.mark.parametrize("fpc", [2, 3, 4, 7])
def test_grouped_frames_are_bit_identical_to_one_frame_per_call(fpc):
    """sf_vae_decode_frames on groups of latent frames (what fills the chip at the low-resolution stages) against the
    reference's iteration, one latent frame per call (vae.py:566-578): same bits, for whole clips, for ragged streaming
    chunks, and across restarts of the sliding history windows."""
    shape = vw.VAE_REDUCED
    sd = vw.synth_vae_state_dict(shape, seed=5)
    lat = torch.randn(1, 14, 16, 10, 6, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).to(DEV)
    one = sfa.WanVAEWrapper(sd, device=DEV, shape=shape, frames_per_call=1)
    ref = one.decode_to_pixel(lat, use_cache=False)
    assert ref.shape[1] == 1 + 4 * 13
    vae = sfa.WanVAEWrapper(sd, device=DEV, shape=shape, frames_per_call=fpc)
    assert torch.equal(vae.decode_to_pixel(lat, use_cache=False), ref)
    # streaming: chunks of 3 latent frames (the first holds the frame that follows the reset), then a ragged tail
    vae.model.clear_cache()
    parts = [vae.decode_to_pixel(lat[:, a:b], use_cache=True) for a, b in ((0, 3), (3, 6), (6, 9), (9, 12), (12, 14))]
    assert torch.equal(torch.cat(parts, 1), ref)
    # one frame at a time through the grouped decoder, too (windows slide and restart every few calls)
    vae.model.clear_cache()
    parts = [vae.decode_to_pixel(lat[:, a:a + 1], use_cache=True) for a in range(14)]
    assert torch.equal(torch.cat(parts, 1), ref)


def test_decode_frames_rejects_bad_window_arguments():
    shape = vw.VAE_REDUCED
    vae = sfa.WanVAEWrapper(vw.synth_vae_state_dict(shape, seed=0), device=DEV, shape=shape, frames_per_call=2)
    dec = vae.model
    z = torch.zeros(3, 16, 8, 8, dtype=torch.bfloat16, device=DEV)
    state, scratch = dec._buffers(8, 8)
    out = torch.empty(12, 3, 64, 64, dtype=torch.float32, device=DEV)
    K = dec.window_frames
    op = torch.ops.sf_hip.vae_decode_frames
    with pytest.raises((RuntimeError, ValueError), match="decoded alone"):
        op(dec._handle, state, scratch, z[:2], out, 8, 8, K, 0, 0, 0)            # the first chunk with a second frame
    with pytest.raises(ValueError, match="out holds"):
        op(dec._handle, state, scratch, z[:2], out[:7], 8, 8, K, 1, 1, 1)        # 2 latent frames = 8 pixel frames
    with pytest.raises(ValueError, match="uint8"):
        op(dec._handle, state.view(torch.int16), scratch, z[:1], out, 8, 8, K, 1, 1, 1)
    with pytest.raises(RuntimeError, match="n_frames"):
        op(dec._handle, state, scratch, z, out, 8, 8, K, 1, 1, 1)                # 3 frames > window_frames - 1
    with pytest.raises(RuntimeError, match="does not fit"):
        op(dec._handle, state, scratch, z[:2], out, 8, 8, K, 2, 2, 2)            # slots 2, 3 of 3
    with pytest.raises(RuntimeError, match="history_at"):
        op(dec._handle, state, scratch, z[:1], out, 8, 8, K, 2, 1, 2)            # a restart must go to window 0


def test_decode_matches_oracle_at_another_size_and_batch():
    shape = vw.VAE_REDUCED
    sd = vw.synth_vae_state_dict(shape, seed=3)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(2, 2, 16, 10, 6, generator=g).to(torch.bfloat16)
    cfg = vo.VaeOracleConfig(dim=shape.dim)
    ref, _ = vo.decode_to_pixel(cfg, vo.prepare_weights(sd, torch.float32), lat, torch.tensor(vw.LATENT_MEAN), torch.tensor(vw.LATENT_STD))
    out = sfa.WanVAEWrapper(sd, device=DEV, shape=shape).decode_to_pixel(lat.to(DEV))
    assert out.shape == ref.shape == (2, 5, 3, 80, 48)
    assert rel(out, ref) < TOL


def test_use_cache_requires_batch_one():
    shape = vw.VAE_REDUCED
    vae = sfa.WanVAEWrapper(vw.synth_vae_state_dict(shape, seed=0), device=DEV, shape=shape)
    with pytest.raises(AssertionError, match="Batch size must be 1"):
        vae.decode_to_pixel(torch.zeros(2, 1, 16, 4, 4, dtype=torch.bfloat16, device=DEV), use_cache=True)
    with pytest.raises(NotImplementedError):
        vae.encode_to_latent(torch.zeros(1, 3, 1, 32, 32, device=DEV))
