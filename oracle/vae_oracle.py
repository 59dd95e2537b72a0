"""CPU oracle for the Wan VAE decode (latents -> pixels), SURVEY.md section 8f row 1.

TEST INFRASTRUCTURE ONLY -- only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it, never the product package.

A from-scratch CPU restatement (torch CPU tensors as the array library) of
`WanVAEWrapper.decode_to_pixel` -> `WanVAE_.decode` / `cached_decode` -> `Decoder3d.forward`
(utils/wan_wrapper.py:95-117, wan/modules/vae.py:556-593, :423-472).  Where the reference threads a
list of 32 `feat_cache` tensors and an index counter through the modules, this file keeps, per causal
convolution, the last two input frames ("history") -- the same information -- and states the two
quirks of the bookkeeping explicitly (see `Resample` below).

Parity status: PINNED.  `oracle/make_golden_vae.py` imports the reference's `WanVAE_` on CPU, loads the
seeded weights of `self_forcing_amd.vae_weights.synth_vae_state_dict` into it, runs `decode` /
`cached_decode` and stores inputs + outputs under `tests/golden/vae_*.npz`;
`tests/test_vae_oracle_golden.py` checks this file against them (fp32 <= 1e-5 relative).  The reference
ships no golden vectors for the VAE, so the generated fixtures are the pin.

Numeric modes as in `wan_oracle.py`: weights prepared in float32 = math oracle (bf16-rounded weights,
fp32 arithmetic); weights in bfloat16 = every torch op rounds to bf16 like the reference run under
`pipeline.to(dtype=torch.bfloat16)` (inference.py:72).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
CACHE_T = 2  # vae.py:14


@dataclass
class VaeOracleConfig:
    """Decoder-shaping arguments of WanVAE_ (vae.py:591-603)."""
    dim: int = 96
    z_dim: int = 16
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temperal_upsample: Tuple[bool, ...] = (True, True, False)

    @property
    def dims(self) -> List[int]:
        return [self.dim * u for u in (self.dim_mult[-1],) + tuple(self.dim_mult[::-1])]


def prepare_weights(sd: Dict[str, Tensor], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: v.detach().to("cpu").to(dtype) for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def causal_conv3d(x: Tensor, w: Tensor, b: Tensor, hist: Optional[Tensor]) -> Tensor:
    """CausalConv3d.forward (vae.py:17-38): pad (kw//2, kh//2) spatially on both sides and
    2*(kt//2) frames IN FRONT; cached frames stand in for that many of the front zeros.
    x [B,C,T,H,W]; hist None or [B,C,<=2,H,W]."""
    kt, kh, kw = w.shape[2:]
    pt = 2 * (kt // 2)
    if hist is not None and pt > 0:
        x = torch.cat([hist.to(x.dtype), x], dim=2)
        pt -= hist.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pt, 0))
    return F.conv3d(x, w, b)


def rms_norm(x: Tensor, gamma: Tensor) -> Tensor:
    """RMS_norm (vae.py:41-56): L2-normalise over channels (eps 1e-12 on the norm, F.normalize),
    times sqrt(C), times gamma.  x is channel-first [B,C,...]."""
    c = x.shape[1]
    g = gamma.reshape(1, c, *([1] * (x.dim() - 2)))
    return F.normalize(x, dim=1) * (c ** 0.5) * g


def _next_hist(x: Tensor, hist: Optional[Tensor]) -> Tensor:
    """The cache update every cached conv performs (vae.py:206-214): keep the last two input frames;
    a one-frame chunk borrows the last frame of the previous cache."""
    cache_x = x[:, :, -CACHE_T:].clone()
    if cache_x.shape[2] < 2 and hist is not None:
        cache_x = torch.cat([hist[:, :, -1:].to(cache_x.dtype), cache_x], dim=2)
    return cache_x


class DecoderState:
    """Per-stream history of every causal convolution, keyed by the conv's weight name; `rep` holds
    the upsample3d blocks that have seen their first chunk (the reference's 'Rep' marker)."""

    def __init__(self):
        self.hist: Dict[str, Tensor] = {}
        self.rep: Dict[str, bool] = {}


def cached_conv(st: DecoderState, name: str, x: Tensor, W: Dict[str, Tensor]) -> Tensor:
    h = st.hist.get(name)
    y = causal_conv3d(x, W[name + ".weight"], W[name + ".bias"], h)
    st.hist[name] = _next_hist(x, h)
    return y


def residual_block(st: DecoderState, p: str, x: Tensor, W: Dict[str, Tensor]) -> Tensor:
    """ResidualBlock.forward (vae.py:202-221): shortcut (1x1x1 conv when the width changes, never
    cached: kt = 1), RMS-norm -> SiLU -> conv3 -> RMS-norm -> SiLU -> conv3, plus shortcut."""
    if (p + "shortcut.weight") in W:
        h = causal_conv3d(x, W[p + "shortcut.weight"], W[p + "shortcut.bias"], None)
    else:
        h = x
    y = F.silu(rms_norm(x, W[p + "residual.0.gamma"]))
    y = cached_conv(st, p + "residual.2", y, W)
    y = F.silu(rms_norm(y, W[p + "residual.3.gamma"]))
    y = cached_conv(st, p + "residual.6", y, W)
    return y + h


def attention_block(p: str, x: Tensor, W: Dict[str, Tensor]) -> Tensor:
    """AttentionBlock.forward (vae.py:241-264): per frame, single head over the H*W positions, head
    width = C; RMS-norm (2-D gamma), 1x1 qkv conv, softmax(q k^T / sqrt(C)) v, 1x1 proj, + identity."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, W[p + "norm.gamma"])
    qkv = F.conv2d(y, W[p + "to_qkv.weight"], W[p + "to_qkv.bias"])          # [bt, 3c, h, w]
    qkv = qkv.reshape(b * t, 3 * c, h * w).permute(0, 2, 1)                    # [bt, hw, 3c]
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    if q.dtype == torch.float32:
        s = torch.softmax((q @ k.transpose(1, 2)) / math.sqrt(c), dim=-1)
        o = s @ v
    else:
        o = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)).squeeze(1)
    o = o.permute(0, 2, 1).reshape(b * t, c, h, w)
    o = F.conv2d(o, W[p + "proj.weight"], W[p + "proj.bias"])
    o = o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return o + x


def resample_up(st: DecoderState, p: str, mode: str, x: Tensor, W: Dict[str, Tensor]) -> Tensor:
    """Resample.forward for 'upsample2d' / 'upsample3d' (vae.py:104-141).

    upsample3d doubles the frame count with a (3,1,1) causal conv C -> 2C whose output channels
    [0:C] / [C:2C] become frames 2t / 2t+1 (vae.py:134-137).  Two quirks of the cache bookkeeping:
      * the FIRST chunk a stream sees skips the time conv altogether ('Rep', vae.py:109-111): latent
        frame 0 yields one frame, every later latent frame two -> 1 + 4 (F-1) pixel frames in all;
      * the second chunk runs the time conv with NO history (two zero frames in front, vae.py:129-130),
        i.e. chunk 0's features never enter the time conv; the cache stored after it is [0, x]
        (vae.py:124-128).
    Then nearest-neighbour 2x spatial upsampling (computed in float, vae.py:61-65) and a 3x3 Conv2d
    C -> C/2 per frame."""
    b, c, t, h, w = x.shape
    if mode == "upsample3d":
        name = p + "time_conv"
        if not st.rep.get(name, False):
            st.rep[name] = True          # first chunk: marker only, x passes through
        else:
            hist = st.hist.get(name)     # None on the second chunk
            cache_x = x[:, :, -CACHE_T:].clone()
            if cache_x.shape[2] < 2:
                front = hist[:, :, -1:].to(cache_x.dtype) if hist is not None else torch.zeros_like(cache_x)
                cache_x = torch.cat([front, cache_x], dim=2)
            y = causal_conv3d(x, W[name + ".weight"], W[name + ".bias"], hist)
            st.hist[name] = cache_x
            y = y.reshape(b, 2, c, t, h, w)
            x = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, t * 2, h, w)
            t = t * 2
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest").to(x.dtype)
    y = F.conv2d(y, W[p + "resample.1.weight"], W[p + "resample.1.bias"], padding=1)
    return y.reshape(b, t, c // 2, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


# --------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------
def decoder_layout(cfg: VaeOracleConfig):
    """Module order of Decoder3d.__init__ (vae.py:386-416): list of ('res', prefix) / ('up', prefix, mode)."""
    dims = cfg.dims
    seq = []
    idx = 0
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(cfg.num_res_blocks + 1):
            seq.append(("res", f"decoder.upsamples.{idx}."))
            idx += 1
        if i != len(cfg.dim_mult) - 1:
            seq.append(("up", f"decoder.upsamples.{idx}.", "upsample3d" if cfg.temperal_upsample[i] else "upsample2d"))
            idx += 1
    return seq


def decoder_chunk(cfg: VaeOracleConfig, st: DecoderState, x: Tensor, W: Dict[str, Tensor]) -> Tensor:
    """Decoder3d.forward on one chunk of latent frames with caches (vae.py:423-472).  x [B,z,T,h,w]."""
    x = cached_conv(st, "decoder.conv1", x, W)
    x = residual_block(st, "decoder.middle.0.", x, W)
    x = attention_block("decoder.middle.1.", x, W)
    x = residual_block(st, "decoder.middle.2.", x, W)
    for item in decoder_layout(cfg):
        if item[0] == "res":
            x = residual_block(st, item[1], x, W)
        else:
            x = resample_up(st, item[1], item[2], x, W)
    x = F.silu(rms_norm(x, W["decoder.head.0.gamma"]))
    return cached_conv(st, "decoder.head.2", x, W)


def unscale_latent(z: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """z / (1/std) + mean per channel, with scale = [mean, 1/std] cast to the latent dtype first
    (utils/wan_wrapper.py:104-106, vae.py:559-561).  z [B,C,T,H,W]."""
    inv = 1.0 / std.to(z.dtype)          # the division happens AFTER the cast, as in the reference
    return z / inv.view(1, -1, 1, 1, 1) + mean.to(z.dtype).view(1, -1, 1, 1, 1)


def vae_decode(cfg: VaeOracleConfig, W: Dict[str, Tensor], z: Tensor, mean: Tensor, std: Tensor,
               state: Optional[DecoderState] = None) -> Tuple[Tensor, DecoderState]:
    """WanVAE_.decode / cached_decode (vae.py:556-593): un-scale, 1x1x1 conv2, then the decoder ONE
    latent frame at a time, concatenated over time.  `state=None` starts from cleared caches (decode);
    passing the returned state back in continues a stream (cached_decode).  z [B,C,T,H,W]."""
    st = state if state is not None else DecoderState()
    z = unscale_latent(z, mean, std)
    x = causal_conv3d(z, W["conv2.weight"], W["conv2.bias"], None)
    outs = [decoder_chunk(cfg, st, x[:, :, i:i + 1], W) for i in range(x.shape[2])]
    return torch.cat(outs, dim=2), st


def decode_to_pixel(cfg: VaeOracleConfig, W: Dict[str, Tensor], latent: Tensor, mean: Tensor, std: Tensor,
                    state: Optional[DecoderState] = None) -> Tuple[Tensor, DecoderState]:
    """WanVAEWrapper.decode_to_pixel (utils/wan_wrapper.py:95-117): latent [B,F,C,H,W] in the dtype
    of the weights; returns float32 [B, 1+4(F-1), 3, 8H, 8W] clamped to [-1, 1].  With a state the batch
    must be 1 (the reference asserts it, :100-101)."""
    wdtype = W["conv2.weight"].dtype
    zs = latent.to(wdtype).permute(0, 2, 1, 3, 4)
    if state is not None and zs.shape[0] != 1:
        raise AssertionError("Batch size must be 1 when using cache")
    outs = []
    st = state
    for u in zs:
        y, st_u = vae_decode(cfg, W, u.unsqueeze(0), mean, std, state)
        st = st_u
        outs.append(y.float().clamp_(-1, 1).squeeze(0))
    out = torch.stack(outs, dim=0).permute(0, 2, 1, 3, 4)
    return out, st
