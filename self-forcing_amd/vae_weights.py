"""Shape description and weight handling of the Wan VAE *decoder* (latents -> pixels).

Key names are those of the reference's `WanVAE_.state_dict()` restricted to what `decode` reads
(wan/modules/vae.py:369-429 Decoder3d, :503 conv2), so `Wan2.1_VAE.pth` loads unchanged (its encoder /
conv1 tensors are ignored).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import torch

Tensor = torch.Tensor

# per-channel latent statistics (utils/wan_wrapper.py:59-68; the same numbers as wan/modules/vae.py:634-643)
LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


@dataclass(frozen=True)
class VaeShape:
    """Constructor arguments of WanVAE_ that shape the decoder (wan/modules/vae.py:591-603:
    dim 96, z_dim 16, dim_mult [1,2,4,4], 2 res blocks, temporal upsampling in the first two stages)."""
    dim: int = 96
    z_dim: int = 16
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temperal_upsample: Tuple[bool, ...] = (True, True, False)   # reversed temperal_downsample (vae.py:501)

    def as_dict(self) -> dict:
        return asdict(self)

    @property
    def dims(self) -> List[int]:
        return [self.dim * u for u in (self.dim_mult[-1],) + tuple(self.dim_mult[::-1])]

    @property
    def spatial_factor(self) -> int:
        return 2 ** (len(self.dim_mult) - 1)

    @property
    def temporal_factor(self) -> int:
        return 2 ** sum(1 for t in self.temperal_upsample if t)


WAN_VAE = VaeShape()
# reduced decoder for the parity fixtures (every channel count stays a multiple of 32)
VAE_REDUCED = VaeShape(dim=32)


@dataclass(frozen=True)
class ResBlockSpec:
    prefix: str
    in_dim: int
    out_dim: int


@dataclass(frozen=True)
class ResampleSpec:
    prefix: str
    dim: int
    mode: str   # 'upsample2d' | 'upsample3d'


def decoder_layout(s: VaeShape):
    """The module sequence of Decoder3d (vae.py:390-421): returns (middle, upsamples) where middle is
    [ResBlockSpec, 'attn', ResBlockSpec] and upsamples a list of ResBlockSpec / ResampleSpec in order."""
    dims = s.dims
    middle = [ResBlockSpec("decoder.middle.0.", dims[0], dims[0]), "decoder.middle.1.",
              ResBlockSpec("decoder.middle.2.", dims[0], dims[0])]
    ups = []
    idx = 0
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            in_dim = in_dim // 2
        for _ in range(s.num_res_blocks + 1):
            ups.append(ResBlockSpec(f"decoder.upsamples.{idx}.", in_dim, out_dim))
            idx += 1
            in_dim = out_dim
        if i != len(s.dim_mult) - 1:
            ups.append(ResampleSpec(f"decoder.upsamples.{idx}.", out_dim,
                                    "upsample3d" if s.temperal_upsample[i] else "upsample2d"))
            idx += 1
    return middle, ups


def vae_param_shapes(s: VaeShape) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}
    z, d0 = s.z_dim, s.dims[0]
    out["conv2.weight"] = (z, z, 1, 1, 1)
    out["conv2.bias"] = (z,)
    out["decoder.conv1.weight"] = (d0, z, 3, 3, 3)
    out["decoder.conv1.bias"] = (d0,)

    def res(spec: ResBlockSpec):
        p = spec.prefix
        out[p + "residual.0.gamma"] = (spec.in_dim, 1, 1, 1)
        out[p + "residual.2.weight"] = (spec.out_dim, spec.in_dim, 3, 3, 3)
        out[p + "residual.2.bias"] = (spec.out_dim,)
        out[p + "residual.3.gamma"] = (spec.out_dim, 1, 1, 1)
        out[p + "residual.6.weight"] = (spec.out_dim, spec.out_dim, 3, 3, 3)
        out[p + "residual.6.bias"] = (spec.out_dim,)
        if spec.in_dim != spec.out_dim:
            out[p + "shortcut.weight"] = (spec.out_dim, spec.in_dim, 1, 1, 1)
            out[p + "shortcut.bias"] = (spec.out_dim,)

    middle, ups = decoder_layout(s)
    res(middle[0])
    a = middle[1]
    out[a + "norm.gamma"] = (d0, 1, 1)
    out[a + "to_qkv.weight"] = (3 * d0, d0, 1, 1)
    out[a + "to_qkv.bias"] = (3 * d0,)
    out[a + "proj.weight"] = (d0, d0, 1, 1)
    out[a + "proj.bias"] = (d0,)
    res(middle[2])
    for spec in ups:
        if isinstance(spec, ResBlockSpec):
            res(spec)
        else:
            out[spec.prefix + "resample.1.weight"] = (spec.dim // 2, spec.dim, 3, 3)
            out[spec.prefix + "resample.1.bias"] = (spec.dim // 2,)
            if spec.mode == "upsample3d":
                out[spec.prefix + "time_conv.weight"] = (2 * spec.dim, spec.dim, 3, 1, 1)
                out[spec.prefix + "time_conv.bias"] = (2 * spec.dim,)
    dl = s.dims[-1]
    out["decoder.head.0.gamma"] = (dl, 1, 1, 1)
    out["decoder.head.2.weight"] = (3, dl, 3, 3, 3)
    out["decoder.head.2.bias"] = (3,)
    return out


def synth_vae_state_dict(s: VaeShape, seed: int = 0, dtype=torch.bfloat16) -> Dict[str, Tensor]:
    """Seeded random-init decoder weights on the CPU (there is no network for Wan2.1_VAE.pth).
    Convolutions ~ U(-a, a) with a = sqrt(3 / fan_in) (unit gain: activations keep their scale through
    the 30-odd layers), biases ~ N(0, .02), RMS-norm gammas ~ 1 + N(0, .1).  The attention projection
    `proj` is NOT zero (the reference zero-initialises it, vae.py:239, which would make the block the
    identity and the test vacuous).  Drawn tensor by tensor in `vae_param_shapes` order."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for name, shape in vae_param_shapes(s).items():
        if name.endswith("gamma"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            a = math.sqrt(3.0 / fan_in)
            t = torch.empty(shape).uniform_(-a, a, generator=g)
        sd[name] = t.to(dtype)
    return sd


def vae_decode_flops(s: VaeShape, lat_h: int, lat_w: int, latent_frames: int) -> float:
    """Algorithmic FLOPs (multiply-add = 2) of `decode` on `latent_frames` latent frames: every
    convolution at its true channel counts (2 * taps * Cin * Cout per output position), the attention
    block's projections and its two token-by-token products.  The first latent frame runs every stage
    at one frame (no time convolution); later frames double the frame count after each upsample3d."""
    middle, ups = decoder_layout(s)
    d0 = s.dims[0]

    def one(first: bool) -> float:
        h, w, t = lat_h, lat_w, 1
        pos = h * w
        fl = 2.0 * s.z_dim * s.z_dim * pos + 2.0 * 27 * s.z_dim * d0 * pos               # conv2, decoder.conv1
        fl += 2 * (2.0 * 27 * d0 * d0 * pos * 2)                                             # two middle res blocks
        fl += 2.0 * d0 * 3 * d0 * pos + 2.0 * d0 * d0 * pos + 4.0 * pos * pos * d0           # attention block
        for spec in ups:
            pos = t * h * w
            if isinstance(spec, ResBlockSpec):
                fl += 2.0 * 27 * spec.in_dim * spec.out_dim * pos + 2.0 * 27 * spec.out_dim * spec.out_dim * pos
                if spec.in_dim != spec.out_dim:
                    fl += 2.0 * spec.in_dim * spec.out_dim * pos
            else:
                if spec.mode == "upsample3d" and not first:
                    fl += 2.0 * 3 * spec.dim * 2 * spec.dim * pos
                    t *= 2
                h, w = 2 * h, 2 * w
                fl += 2.0 * 9 * spec.dim * (spec.dim // 2) * t * h * w
        fl += 2.0 * 27 * s.dims[-1] * 3 * t * h * w                                          # head
        return fl

    return one(True) + (latent_frames - 1) * one(False)
