"""Shape description and weight handling of the umT5 text ENCODER (prompt -> [512, 4096] embeddings).

Key names are those of the reference's `T5Encoder.state_dict()` (wan/modules/t5.py:266-312), i.e. of
`models_t5_umt5-xxl-enc-bf16.pth` as `WanTextEncoder` loads it (utils/wan_wrapper.py:15-34).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, Tuple

import torch

Tensor = torch.Tensor


@dataclass(frozen=True)
class T5Shape:
    """Encoder-shaping arguments of `umt5_xxl` (wan/modules/t5.py:456-469); shared_pos is False: every
    block owns a relative-position embedding."""
    vocab_size: int = 256384
    dim: int = 4096
    dim_attn: int = 4096
    dim_ffn: int = 10240
    num_heads: int = 64
    num_layers: int = 24
    num_buckets: int = 32
    max_dist: int = 128
    eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.dim_attn // self.num_heads

    def as_dict(self) -> dict:
        return asdict(self)


UMT5_XXL = T5Shape()
# reduced encoder for the parity fixtures (head_dim stays 64, widths multiples of 512 for the norm kernels)
T5_REDUCED = T5Shape(vocab_size=512, dim=512, dim_attn=512, dim_ffn=1024, num_heads=8, num_layers=2)


def t5_param_shapes(s: T5Shape) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {"token_embedding.weight": (s.vocab_size, s.dim)}
    for i in range(s.num_layers):
        p = f"blocks.{i}."
        out[p + "norm1.weight"] = (s.dim,)
        for l in ("q", "k", "v"):
            out[p + f"attn.{l}.weight"] = (s.dim_attn, s.dim)
        out[p + "attn.o.weight"] = (s.dim, s.dim_attn)
        out[p + "norm2.weight"] = (s.dim,)
        out[p + "ffn.gate.0.weight"] = (s.dim_ffn, s.dim)
        out[p + "ffn.fc1.weight"] = (s.dim_ffn, s.dim)
        out[p + "ffn.fc2.weight"] = (s.dim, s.dim_ffn)
        out[p + "pos_embedding.embedding.weight"] = (s.num_buckets, s.num_heads)
    out["norm.weight"] = (s.dim,)
    return out


def synth_t5_state_dict(s: T5Shape, seed: int = 0, dtype=torch.bfloat16) -> Dict[str, Tensor]:
    """Seeded random-init encoder weights on the CPU, with the distributions of the reference's
    `init_weights` (t5.py:27-44) except: norm weights ~ 1 + N(0, .1) instead of ones, and the relative-
    position embeddings ~ N(0, 1) (the reference's std of 0.016 would leave the bias term untested).
    Drawn tensor by tensor in `t5_param_shapes` order."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for name, shape in t5_param_shapes(s).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name == "token_embedding.weight":
            t = torch.randn(shape, generator=g)
        elif name.endswith("pos_embedding.embedding.weight"):
            t = torch.randn(shape, generator=g)
        elif name.endswith("attn.q.weight"):
            t = torch.randn(shape, generator=g) * (s.dim * s.head_dim) ** -0.5      # m.dim_attn there is the per-layer dim_attn
        elif name.endswith("attn.o.weight"):
            t = torch.randn(shape, generator=g) * (s.dim_attn) ** -0.5
        elif name.endswith("ffn.fc2.weight"):
            t = torch.randn(shape, generator=g) * s.dim_ffn ** -0.5
        else:
            t = torch.randn(shape, generator=g) * s.dim ** -0.5
        sd[name] = t.to(dtype)
    return sd
