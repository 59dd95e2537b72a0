"""Model shape description and weight handling for the causal Wan DiT hot path.

Key names are those of the reference's `CausalWanModel.state_dict()`
(wan/modules/causal_model.py:457-478) so a reference checkpoint loads unchanged; the
reference's `generator` checkpoints prefix them with `model.` (utils/wan_wrapper.py:139).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, Iterable, Tuple

import torch

Tensor = torch.Tensor


@dataclass(frozen=True)
class WanShape:
    """Constructor arguments of CausalWanModel (wan/modules/causal_model.py:381-398)."""
    dim: int = 1536
    ffn_dim: int = 8960
    num_heads: int = 12
    num_layers: int = 30
    in_dim: int = 16
    out_dim: int = 16
    freq_dim: int = 256
    text_dim: int = 4096
    text_len: int = 512
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6
    local_attn_size: int = -1
    sink_size: int = 0

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads

    def as_dict(self) -> dict:
        return asdict(self)

    def replace(self, **kw) -> "WanShape":
        d = asdict(self)
        d.update(kw)
        d["patch_size"] = tuple(d["patch_size"])
        return WanShape(**d)


# wan/configs/wan_t2v_1_3B.py:20-29 and wan/configs/wan_t2v_14B.py:20-29
WAN_1_3B = WanShape(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)
WAN_14B = WanShape(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40)
# reduced shape used by the parity tests / golden fixtures (same head_dim = 128)
WAN_REDUCED = WanShape(dim=512, ffn_dim=1024, num_heads=4, num_layers=2, text_dim=256)

NAMED_SHAPES = {"Wan2.1-T2V-1.3B": WAN_1_3B, "Wan2.1-T2V-14B": WAN_14B, "reduced": WAN_REDUCED}


POSE_DIM = 5120  # UniAnimate pose-embedding width (causal_model.py:493-503)


def param_shapes(s: WanShape, pose: bool = False) -> Dict[str, Tuple[int, ...]]:
    """Every tensor the hot path reads, with its shape.  `pose=True` appends the optional
    `pose_proj` Linear(5120, dim) of the fork's pose conditioning (Identity when dim == 5120)."""
    C, Fd = s.dim, s.ffn_dim
    P = s.patch_size[0] * s.patch_size[1] * s.patch_size[2]
    out: Dict[str, Tuple[int, ...]] = {
        "patch_embedding.weight": (C, s.in_dim, *s.patch_size),
        "patch_embedding.bias": (C,),
        "text_embedding.0.weight": (C, s.text_dim), "text_embedding.0.bias": (C,),
        "text_embedding.2.weight": (C, C), "text_embedding.2.bias": (C,),
        "time_embedding.0.weight": (C, s.freq_dim), "time_embedding.0.bias": (C,),
        "time_embedding.2.weight": (C, C), "time_embedding.2.bias": (C,),
        "time_projection.1.weight": (6 * C, C), "time_projection.1.bias": (6 * C,),
        "head.head.weight": (P * s.out_dim, C), "head.head.bias": (P * s.out_dim,),
        "head.modulation": (1, 2, C),
    }
    for i in range(s.num_layers):
        p = f"blocks.{i}."
        out[p + "modulation"] = (1, 6, C)
        out[p + "norm3.weight"] = (C,)
        out[p + "norm3.bias"] = (C,)
        for a in ("self_attn", "cross_attn"):
            for l in ("q", "k", "v", "o"):
                out[p + f"{a}.{l}.weight"] = (C, C)
                out[p + f"{a}.{l}.bias"] = (C,)
            out[p + f"{a}.norm_q.weight"] = (C,)
            out[p + f"{a}.norm_k.weight"] = (C,)
        out[p + "ffn.0.weight"] = (Fd, C)
        out[p + "ffn.0.bias"] = (Fd,)
        out[p + "ffn.2.weight"] = (C, Fd)
        out[p + "ffn.2.bias"] = (C,)
    if pose and C != POSE_DIM:   # appended LAST: the seeded draws of all other tensors do not move
        out["pose_proj.weight"] = (C, POSE_DIM)
        out["pose_proj.bias"] = (C,)
    return out


def synth_state_dict(s: WanShape, seed: int = 0, dtype=torch.bfloat16,
                     modulation_gain: float = 1.0, pose: bool = False) -> Dict[str, Tensor]:
    """Seeded random-init weights on the CPU (SURVEY.md section 8d recipe).

    Follows the distributions of `CausalWanModel.init_weights`
    (wan/modules/causal_model.py:1106-1128: Xavier-uniform Linears, N(0,.02) for the
    text/time MLPs, modulation ~ randn/sqrt(C)) EXCEPT where that init would make the
    test vacuous: `head.head.weight` ~ N(0,.02) instead of zeros (output would be 0),
    every bias ~ N(0,.02) instead of zeros (so bias epilogues are exercised), and the
    norm scale vectors ~ 1 + N(0,.1) (norm3.bias ~ N(0,.1)).  Deterministic for a given
    torch version: drawn tensor-by-tensor in `param_shapes` order from one CPU generator.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for name, shape in param_shapes(s, pose).items():
        if name.endswith("modulation"):
            t = torch.randn(shape, generator=g) * (modulation_gain / math.sqrt(s.dim))
        elif name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm3.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("norm3.bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name.startswith("text_embedding") or name.startswith("time_embedding") or name == "head.head.weight":
            t = 0.02 * torch.randn(shape, generator=g)
        else:  # Xavier uniform on [out, in(flattened)]
            fan_out = shape[0]
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            a = math.sqrt(6.0 / (fan_in + fan_out))
            t = torch.empty(shape).uniform_(-a, a, generator=g)
        sd[name] = t.to(dtype)
    return sd


def strip_prefix(sd: Dict[str, Tensor], prefixes: Iterable[str] = ("model.",)) -> Dict[str, Tensor]:
    """Accept the reference's `generator` checkpoints (keys prefixed `model.`,
    inference.py:69-71)."""
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out


def merge_lora(sd: Dict[str, Tensor], alpha: float, rank: int) -> Dict[str, Tensor]:
    """Fold LoRA adapters into their base weights: W += (alpha/rank) * B @ A.

    The reference keeps `base`, `lora_A`, `lora_B` separate and evaluates
    base(x) + B(A(x)) * alpha/rank at run time (utils/lora.py:47-50); the adapters wrap
    q,k,v,o of both attentions and ffn.0/ffn.2 (utils/lora.py:116-140), which renames
    `<lin>.weight` to `<lin>.base.weight`.  Merging offline is exact up to one bf16
    rounding of the merged matrix and removes 14 % of run-time FLOPs."""
    out: Dict[str, Tensor] = {}
    scale = alpha / rank
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        if ".base." in k:
            stem = k.split(".base.")[0]
            leaf = k.split(".base.")[1]
            if leaf == "weight":
                A = sd.get(stem + ".lora_A.weight")
                B = sd.get(stem + ".lora_B.weight")
                if A is not None and B is not None:
                    v = (v.float() + scale * (B.float() @ A.float())).to(v.dtype)
            out[stem + "." + leaf] = v
        else:
            out[k] = v
    return out
