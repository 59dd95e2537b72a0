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


# ------------------------------------------------------------------------------------------ LoRA files
LORA_DEFAULT_TARGETS = ("q", "k", "v", "o")        # utils/wan_wrapper.py:147
_LORA_PREFIXES = ("diffusion_model.", "model.", "pipe.dit.", "pipe.")     # utils/lora.py:97, :179


def lora_target_linears(s: WanShape, targets: Iterable[str] = LORA_DEFAULT_TARGETS):
    """Names of the Linears `apply_lora(model, target_modules=targets)` wraps (utils/lora.py:106-140): the attribute
    names in `targets` on BOTH attentions of every block (cross-attention subclasses the self-attention class), and
    `ffn.<idx>` entries on the block's feed-forward Sequential."""
    names = []
    attn = [t for t in targets if not t.startswith("ffn.")]
    ffn = [t for t in targets if t.startswith("ffn.") and t.split(".", 1)[1].isdigit()]
    for i in range(s.num_layers):
        for a in ("self_attn", "cross_attn"):
            names += [f"blocks.{i}.{a}.{t}" for t in attn if t in ("q", "k", "v", "o")]
        names += [f"blocks.{i}.{t}" for t in ffn if t in ("ffn.0", "ffn.2")]
    return names


def synth_lora_state_dict(s: WanShape, rank: int, seed: int = 0, targets: Iterable[str] = LORA_DEFAULT_TARGETS,
                          b_std: float = 0.05, dtype=torch.bfloat16, prefix: str = "") -> Dict[str, Tensor]:
    """Seeded NON-ZERO adapters for the parity tests, in the reference's key layout (`<linear>.lora_A.weight`
    [rank, in], `<linear>.lora_B.weight` [out, rank]).  A follows LoRALinear's init (kaiming_uniform(a=sqrt(5)) =
    U(-1/sqrt(in), 1/sqrt(in)), utils/lora.py:32); B ~ N(0, b_std) instead of the reference's zeros (:33), which would
    make every LoRA test vacuous."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = param_shapes(s)
    sd: Dict[str, Tensor] = {}
    for name in lora_target_linears(s, targets):
        out_f, in_f = shapes[name + ".weight"]
        bound = 1.0 / math.sqrt(in_f)
        sd[f"{prefix}{name}.lora_A.weight"] = torch.empty(rank, in_f).uniform_(-bound, bound, generator=g).to(dtype)
        sd[f"{prefix}{name}.lora_B.weight"] = (b_std * torch.randn(out_f, rank, generator=g)).to(dtype)
    return sd


def load_lora_file(path: str) -> Dict[str, Tensor]:
    """`.safetensors`, or a torch checkpoint read with weights_only=True (utils/lora.py:53-61)."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(str(path))
    return torch.load(str(path), map_location="cpu", weights_only=True)


def apply_lora_file(sd: Dict[str, Tensor], lora_sd: Dict[str, Tensor], s: WanShape, rank: int, alpha: float,
                    targets: Iterable[str] = LORA_DEFAULT_TARGETS):
    """What `apply_lora` + `load_lora_weights` (utils/lora.py:100-234) amount to at inference, with the adapters folded
    into the base matrices: for every (lora_B | lora_up, lora_A | lora_down) pair of the file whose target -- after
    dropping an optional `.default`, a known prefix and a trailing `.weight` -- is one of the wrapped Linears,
    W += (alpha / rank) * up @ down.  `rank` is the CONSTRUCTOR's rank, as in the reference (`module.scaling =
    alpha / module.rank`, :228-230), not the file's.  Pairs for other modules are skipped, as the reference skips them.
    A file without a single (up, down) pair raises ValueError as `load_lora_weights` does (:150-152); a file whose pairs
    all miss the wrapped Linears (a mis-prefixed or foreign adapter file) would leave the base model running silently:
    that case warns.  Returns (state dict, loaded, skipped)."""
    wrapped = set(lora_target_linears(s, targets))
    out = dict(sd)
    loaded = skipped = pairs = 0
    scale = alpha / rank
    for key_up in lora_sd:
        if "lora_B" in key_up:
            key_down = key_up.replace("lora_B", "lora_A")
        elif "lora_up" in key_up:
            key_down = key_up.replace("lora_up", "lora_down")
        else:
            continue
        if key_down not in lora_sd:
            continue
        pairs += 1
        parts = key_up.split(".")
        tok = "lora_B" if "lora_B" in parts else "lora_up" if "lora_up" in parts else None
        if tok is None:
            skipped += 1
            continue
        i = parts.index(tok)
        parts.pop(i)
        if i < len(parts) and parts[i] == "default":
            parts.pop(i)
        target = ".".join(parts)
        for pre in _LORA_PREFIXES:
            if target.startswith(pre):
                target = target[len(pre):]
                break
        if target.endswith(".weight") or target.endswith(".bias"):
            target = target.rsplit(".", 1)[0]
        if target.endswith(".base"):
            target = target[:-5]
        if target not in wrapped or target + ".weight" not in out:
            skipped += 1
            continue
        up, down = lora_sd[key_up].float(), lora_sd[key_down].float()
        W = out[target + ".weight"]
        if up.shape[0] != W.shape[0] or down.shape[1] != W.shape[1] or up.shape[1] != down.shape[0]:
            raise ValueError(f"LoRA pair for {target}: up {tuple(up.shape)} / down {tuple(down.shape)} do not fit weight {tuple(W.shape)}")
        # the reference copies the file's tensors into the adapters in the MODEL's dtype (bf16) before use (:223-227)
        up, down = up.to(W.dtype).float(), down.to(W.dtype).float()
        out[target + ".weight"] = (W.float() + scale * (up @ down)).to(W.dtype)
        loaded += 1
    if pairs == 0:
        raise ValueError("No LoRA pairs found (expected lora_A/lora_B or lora_up/lora_down).")
    if loaded == 0:
        import warnings
        warnings.warn(f"LoRA file holds {pairs} adapter pair(s) but NONE maps to a wrapped Linear of this model (targets "
                      f"{sorted(set(targets))}; first key: {next(iter(lora_sd))!r}): the base model runs unchanged", stacklevel=2)
    return out, loaded, skipped
