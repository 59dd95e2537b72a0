"""umT5 text encoder on the GPU behind the reference's `WanTextEncoder` surface.

Mirrors `utils/wan_wrapper.py:15-55`:

    enc = WanTextEncoder(state_dict=..., tokenizer=tok, device="cuda")
    cond = enc(text_prompts=["a red fox"])          # {"prompt_embeds": [B, 512, 4096] bf16, zero padded}

The encoder pass is ONE C call (`sf_t5_encode`, csrc/t5_encoder.hip) over the kernels of the rollout path
(bf16 MFMA GEMM, RMS norm) plus the relative-position-bias softmax.  There is no eager/CPU fallback.  The
tokenizer (the sentencepiece model of google/umt5-xxl) is not part of this path and is not shipped here:
pass any callable `tokenizer(list_of_str) -> (ids [B, 512] int64, mask [B, 512] int64)` -- the reference's
`HuggingfaceTokenizer(..., seq_len=512, clean='whitespace')(texts, return_mask=True, add_special_tokens=True)`
is one -- or call `encode_ids(ids, mask)` with token ids directly.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional

import torch

from . import _lib, torch_ops
from .t5_weights import T5Shape, UMT5_XXL, t5_param_shapes

Tensor = torch.Tensor


def relative_position_buckets(seq_len: int, num_buckets: int = 32, max_dist: int = 128) -> Tensor:
    """bucket[d + seq_len - 1] for every relative position d = key - query in (-seq_len, seq_len):
    T5RelativeEmbedding._relative_position_bucket, bidirectional (wan/modules/t5.py:236-256), computed with the
    reference's own float formula on the host so that bucket boundaries agree exactly."""
    rel = torch.arange(-(seq_len - 1), seq_len)
    nb = num_buckets // 2
    buckets = (rel > 0).long() * nb
    rp = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (buckets + torch.where(rp < max_exact, rp, large)).to(torch.int32)


class UMT5Encoder:
    """Device-resident encoder weights + the C model descriptor (counterpart of `T5Encoder`, t5.py:266-312)."""

    def __init__(self, shape: T5Shape, state_dict: Dict[str, Tensor], device):
        self.shape = shape
        self.device = torch.device(device)
        self._keep: List[Tensor] = []
        self._ws: Dict[tuple, Tensor] = {}
        self._buckets: Dict[int, Tensor] = {}
        need = t5_param_shapes(shape)
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"T5 state dict lacks {len(missing)} encoder tensors, e.g. {missing[:4]}")
        for k, shp in need.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(state_dict[k].shape)}")
        if shape.head_dim != 64 or shape.dim % 512 or shape.dim_ffn % 64:
            raise ValueError("supported encoder shapes: head_dim 64, dim a multiple of 512, dim_ffn a multiple of 64")
        sd = state_dict
        m = _lib.T5Model()
        m.vocab, m.dim, m.dim_attn, m.dim_ffn = shape.vocab_size, shape.dim, shape.dim_attn, shape.dim_ffn
        m.num_heads, m.num_layers, m.num_buckets, m.eps = shape.num_heads, shape.num_layers, shape.num_buckets, shape.eps
        m.token_embedding = self._dev(sd["token_embedding.weight"]).data_ptr()
        layers = (_lib.T5Layer * shape.num_layers)()
        for i in range(shape.num_layers):
            p, ly = f"blocks.{i}.", layers[i]
            ly.norm1_w = self._dev(sd[p + "norm1.weight"]).data_ptr()
            ly.qk_w = self._dev(torch.cat([sd[p + "attn.q.weight"], sd[p + "attn.k.weight"]], 0)).data_ptr()
            ly.v_w = self._dev(sd[p + "attn.v.weight"]).data_ptr()
            ly.o_w = self._dev(sd[p + "attn.o.weight"]).data_ptr()
            ly.norm2_w = self._dev(sd[p + "norm2.weight"]).data_ptr()
            ly.gate_w = self._dev(sd[p + "ffn.gate.0.weight"]).data_ptr()
            ly.fc1_w = self._dev(sd[p + "ffn.fc1.weight"]).data_ptr()
            ly.fc2_w = self._dev(sd[p + "ffn.fc2.weight"]).data_ptr()
            ly.pos_emb = self._dev(sd[p + "pos_embedding.embedding.weight"]).data_ptr()
        self._layers = layers
        m.layers_host = C.cast(layers, C.POINTER(_lib.T5Layer))
        m.final_norm_w = self._dev(sd["norm.weight"]).data_ptr()
        self.cmodel = m
        self._handle = torch_ops.register_model(self)

    def _dev(self, t: Tensor) -> Tensor:
        t = t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
        self._keep.append(t)
        return t

    def param_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)

    def __call__(self, ids: Tensor, mask: Tensor) -> Tensor:
        """`T5Encoder.forward(ids, mask)` + the zero padding of `WanTextEncoder.forward`: ids, mask [B, L] ->
        bf16 [B, L, dim]."""
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError(f"expected ids and mask of one shape [B, L], got {tuple(ids.shape)} and {tuple(mask.shape)}")
        if int(ids.min()) < 0 or int(ids.max()) >= self.shape.vocab_size:
            raise ValueError(f"token id outside [0, {self.shape.vocab_size})")
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = mask.to(device=self.device, dtype=torch.int64).contiguous()
        B, L = ids.shape
        if L not in self._buckets:
            self._buckets[L] = relative_position_buckets(L, self.shape.num_buckets, self.shape.max_dist).to(self.device)
        key = (B, L, torch.cuda.current_stream(self.device).cuda_stream)
        if key not in self._ws:
            n = _lib.lib().sf_t5_workspace_bytes(C.byref(self.cmodel), B, L)
            if n == 0:
                _lib.check(-1, "sf_t5_workspace_bytes")
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        ws = self._ws[key]
        return torch.ops.sf_hip.t5_encode(self._handle, ids, mask, self._buckets[L], ws)


# default locations of the reference (utils/wan_wrapper.py:26-35)
T5_CHECKPOINTS = ("/tmp/models_t5_umt5-xxl-enc-bf16.pth", "wan_models/Wan2.1-T2V-1.3B/models_t5_umt5-xxl-enc-bf16.pth")
T5_TOKENIZER_DIR = "wan_models/Wan2.1-T2V-1.3B/google/umt5-xxl/"


def load_t5_checkpoint(path: Optional[str] = None) -> Dict[str, Tensor]:
    """The encoder tensors of `models_t5_umt5-xxl-enc-bf16.pth` from the reference's default locations.  Loaded with
    `weights_only=True` (the reference itself unpickles with weights_only=False, wan_wrapper.py:30-32)."""
    import os
    for cand in ([path] if path else T5_CHECKPOINTS):
        if os.path.exists(cand):
            return torch.load(cand, map_location="cpu", weights_only=True)
    raise FileNotFoundError(
        f"umT5 checkpoint not found (looked at {[path] if path else list(T5_CHECKPOINTS)}): download it as the reference's "
        "README describes, or construct WanTextEncoder(state_dict=...) / inject a text_encoder= into the pipeline")


class HuggingfaceTokenizer:
    """`HuggingfaceTokenizer(name, seq_len=512, clean='whitespace')` of wan/modules/tokenizers.py:38-73 for a LOCAL
    tokenizer directory: html-unescape twice, strip, collapse whitespace, then the sentencepiece tokenizer with
    padding / truncation to `seq_len`.  (`ftfy.fix_text`, the first cleaning step of the reference, is not installed
    here and is skipped: it only repairs mojibake.)"""

    def __init__(self, name: str = T5_TOKENIZER_DIR, seq_len: int = 512):
        import os
        if not os.path.isdir(name):
            raise FileNotFoundError(f"tokenizer directory {name!r} not found; pass tokenizer=callable(texts) -> (ids, mask)")
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(name, local_files_only=True)
        self.seq_len = seq_len

    @staticmethod
    def clean(text: str) -> str:
        import html
        import re
        return re.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip()

    def __call__(self, texts):
        if isinstance(texts, str):
            texts = [texts]
        out = self.tokenizer([self.clean(t) for t in texts], return_tensors="pt", padding="max_length", truncation=True,
                             max_length=self.seq_len, add_special_tokens=True)
        return out.input_ids, out.attention_mask


class WanTextEncoder(torch.nn.Module):
    """Drop-in for the reference's `WanTextEncoder` (utils/wan_wrapper.py:15-55).  `WanTextEncoder()` -- no arguments,
    as the reference's pipelines construct it -- loads the checkpoint and the tokenizer from the reference's default
    local paths (weights-only) and raises FileNotFoundError when they are absent; there is no download."""

    def __init__(self, state_dict: Optional[Dict[str, Tensor]] = None, tokenizer: Optional[Callable] = None, device="cuda",
                 shape: T5Shape = UMT5_XXL, checkpoint_path: Optional[str] = None):
        super().__init__()
        if state_dict is None:
            state_dict = load_t5_checkpoint(checkpoint_path)
            if tokenizer is None:
                tokenizer = HuggingfaceTokenizer()
        self.text_encoder = UMT5Encoder(shape, state_dict, device)
        self.tokenizer = tokenizer

    @property
    def device(self):
        return self.text_encoder.device

    def encode_ids(self, ids: Tensor, mask: Tensor) -> dict:
        return {"prompt_embeds": self.text_encoder(ids, mask)}

    def forward(self, text_prompts: List[str]) -> dict:
        if self.tokenizer is None:
            raise NotImplementedError(
                "no tokenizer: the sentencepiece model of google/umt5-xxl is not shipped with this package; pass "
                "tokenizer=callable(texts) -> (ids, mask) or use encode_ids(ids, mask)")
        ids, mask = self.tokenizer(text_prompts)
        return self.encode_ids(ids, mask)
