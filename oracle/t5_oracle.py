"""CPU oracle for the umT5 text encoder (prompt token ids -> prompt embeddings), SURVEY.md section 8f row 3.

TEST INFRASTRUCTURE ONLY -- only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, never the product package.

A from-scratch CPU restatement (torch CPU tensors as the array library) of `WanTextEncoder.forward`
after tokenisation (utils/wan_wrapper.py:40-55) -> `T5Encoder.forward` (wan/modules/t5.py:299-312) ->
`T5SelfAttention` / `T5Attention` / `T5FeedForward` / `T5LayerNorm` / `T5RelativeEmbedding`
(t5.py:53-264).  The tokenizer (sentencepiece model of google/umt5-xxl) is not part of this path: the
boundary is (ids, mask).

Parity status: PINNED.  `oracle/make_golden_t5.py` imports the reference's `T5Encoder` on CPU, loads the
seeded weights of `self_forcing_amd.t5_weights.synth_t5_state_dict`, and stores its outputs under
`tests/golden/t5_*.npz`; `tests/test_t5_oracle_golden.py` checks this file against them.

Numeric modes as in `wan_oracle.py` (weights float32 = math oracle; bfloat16 = the reference's rounding points).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class T5OracleConfig:
    """umt5_xxl encoder arguments (t5.py:456-469)."""
    dim: int = 4096
    dim_attn: int = 4096
    dim_ffn: int = 10240
    num_heads: int = 64
    num_layers: int = 24
    num_buckets: int = 32
    max_dist: int = 128
    eps: float = 1e-6


def prepare_weights(sd: Dict[str, Tensor], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: v.detach().to("cpu").to(dtype) for k, v in sd.items()}


def t5_layer_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """T5LayerNorm (t5.py:53-66): RMS norm in fp32, cast to the weight dtype, then times weight."""
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.type_as(w)
    return w * y


def relative_position_bucket(rel_pos: Tensor, num_buckets: int, max_dist: int) -> Tensor:
    """Bidirectional bucketing (t5.py:236-256): half the buckets for each sign; exact below
    num_buckets/4, logarithmic up to max_dist, clamped."""
    nb = num_buckets // 2
    buckets = (rel_pos > 0).long() * nb
    rp = rel_pos.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(rp < max_exact, rp, large)


def position_bias(emb: Tensor, lq: int, lk: int, num_buckets: int, max_dist: int) -> Tensor:
    """T5RelativeEmbedding.forward (t5.py:222-234): [1, heads, lq, lk]; rel_pos = key index - query index."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    b = relative_position_bucket(rel, num_buckets, max_dist)
    return F.embedding(b, emb).permute(2, 0, 1).unsqueeze(0).contiguous()


def t5_attention(cfg: T5OracleConfig, W: Dict[str, Tensor], p: str, x: Tensor, mask: Optional[Tensor], bias: Tensor) -> Tensor:
    """T5Attention.forward (t5.py:88-122): no 1/sqrt(d) scaling; bias + key mask (finfo.min) added to the
    logits; softmax in fp32."""
    b, n, c = x.size(0), cfg.num_heads, cfg.dim_attn // cfg.num_heads
    q = F.linear(x, W[p + "q.weight"]).view(b, -1, n, c)
    k = F.linear(x, W[p + "k.weight"]).view(b, -1, n, c)
    v = F.linear(x, W[p + "v.weight"]).view(b, -1, n, c)
    attn_bias = x.new_zeros(b, n, q.size(1), k.size(1))
    attn_bias = attn_bias + bias.to(x.dtype)
    if mask is not None:
        attn_bias = attn_bias.masked_fill(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + attn_bias
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    y = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, -1, n * c)
    return F.linear(y, W[p + "o.weight"])


def gelu_tanh(x: Tensor) -> Tensor:
    """The reference's own GELU module (t5.py:46-50)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_ffn(W: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """T5FeedForward.forward (t5.py:137-142): fc2(fc1(x) * gelu(gate(x)))."""
    return F.linear(F.linear(x, W[p + "fc1.weight"]) * gelu_tanh(F.linear(x, W[p + "gate.0.weight"])), W[p + "fc2.weight"])


def t5_encode(cfg: T5OracleConfig, W: Dict[str, Tensor], ids: Tensor, mask: Optional[Tensor]) -> Tensor:
    """T5Encoder.forward (t5.py:299-312) with per-block position embeddings (shared_pos=False, t5.py:171-176).
    ids [B, L] int64, mask [B, L] (1 = token) -> [B, L, dim]."""
    x = F.embedding(ids, W["token_embedding.weight"])
    L = x.size(1)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        e = position_bias(W[p + "pos_embedding.embedding.weight"], L, L, cfg.num_buckets, cfg.max_dist)
        x = x + t5_attention(cfg, W, p + "attn.", t5_layer_norm(x, W[p + "norm1.weight"], cfg.eps), mask, e)
        x = x + t5_ffn(W, p + "ffn.", t5_layer_norm(x, W[p + "norm2.weight"], cfg.eps))
    return t5_layer_norm(x, W["norm.weight"], cfg.eps)


def text_encoder_forward(cfg: T5OracleConfig, W: Dict[str, Tensor], ids: Tensor, mask: Tensor) -> Tensor:
    """WanTextEncoder.forward after the tokenizer (utils/wan_wrapper.py:44-55): rows past each prompt's
    length are set to zero."""
    ctx = t5_encode(cfg, W, ids, mask).clone()
    for u, n in zip(ctx, mask.gt(0).sum(dim=1).long()):
        u[n:] = 0.0
    return ctx
