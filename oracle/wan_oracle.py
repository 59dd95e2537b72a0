"""CPU oracle for the Self-Forcing chunk-wise autoregressive denoising rollout.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch CPU restatement (torch CPU
tensors as the array library) of the reference algorithm for ONE path:
`CausalInferencePipeline.inference` -> `WanDiffusionWrapper.forward` ->
`CausalWanModel._forward_inference`.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it -- never the product package.

Parity status: PINNED.  `oracle/make_golden.py` imports the reference itself (on CPU,
in the build container, with the import shims of SURVEY.md Appendix A.5), runs it on
seeded inputs and stores its outputs under `tests/golden/`; `tests/test_oracle_golden.py`
checks every function below against those vectors (fp32 mode <= 1e-5 relative, bf16
mode within the reference's own bf16 noise).  The reference's own tests hold no golden
vectors for this path, so these generated fixtures are the pin.

All citations are `path:line` relative to the reference checkout.

Two numeric modes, selected by the dtype of the prepared weights:
  * torch.float32 -- the "math oracle": bf16-rounded weights, every op in fp32
    (the fp64 islands of the reference are kept in fp64).
  * torch.bfloat16 -- mimics the reference's rounding points (each torch op on bf16
    tensors rounds its result to bf16, exactly as the reference's op sequence does).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Shape parameters of CausalWanModel (wan/modules/causal_model.py:381-398)."""
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
    # The reference hard-codes 32760 / local_attn_size*1560 (causal_model.py:77).  When
    # None we reproduce that literally; tests at other latent sizes may set it.
    max_attention_size: Optional[int] = None

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads

    def attn_window(self) -> int:
        if self.max_attention_size is not None:
            return self.max_attention_size
        return 32760 if self.local_attn_size == -1 else self.local_attn_size * 1560


# --------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: Tensor) -> Tensor:
    """wan/modules/model.py:15-25.  float64 in, float64 out: [len(position), dim]."""
    assert dim % 2 == 0
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.pow(torch.tensor(10000.0, dtype=torch.float64),
                    -torch.arange(half, dtype=torch.float64) / half)
    ang = pos[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_angles(max_seq_len: int, dim: int, theta: float = 10000.0) -> Tensor:
    """Angle table of wan/modules/model.py:29-36 (the argument of torch.polar), fp64
    [max_seq_len, dim // 2]."""
    assert dim % 2 == 0
    inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                          torch.arange(0, dim, 2, dtype=torch.float64) / dim)
    return torch.arange(max_seq_len, dtype=torch.float64)[:, None] * inv[None, :]


def rope_tables(head_dim: int, max_pos: int = 1024) -> Tuple[Tensor, Tensor, Tuple[int, int, int]]:
    """cos/sin tables [max_pos, head_dim//2] (fp64) laid out as the reference's
    `self.freqs` (causal_model.py:481-488): columns [0,c0) time, [c0,c0+c1) height,
    [c0+c1, c) width with the split of causal_model.py:32."""
    d = head_dim
    ang = torch.cat([
        rope_angles(max_pos, d - 4 * (d // 6)),
        rope_angles(max_pos, 2 * (d // 6)),
        rope_angles(max_pos, 2 * (d // 6)),
    ], dim=1)
    c = d // 2
    split = (c - 2 * (c // 3), c // 3, c // 3)
    return ang.cos(), ang.sin(), split


def causal_rope_apply(x: Tensor, grid: Tuple[int, int, int], cos: Tensor, sin: Tensor,
                      split: Tuple[int, int, int], start_frame: int = 0) -> Tensor:
    """wan/modules/causal_model.py:28-56 restated with explicit cos/sin (no complex
    dtype).  x: [B, L, n, d] with L == f*h*w; pair j = channels (2j, 2j+1); the time
    index is offset by `start_frame`.  fp64 math, result cast to x.dtype."""
    f, h, w = grid
    B, L, n, d = x.shape
    assert L == f * h * w, "oracle handles un-padded sequences only"
    c0, c1, c2 = split
    fi = torch.arange(f).view(f, 1, 1).expand(f, h, w).reshape(-1) + start_frame
    hi = torch.arange(h).view(1, h, 1).expand(f, h, w).reshape(-1)
    wi = torch.arange(w).view(1, 1, w).expand(f, h, w).reshape(-1)
    cs = torch.cat([cos[fi, :c0], cos[hi, c0:c0 + c1], cos[wi, c0 + c1:]], dim=1)  # [L, d/2]
    sn = torch.cat([sin[fi, :c0], sin[hi, c0:c0 + c1], sin[wi, c0 + c1:]], dim=1)
    xd = x.to(torch.float64).reshape(B, L, n, d // 2, 2)
    re, im = xd[..., 0], xd[..., 1]
    cs = cs.view(1, L, 1, d // 2)
    sn = sn.view(1, L, 1, d // 2)
    out = torch.stack([re * cs - im * sn, re * sn + im * cs], dim=-1)
    return out.reshape(B, L, n, d).to(x.dtype)


def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """WanRMSNorm, wan/modules/model.py:70-86: fp32 normalise over the LAST dim (all
    heads jointly), cast back, then multiply by the weight in the activation dtype."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y.type_as(x) * weight


def layer_norm(x: Tensor, eps: float, weight: Optional[Tensor] = None,
               bias: Optional[Tensor] = None) -> Tensor:
    """WanLayerNorm, wan/modules/model.py:89-99 (nn.LayerNorm over the last dim)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps).type_as(x)


def sdpa(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(d)) v without mask (wan/modules/attention.py:156-202, the
    SDPA branch; flash-attn computes the same function).  q [B,Lq,n,d], k/v [B,Lk,n,d]
    -> [B,Lq,n,d].  Scores / softmax / PV accumulate in fp32; the result is rounded to
    q.dtype (the reference returns bf16 from the kernel)."""
    d = q.shape[-1]
    qf = q.float().transpose(1, 2)
    kf = k.float().transpose(1, 2)
    vf = v.float().transpose(1, 2)
    if q.shape[0] * q.shape[2] * q.shape[1] * k.shape[1] > (1 << 28):
        # big score matrices one head at a time (same arithmetic per head; bounds the memory of the full-size tests)
        o = torch.empty_like(qf)
        for b in range(qf.shape[0]):
            for h in range(qf.shape[1]):
                s = torch.matmul(qf[b, h], kf[b, h].transpose(-1, -2)) * (1.0 / math.sqrt(d))
                o[b, h] = torch.matmul(torch.softmax(s, dim=-1), vf[b, h])
        return o.transpose(1, 2).contiguous().to(q.dtype)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * (1.0 / math.sqrt(d))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vf)
    return o.transpose(1, 2).contiguous().to(q.dtype)


def gelu_tanh(x: Tensor) -> Tensor:
    return F.gelu(x, approximate="tanh")


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def prepare_weights(sd: Dict[str, Tensor], dtype: torch.dtype) -> Dict[str, Tensor]:
    """Round every tensor to bf16 (what `pipeline.to(bfloat16)` does, inference.py:73)
    and then present it in `dtype` (bf16 for the faithful mode, fp32 for the math
    oracle)."""
    return {k: v.detach().to(torch.bfloat16).to(dtype).contiguous() for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# caches (same dict schema as the reference)
# --------------------------------------------------------------------------------------
def init_kv_cache(cfg: OracleConfig, batch: int, cache_tokens: int, dtype) -> List[dict]:
    """pipeline/causal_inference.py:278-298 with the shape taken from cfg."""
    return [{
        "k": torch.zeros(batch, cache_tokens, cfg.num_heads, cfg.head_dim, dtype=dtype),
        "v": torch.zeros(batch, cache_tokens, cfg.num_heads, cfg.head_dim, dtype=dtype),
        "global_end_index": torch.tensor([0], dtype=torch.long),
        "local_end_index": torch.tensor([0], dtype=torch.long),
    } for _ in range(cfg.num_layers)]


def init_crossattn_cache(cfg: OracleConfig, batch: int, dtype) -> List[dict]:
    """pipeline/causal_inference.py:300-312."""
    return [{
        "k": torch.zeros(batch, cfg.text_len, cfg.num_heads, cfg.head_dim, dtype=dtype),
        "v": torch.zeros(batch, cfg.text_len, cfg.num_heads, cfg.head_dim, dtype=dtype),
        "is_init": False,
    } for _ in range(cfg.num_layers)]


def kv_cache_plan(local_end: int, global_end: int, current_start: int, n_new: int,
                  capacity: int, local_attn_size: int, sink_tokens: int, window: int):
    """Index arithmetic of wan/modules/causal_model.py:202-236 on host integers.

    Returns (evict, keep, new_local_end, write_start, attn_start).  `evict` > 0 means:
    move cache[sink+evict : sink+evict+keep] to cache[sink : sink+keep] first."""
    current_end = current_start + n_new
    evict = keep = 0
    if local_attn_size != -1 and current_end > global_end and n_new + local_end > capacity:
        evict = n_new + local_end - capacity
        keep = local_end - evict - sink_tokens
        new_local_end = local_end + current_end - global_end - evict
    else:
        new_local_end = local_end + current_end - global_end
    write_start = new_local_end - n_new
    attn_start = max(0, new_local_end - window)
    return evict, keep, new_local_end, write_start, attn_start


# --------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------
def self_attention(W: Dict[str, Tensor], pre: str, cfg: OracleConfig, x: Tensor,
                   grid: Tuple[int, int, int], rope, kv: dict, current_start: int) -> Tensor:
    """CausalWanSelfAttention.forward, kv-cache branch
    (wan/modules/causal_model.py:106-118, 194-241)."""
    B, s, _ = x.shape
    n, d = cfg.num_heads, cfg.head_dim
    q = rms_norm(F.linear(x, W[pre + "q.weight"], W[pre + "q.bias"]), W[pre + "norm_q.weight"], cfg.eps)
    k = rms_norm(F.linear(x, W[pre + "k.weight"], W[pre + "k.bias"]), W[pre + "norm_k.weight"], cfg.eps)
    v = F.linear(x, W[pre + "v.weight"], W[pre + "v.bias"])
    q, k, v = (u.view(B, s, n, d) for u in (q, k, v))

    frame_seqlen = grid[1] * grid[2]
    start_frame = current_start // frame_seqlen
    cos, sin, split = rope
    q = causal_rope_apply(q, grid, cos, sin, split, start_frame).type_as(v)
    k = causal_rope_apply(k, grid, cos, sin, split, start_frame).type_as(v)

    cap = kv["k"].shape[1]
    evict, keep, local_end, wstart, astart = kv_cache_plan(
        int(kv["local_end_index"].item()), int(kv["global_end_index"].item()),
        current_start, s, cap, cfg.local_attn_size, cfg.sink_size * frame_seqlen,
        cfg.attn_window())
    if evict > 0:
        sink = cfg.sink_size * frame_seqlen
        for name in ("k", "v"):
            kv[name][:, sink:sink + keep] = kv[name][:, sink + evict:sink + evict + keep].clone()
    kv["k"][:, wstart:local_end] = k
    kv["v"][:, wstart:local_end] = v
    o = sdpa(q, kv["k"][:, astart:local_end], kv["v"][:, astart:local_end])
    kv["global_end_index"].fill_(current_start + s)
    kv["local_end_index"].fill_(local_end)
    return F.linear(o.flatten(2), W[pre + "o.weight"], W[pre + "o.bias"])


def cross_attention(W: Dict[str, Tensor], pre: str, cfg: OracleConfig, x: Tensor,
                    context: Tensor, cache: Optional[dict]) -> Tensor:
    """WanT2VCrossAttention.forward, wan/modules/model.py:159-194.  No key-length mask:
    all text_len positions take part in the softmax (context_lens is None,
    causal_model.py:836)."""
    B = x.shape[0]
    n, d = cfg.num_heads, cfg.head_dim
    q = rms_norm(F.linear(x, W[pre + "q.weight"], W[pre + "q.bias"]), W[pre + "norm_q.weight"], cfg.eps)
    q = q.view(B, -1, n, d)
    if cache is not None and cache["is_init"]:
        k, v = cache["k"], cache["v"]
    else:
        k = rms_norm(F.linear(context, W[pre + "k.weight"], W[pre + "k.bias"]),
                     W[pre + "norm_k.weight"], cfg.eps).view(B, -1, n, d)
        v = F.linear(context, W[pre + "v.weight"], W[pre + "v.bias"]).view(B, -1, n, d)
        if cache is not None:
            cache["is_init"] = True
            cache["k"], cache["v"] = k, v
    o = sdpa(q, k, v)
    return F.linear(o.flatten(2), W[pre + "o.weight"], W[pre + "o.bias"])


def _per_group(x: Tensor, groups: int) -> Tensor:
    """[B, L, C] -> [B, groups, L/groups, C]; the group count is the second dim of the
    timestep tensor, NOT the frame count of the grid (causal_model.py:307)."""
    return x.unflatten(1, (groups, x.shape[1] // groups))


def attention_block(W: Dict[str, Tensor], i: int, cfg: OracleConfig, x: Tensor, e0: Tensor,
                    grid, rope, context: Tensor, kv: dict, cross: Optional[dict],
                    current_start: int) -> Tensor:
    """CausalWanAttentionBlock.forward, wan/modules/causal_model.py:284-336.
    e0: [B, G, 6, C]."""
    pre = f"blocks.{i}."
    G = e0.shape[1]
    e = (W[pre + "modulation"].unsqueeze(1) + e0).chunk(6, dim=2)  # 6 x [B, G, 1, C]
    h = (_per_group(layer_norm(x, cfg.eps), G) * (1 + e[1]) + e[0]).flatten(1, 2)
    y = self_attention(W, pre + "self_attn.", cfg, h, grid, rope, kv, current_start)
    x = x + (_per_group(y, G) * e[2]).flatten(1, 2)
    x = x + cross_attention(W, pre + "cross_attn.", cfg,
                            layer_norm(x, cfg.eps, W[pre + "norm3.weight"], W[pre + "norm3.bias"]),
                            context, cross)
    h = (_per_group(layer_norm(x, cfg.eps), G) * (1 + e[4]) + e[3]).flatten(1, 2)
    y = F.linear(gelu_tanh(F.linear(h, W[pre + "ffn.0.weight"], W[pre + "ffn.0.bias"])),
                 W[pre + "ffn.2.weight"], W[pre + "ffn.2.bias"])
    x = x + (_per_group(y, G) * e[5]).flatten(1, 2)
    return x


def patch_embed(W: Dict[str, Tensor], cfg: OracleConfig, x: Tensor) -> Tuple[Tensor, Tuple[int, int, int]]:
    """Conv3d(in_dim, dim, k=s=patch) of causal_model.py:458-459, 775-781 as an explicit
    non-overlapping-patch gather followed by a matmul.  x: [B, C_in, F, H, W] ->
    tokens [B, f*h*w, dim] in (f, h, w) row-major order."""
    B, Cin, Fr, H, Wd = x.shape
    pt, ph, pw = cfg.patch_size
    f, h, w = Fr // pt, H // ph, Wd // pw
    cols = x.view(B, Cin, f, pt, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    cols = cols.reshape(B, f * h * w, Cin * pt * ph * pw)
    wt = W["patch_embedding.weight"].flatten(1)
    return F.linear(cols, wt, W["patch_embedding.bias"]), (f, h, w)


def head_unpatchify(W: Dict[str, Tensor], cfg: OracleConfig, x: Tensor, e: Tensor,
                    grid: Tuple[int, int, int]) -> Tensor:
    """CausalHead.forward (causal_model.py:356-367) + unpatchify (:1081-1104).
    x [B, L, C]; e [B, G, 1, C] -> [B, out_dim, F, H, W]."""
    B = x.shape[0]
    G = e.shape[1]
    m = (W["head.modulation"].unsqueeze(1) + e).chunk(2, dim=2)
    y = F.linear(_per_group(layer_norm(x, cfg.eps), G) * (1 + m[1]) + m[0],
                 W["head.head.weight"], W["head.head.bias"])  # [B, G, L/G, P*c]
    f, h, w = grid
    pt, ph, pw = cfg.patch_size
    c = cfg.out_dim
    y = y.reshape(B, f, h, w, pt, ph, pw, c)
    y = y.permute(0, 7, 1, 4, 2, 5, 3, 6)  # b c f pt h ph w pw
    return y.reshape(B, c, f * pt, h * ph, w * pw)


def time_embeddings(W: Dict[str, Tensor], cfg: OracleConfig, t: Tensor, dtype) -> Tuple[Tensor, Tensor]:
    """causal_model.py:829-832.  t: [B, G] (any real dtype) -> e [B*G, C], e0 [B, G, 6, C]."""
    s = sinusoidal_embedding_1d(cfg.freq_dim, t.flatten()).to(dtype)
    e = F.linear(F.silu(F.linear(s, W["time_embedding.0.weight"], W["time_embedding.0.bias"])),
                 W["time_embedding.2.weight"], W["time_embedding.2.bias"])
    e0 = F.linear(F.silu(e), W["time_projection.1.weight"], W["time_projection.1.bias"])
    return e, e0.unflatten(1, (6, cfg.dim)).unflatten(0, tuple(t.shape))


def text_embedding(W: Dict[str, Tensor], cfg: OracleConfig, context: Tensor) -> Tensor:
    """causal_model.py:837-842: zero-pad to text_len, Linear-GELU(tanh)-Linear."""
    B, L, D = context.shape
    if L < cfg.text_len:
        context = torch.cat([context, context.new_zeros(B, cfg.text_len - L, D)], dim=1)
    return F.linear(gelu_tanh(F.linear(context, W["text_embedding.0.weight"], W["text_embedding.0.bias"])),
                    W["text_embedding.2.weight"], W["text_embedding.2.bias"])


def forward_inference(W: Dict[str, Tensor], cfg: OracleConfig, x: Tensor, t: Tensor,
                      context: Tensor, kv_cache: List[dict], crossattn_cache: List[dict],
                      current_start: int, rope=None, add_condition: Optional[Tensor] = None) -> Tensor:
    """CausalWanModel._forward_inference, wan/modules/causal_model.py:725-893.
    x: [B, C_in, F, H, W]; t: [B, G]; context: [B, <=text_len, text_dim]
    -> flow [B, C_out, F, H, W]."""
    dtype = W["patch_embedding.weight"].dtype
    if rope is None:
        rope = rope_tables(cfg.head_dim)
    tok, grid = patch_embed(W, cfg, x.to(dtype))
    if add_condition is not None:
        # Pose tokens of the fork, causal_model.py:786-819: x += pose_proj(add_condition), lengths must match.
        # PARITY UNPINNED for this branch: in the reference snapshot the inference path itself raises at
        # causal_model.py:794 (`x.view(B, L, C)` with L computed from an already-batched [B, L, C] tensor), so
        # no reference output exists; this restates the evident intent, identical to the arithmetic of the
        # training branch (causal_model.py:980-994).
        if add_condition.shape[1] != tok.shape[1]:
            raise ValueError(f"add_condition spatial dim {add_condition.shape[1]} doesn't match x spatial dim {tok.shape[1]}")
        if "pose_proj.weight" in W:
            tok = tok + F.linear(add_condition.to(dtype), W["pose_proj.weight"], W["pose_proj.bias"])
        else:   # dim == 5120: `pose_proj = nn.Identity()` (causal_model.py:500-503) -- a plain add of the pose tokens
            if add_condition.shape[2] != tok.shape[2]:
                raise ValueError(f"no pose_proj weights: add_condition must be {tok.shape[2]} wide, got {add_condition.shape[2]}")
            tok = tok + add_condition.to(dtype)
    e, e0 = time_embeddings(W, cfg, t, dtype)
    ctx = text_embedding(W, cfg, context.to(dtype))
    h = tok
    for i in range(cfg.num_layers):
        h = attention_block(W, i, cfg, h, e0, grid, rope, ctx, kv_cache[i],
                            crossattn_cache[i], current_start)
    return head_unpatchify(W, cfg, h, e.unflatten(0, tuple(t.shape)).unsqueeze(2), grid)


# --------------------------------------------------------------------------------------
# scheduler + wrapper
# --------------------------------------------------------------------------------------
class FlowMatchTables:
    """FlowMatchScheduler(shift, sigma_min=0, extra_one_step=True).set_timesteps(1000)
    -- utils/scheduler.py:106-141 as used by utils/wan_wrapper.py:171-174."""

    def __init__(self, shift: float, num_train_timesteps: int = 1000, n: int = 1000):
        sig = torch.linspace(1.0, 0.0, n + 1)[:-1]
        self.sigmas = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = self.sigmas * num_train_timesteps

    def sigma_of(self, timestep: Tensor) -> Tensor:
        """nearest-entry lookup, utils/scheduler.py:172-174."""
        idx = torch.argmin((self.timesteps.unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
        return self.sigmas[idx]

    def add_noise(self, x0: Tensor, noise: Tensor, timestep: Tensor) -> Tensor:
        """utils/scheduler.py:159-176: (1-sigma) x0 + sigma eps in fp32, cast to noise's dtype."""
        sigma = self.sigma_of(timestep.flatten()).reshape(-1, 1, 1, 1)
        return ((1 - sigma) * x0 + sigma * noise).type_as(noise)

    def warp(self, steps: Sequence[int]) -> Tensor:
        """pipeline/causal_inference.py:27-31."""
        table = torch.cat((self.timesteps, torch.tensor([0], dtype=torch.float32)))
        return table[1000 - torch.tensor(list(steps), dtype=torch.long)]


def flow_to_x0(sched: FlowMatchTables, flow: Tensor, xt: Tensor, timestep: Tensor) -> Tensor:
    """utils/wan_wrapper.py:204-228: x0 = xt - sigma_t * flow in float64."""
    sig = sched.sigmas.double()
    ts = sched.timesteps.double()
    idx = torch.argmin((ts.unsqueeze(0) - timestep.double().unsqueeze(1)).abs(), dim=1)
    s = sig[idx].reshape(-1, 1, 1, 1)
    return (xt.double() - s * flow.double()).to(flow.dtype)


def wrapper_forward(W, cfg: OracleConfig, sched: FlowMatchTables, noisy: Tensor, prompt_embeds: Tensor,
                    timestep: Tensor, kv_cache, crossattn_cache, current_start: int, rope=None, add_condition=None):
    """WanDiffusionWrapper.forward, kv branch -- utils/wan_wrapper.py:253-300, 340-349.
    noisy [B, F, C, H, W]; timestep [B, G] -> (flow_pred, pred_x0), both [B, F, C, H, W]."""
    flow = forward_inference(W, cfg, noisy.permute(0, 2, 1, 3, 4), timestep, prompt_embeds,
                             kv_cache, crossattn_cache, current_start, rope, add_condition).permute(0, 2, 1, 3, 4)
    # timestep.flatten(0,1) has B*G entries; with G == F this is one sigma per frame.
    x0 = flow_to_x0(sched, flow.flatten(0, 1), noisy.flatten(0, 1).to(flow.dtype),
                    timestep.flatten(0, 1)).unflatten(0, flow.shape[:2])
    return flow, x0


# --------------------------------------------------------------------------------------
# rollout
# --------------------------------------------------------------------------------------
@dataclass
class RolloutArgs:
    denoising_step_list: Sequence[int] = (1000, 750, 500, 250)
    warp_denoising_step: bool = True
    num_frame_per_block: int = 3
    independent_first_frame: bool = False
    context_noise: int = 0
    timestep_shift: float = 5.0


def rollout(W, cfg: OracleConfig, args: RolloutArgs, noise: Tensor, prompt_embeds: Tensor,
            renoise: Sequence[Tensor], initial_latent: Optional[Tensor] = None,
            cache_tokens: Optional[int] = None, kv_cache=None, crossattn_cache=None) -> Tensor:
    """CausalInferencePipeline.inference up to the latents -- pipeline/causal_inference.py:47-244.

    `renoise` supplies the epsilon tensors the reference draws with torch.randn_like
    (:208), one per (chunk, non-final step) in call order, each [B*f, C, H, W].
    Returns the latents `output` [B, F_total, C, H, W] in noise.dtype."""
    B, Fn, C, H, Wd = noise.shape
    nf = args.num_frame_per_block
    sched = FlowMatchTables(args.timestep_shift)
    steps = torch.tensor(list(args.denoising_step_list), dtype=torch.long)
    if args.warp_denoising_step:
        steps = sched.warp(args.denoising_step_list)
    if not args.independent_first_frame or initial_latent is not None:
        assert Fn % nf == 0
        num_blocks = Fn // nf
    else:
        assert (Fn - 1) % nf == 0
        num_blocks = (Fn - 1) // nf
    n_in = initial_latent.shape[1] if initial_latent is not None else 0
    frame_seqlen = (H // cfg.patch_size[1]) * (Wd // cfg.patch_size[2])
    dtype = W["patch_embedding.weight"].dtype
    if kv_cache is None:
        if cache_tokens is None:
            cache_tokens = (cfg.local_attn_size if cfg.local_attn_size != -1 else Fn + n_in) * frame_seqlen
        kv_cache = init_kv_cache(cfg, B, cache_tokens, dtype)
        crossattn_cache = init_crossattn_cache(cfg, B, dtype)
    rope = rope_tables(cfg.head_dim)
    out = torch.zeros(B, Fn + n_in, C, H, Wd, dtype=noise.dtype)

    def gen(xin, ts, start_frame):
        return wrapper_forward(W, cfg, sched, xin.to(dtype), prompt_embeds, ts, kv_cache,
                               crossattn_cache, start_frame * frame_seqlen, rope)

    cur = 0
    if initial_latent is not None:  # :136-169 -- context warm-up at t = 0, ONE modulation group
        t0 = torch.zeros(B, 1, dtype=torch.int64)
        if args.independent_first_frame:
            assert (n_in - 1) % nf == 0
            n_in_blocks = (n_in - 1) // nf
            out[:, :1] = initial_latent[:, :1]
            gen(initial_latent[:, :1], t0, cur)
            cur += 1
        else:
            assert n_in % nf == 0
            n_in_blocks = n_in // nf
        for _ in range(n_in_blocks):
            ref = initial_latent[:, cur:cur + nf]
            out[:, cur:cur + nf] = ref
            gen(ref, t0, cur)
            cur += nf

    chunks = [nf] * num_blocks
    if args.independent_first_frame and initial_latent is None:
        chunks = [1] + chunks
    it = iter(renoise)
    for f in chunks:
        x = noise[:, cur - n_in:cur + f - n_in]
        for si in range(len(steps)):
            ts = torch.ones(B, f, dtype=torch.int64) * steps[si]
            _, x0 = gen(x, ts, cur)
            if si < len(steps) - 1:
                eps = next(it)
                nt = steps[si + 1] * torch.ones(B * f, dtype=torch.long)
                x = sched.add_noise(x0.flatten(0, 1), eps.to(x0.dtype), nt).unflatten(0, x0.shape[:2])
        out[:, cur:cur + f] = x0
        gen(x0, torch.ones(B, f, dtype=torch.int64) * args.context_noise, cur)  # :227-235
        cur += f
    return out
