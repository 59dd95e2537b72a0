"""Device-resident causal Wan DiT: weights in HBM + the C-ABI descriptor of one forward.

Counterpart of `CausalWanModel` (wan/modules/causal_model.py:370-513) for the KV-cached
inference branch only (`_forward_inference`, :725-893).  It owns the bf16 weights, laid out for
the kernels (q|k|v and cross k|v projection matrices stacked so that one GEMM serves three / two
Linears), the fp32 RoPE tables and a per-shape workspace; `forward` is ONE C call
(`sf_dit_forward`) that enqueues every kernel of the pass on the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib, torch_ops
from .kvcache import CachePlan
from .weights import WanShape, param_shapes

Tensor = torch.Tensor


def rope_tables(head_dim: int, max_pos: int = 1024, theta: float = 10000.0):
    """cos/sin tables [max_pos, head_dim/2] in the column layout of the reference's `freqs`
    (causal_model.py:481-488; model.py:29-36): [time | height | width] with widths
    (d/2 - 2*(d/6), d/6, d/6)... computed in float64, stored float32."""
    d = head_dim
    parts = []
    for dim in (d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6)):
        inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                              torch.arange(0, dim, 2, dtype=torch.float64) / dim)
        parts.append(torch.arange(max_pos, dtype=torch.float64)[:, None] * inv[None, :])
    ang = torch.cat(parts, dim=1)
    return ang.cos().float().contiguous(), ang.sin().float().contiguous()


class CausalWanModel:
    """Inference-only causal DiT on one GPU.  Attributes mirror what the reference pipeline reads
    from `generator.model`: `local_attn_size`, `sink_size`, `num_frame_per_block`, plus the shape."""

    def __init__(self, shape: WanShape, state_dict: Dict[str, Tensor], device, sched_sigmas: Tensor,
                 sched_timesteps: Tensor):
        if shape.head_dim != 128:
            raise ValueError(f"head_dim must be 128 (dim={shape.dim}, heads={shape.num_heads})")
        if tuple(shape.patch_size) != (1, 2, 2):
            raise ValueError("only patch_size (1, 2, 2) is supported")
        self.shape = shape
        self.device = torch.device(device)
        self.dim, self.num_heads, self.num_layers = shape.dim, shape.num_heads, shape.num_layers
        self.local_attn_size = shape.local_attn_size
        self.sink_size = shape.sink_size
        self.num_frame_per_block = 1
        self.independent_first_frame = False
        self._keep: List[Tensor] = []      # every device tensor the C struct points into
        self._workspaces: Dict[tuple, Tensor] = {}
        self._load(state_dict, sched_sigmas, sched_timesteps)
        self._handle = torch_ops.register_model(self)

    # ---------------------------------------------------------------------------------
    def _dev(self, t: Tensor) -> Tensor:
        t = t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
        self._keep.append(t)
        return t

    def _load(self, sd: Dict[str, Tensor], sigmas: Tensor, timesteps: Tensor) -> None:
        need = param_shapes(self.shape)
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:4]}")
        for k, shp in need.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
        s = self.shape
        m = _lib.Model()
        m.dim, m.ffn_dim, m.num_heads, m.num_layers = s.dim, s.ffn_dim, s.num_heads, s.num_layers
        m.in_dim, m.out_dim, m.freq_dim, m.text_dim, m.text_len = s.in_dim, s.out_dim, s.freq_dim, s.text_dim, s.text_len
        m.eps = s.eps
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        m.patch_w = P(self._dev(sd["patch_embedding.weight"].flatten(1)))
        m.patch_b = P(self._dev(sd["patch_embedding.bias"]))
        for dst, src in (("text0", "text_embedding.0"), ("text2", "text_embedding.2"), ("time0", "time_embedding.0"),
                         ("time2", "time_embedding.2"), ("tproj", "time_projection.1"), ("head", "head.head")):
            setattr(m, dst + "_w", P(self._dev(sd[src + ".weight"])))
            setattr(m, dst + "_b", P(self._dev(sd[src + ".bias"])))
        m.head_mod = P(self._dev(sd["head.modulation"].reshape(2, s.dim)))
        from .weights import POSE_DIM
        self.has_pose_proj = "pose_proj.weight" in sd
        if self.has_pose_proj:   # optional: the fork's pose conditioning (Linear(5120, dim), causal_model.py:493-503)
            m.pose_w, m.pose_b = P(self._dev(sd["pose_proj.weight"])), P(self._dev(sd["pose_proj.bias"]))
            m.pose_dim = sd["pose_proj.weight"].shape[1]
        elif s.dim == POSE_DIM:  # `pose_proj = nn.Identity()` for dim-5120 models (:500-501): no weights, x += add_condition
            m.pose_dim = s.dim
        self.accepts_pose = self.has_pose_proj or s.dim == POSE_DIM
        layers = (_lib.LayerWeights * s.num_layers)()
        for i in range(s.num_layers):
            p = f"blocks.{i}."
            lw = layers[i]
            lw.modulation = P(self._dev(sd[p + "modulation"].reshape(6, s.dim)))
            lw.norm3_w, lw.norm3_b = P(self._dev(sd[p + "norm3.weight"])), P(self._dev(sd[p + "norm3.bias"]))
            sa, ca = p + "self_attn.", p + "cross_attn."
            lw.qkv_w = P(self._dev(torch.cat([sd[sa + "q.weight"], sd[sa + "k.weight"], sd[sa + "v.weight"]], 0)))
            lw.qkv_b = P(self._dev(torch.cat([sd[sa + "q.bias"], sd[sa + "k.bias"], sd[sa + "v.bias"]], 0)))
            lw.norm_q_w, lw.norm_k_w = P(self._dev(sd[sa + "norm_q.weight"])), P(self._dev(sd[sa + "norm_k.weight"]))
            lw.o_w, lw.o_b = P(self._dev(sd[sa + "o.weight"])), P(self._dev(sd[sa + "o.bias"]))
            lw.cq_w, lw.cq_b = P(self._dev(sd[ca + "q.weight"])), P(self._dev(sd[ca + "q.bias"]))
            lw.ckv_w = P(self._dev(torch.cat([sd[ca + "k.weight"], sd[ca + "v.weight"]], 0)))
            lw.ckv_b = P(self._dev(torch.cat([sd[ca + "k.bias"], sd[ca + "v.bias"]], 0)))
            lw.cnorm_q_w, lw.cnorm_k_w = P(self._dev(sd[ca + "norm_q.weight"])), P(self._dev(sd[ca + "norm_k.weight"]))
            lw.co_w, lw.co_b = P(self._dev(sd[ca + "o.weight"])), P(self._dev(sd[ca + "o.bias"]))
            lw.ffn0_w, lw.ffn0_b = P(self._dev(sd[p + "ffn.0.weight"])), P(self._dev(sd[p + "ffn.0.bias"]))
            lw.ffn2_w, lw.ffn2_b = P(self._dev(sd[p + "ffn.2.weight"])), P(self._dev(sd[p + "ffn.2.bias"]))
        self._layers = layers
        m.layers_host = C.cast(layers, C.POINTER(_lib.LayerWeights))
        cos, sin = rope_tables(s.head_dim)
        self.rope_cos = cos.to(self.device)
        self.rope_sin = sin.to(self.device)
        self.sched_sigmas = sigmas.to(device=self.device, dtype=torch.float32).contiguous()
        self.sched_timesteps = timesteps.to(device=self.device, dtype=torch.float32).contiguous()
        m.rope_cos, m.rope_sin = self.rope_cos.data_ptr(), self.rope_sin.data_ptr()
        m.sched_sigmas, m.sched_timesteps = self.sched_sigmas.data_ptr(), self.sched_timesteps.data_ptr()
        m.n_table = self.sched_sigmas.numel()
        self.cmodel = m

    def param_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)

    # ---------------------------------------------------------------------------------
    def workspace(self, B: int, F: int, H: int, W: int, G: int) -> Tensor:
        # one workspace per shape AND stream: concurrent rollouts on different HIP streams share the
        # weights but must not share activations
        key = (B, F, H, W, G, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._workspaces.get(key)
        if ws is None:
            n = _lib.lib().sf_dit_workspace_bytes(C.byref(self.cmodel), B, F, H, W, G)
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._workspaces[key] = ws
        return ws

    def forward(self, noisy: Tensor, timestep: Tensor, prompt_embeds: Optional[Tensor], init_cross: bool,
                k_cache: List[Tensor], v_cache: List[Tensor], ck_cache: List[Tensor], cv_cache: List[Tensor], plan: CachePlan,
                start_frame: int, evict_scratch: Optional[Tensor] = None, cache_only: bool = False,
                add_condition: Optional[Tensor] = None, kv_index: Optional[Tensor] = None):
        """noisy [B,F,in_dim,H,W] bf16 (contiguous); timestep [B,G] float32|int64 on device; *_cache: per-layer cache
        tensors (mutated in place); kv_index: the shared int64 [L, 2] buffer behind the cache dicts' index tensors (the
        pass ends by setting every row to (plan.global_end, plan.local_end)), or None.  Returns (flow, x0)
        [B,F,out_dim,H,W], or (None, None) with cache_only.
        ONE custom-op call: torch.ops.sf_hip.dit_forward -> sf_dit_forward."""
        B, F, Cin, H, W = noisy.shape
        ws = self.workspace(B, F, H, W, timestep.shape[1])
        flow, x0 = torch.ops.sf_hip.dit_forward(
            self._handle, noisy, timestep, prompt_embeds, add_condition, k_cache, v_cache, ck_cache, cv_cache, ws, evict_scratch,
            bool(init_cross), bool(cache_only), plan.sink, plan.evict, plan.keep, plan.write_start, plan.attn_start, plan.local_end,
            start_frame, kv_index, plan.global_end)
        return (None, None) if cache_only else (flow, x0)

    def forward_pair(self, ctx_noisy: Tensor, ctx_timestep: Tensor, noisy: Tensor, timestep: Tensor, k_cache: List[Tensor],
                     v_cache: List[Tensor], ck_cache: List[Tensor], cv_cache: List[Tensor], ctx_plan: CachePlan, plan: CachePlan,
                     ctx_start_frame: int, start_frame: int, evict_scratch: Optional[Tensor] = None, kv_index: Optional[Tensor] = None):
        """The context pass of chunk k (cache only) + the first denoising pass of chunk k + 1 as ONE call
        (torch.ops.sf_hip.dit_forward_pair -> sf_dit_forward_pair): bit-identical to two `forward` calls, but every
        row-wise kernel and GEMM sees both passes' rows at once.  Returns (flow, x0) of the denoising pass."""
        B, F, Cin, H, W = noisy.shape
        ws = self.workspace(2 * B, F, H, W, timestep.shape[1])
        as_list = lambda pl, sf: [pl.sink, pl.evict, pl.keep, pl.write_start, pl.attn_start, pl.local_end, sf]  # noqa: E731
        return torch.ops.sf_hip.dit_forward_pair(self._handle, ctx_noisy, ctx_timestep, noisy, timestep, k_cache, v_cache, ck_cache, cv_cache,
                                                 ws, evict_scratch, as_list(ctx_plan, ctx_start_frame), as_list(plan, start_frame), kv_index,
                                                 plan.global_end)
