"""`WanDiffusionWrapper` -- drop-in for the reference's generator wrapper on the KV-cached path.

Mirrors utils/wan_wrapper.py:120-177, 253-349 of the reference: same constructor keywords, same
`forward(noisy_image_or_video, conditional_dict, timestep, kv_cache, crossattn_cache,
current_start, ...) -> (flow_pred, pred_x0)`, same cache-dict schema (the callee mutates the
caller's caches in place), `.scheduler`, `.get_scheduler()`, `.model.local_attn_size`,
`.model.num_frame_per_block`.  Everything behind `forward` runs in the HIP kernels.

Differences from the reference, all deliberate:
  * weights: a local checkpoint directory (`model_path`) with `*.safetensors`, an explicit
    `state_dict=`, or `random_init_seed=`; there is no implicit download and no silent random init;
  * LoRA (`lora_rank`, `lora_alpha`, `lora_targets`, `lora_path`: utils/wan_wrapper.py:146-165): adapters found in the
    state dict (`<linear>.base.weight` + `lora_A/lora_B`, the layout `apply_lora` leaves behind) or loaded from
    `lora_path` (.safetensors / weights-only .pt; lora_A/lora_B or lora_up/lora_down pairs, the reference's prefix
    handling) are MERGED into the base matrices at load time, W += alpha/rank * B @ A -- the same function up to one
    bf16 rounding of the merged matrix, without the 14 % of extra run-time FLOPs; `lora_dropout` is inference-inert;
  * the attention window for `local_attn_size == -1` is the cache capacity and for rolling mode
    `local_attn_size * frame_seqlen` of the CURRENT latent size (the reference hard-codes
    32760 / `local_attn_size * 1560`, causal_model.py:77, which is only right for 60x104 latents);
  * the non-cached branches (`kv_cache is None`, classify_mode, clean_x teacher forcing) and the i2v inputs
    (`clip_feature`, `y`) raise NotImplementedError; the fork's pose tokens (`add_condition`) are supported.
"""
from __future__ import annotations

import glob
import os
import time
from typing import Dict, List, Optional

import torch

from .kvcache import plan_cache_update
from .model import CausalWanModel
from .scheduler import FlowMatchScheduler
from .weights import (LORA_DEFAULT_TARGETS, NAMED_SHAPES, WanShape, apply_lora_file, load_lora_file, merge_lora, strip_prefix,
                      synth_state_dict)

Tensor = torch.Tensor


def _load_checkpoint_dir(path: str) -> Dict[str, Tensor]:
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(
            f"no *.safetensors under {path!r}. Pass state_dict=..., or random_init_seed=<int> for "
            "seeded random weights (benchmarks / tests).")
    from safetensors.torch import load_file
    sd: Dict[str, Tensor] = {}
    for f in files:
        sd.update(load_file(f))
    return sd


class WanDiffusionWrapper(torch.nn.Module):
    def __init__(self, model_name: str = "Wan2.1-T2V-1.3B", model_path: Optional[str] = None, timestep_shift: float = 8.0,
                 is_causal: bool = False, local_attn_size: int = -1, sink_size: int = 0, lora_rank: Optional[int] = None,
                 lora_alpha: float = 1.0, lora_dropout: float = 0.0, lora_targets: Optional[List[str]] = None,
                 lora_path: Optional[str] = None, *, shape: Optional[WanShape] = None,
                 state_dict: Optional[Dict[str, Tensor]] = None, random_init_seed: Optional[int] = None,
                 device="cuda"):
        super().__init__()
        if not is_causal:
            raise NotImplementedError("only the causal (KV-cached) generator is implemented on this path")
        if shape is None:
            if model_name not in NAMED_SHAPES:
                raise ValueError(f"unknown model_name {model_name!r}; pass shape=WanShape(...)")
            shape = NAMED_SHAPES[model_name]
        shape = shape.replace(local_attn_size=local_attn_size, sink_size=sink_size)
        if state_dict is None:
            if random_init_seed is not None:
                state_dict = synth_state_dict(shape, seed=random_init_seed)
            else:
                state_dict = _load_checkpoint_dir(model_path or f"wan_models/{model_name}/")
        state_dict = strip_prefix(state_dict)
        if any(".lora_A." in k for k in state_dict):
            if not lora_rank:
                raise ValueError("state dict holds LoRA adapters but lora_rank was not given")
            state_dict = merge_lora(state_dict, alpha=lora_alpha, rank=lora_rank)
        self.lora_loaded = self.lora_skipped = 0
        if lora_rank is not None and lora_rank > 0 and lora_path is not None:
            # utils/wan_wrapper.py:146-165: adapters on `lora_targets` (default q, k, v, o of both attentions), weights
            # from `lora_path`; folded into the base matrices here (W += alpha / rank * B @ A) instead of evaluated as
            # base(x) + B(A(x)) * alpha / rank on every call (utils/lora.py:47-50)
            state_dict, self.lora_loaded, self.lora_skipped = apply_lora_file(
                state_dict, load_lora_file(lora_path), shape, lora_rank, lora_alpha, lora_targets or LORA_DEFAULT_TARGETS)
            print(f"Loaded LoRA weights: {self.lora_loaded} (skipped {self.lora_skipped}).")

        self.uniform_timestep = not is_causal
        self.scheduler = FlowMatchScheduler(shift=timestep_shift, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.model = CausalWanModel(shape, state_dict, device, self.scheduler.sigmas, self.scheduler.timesteps)
        self.seq_len = 32760
        self._evict_scratch: Optional[Tensor] = None
        self._init_throttle()

    # --- host-side pacing ---------------------------------------------------------------------
    # One forward is 25-50 ms of GPU work and ~2 ms of host work, so the calling thread runs far ahead of the GPU until the
    # stream's launch queue is full -- and then SPINS inside the HIP runtime for room (measured: the thread's CPU time inside
    # sf_dit_forward equals its wall time, one busy core per rollout thread).  With one process per GPU and two rollout
    # threads per process that is 16 spinning cores on an 8-GPU node for nothing.  Instead the wrapper keeps at most
    # `max_inflight_forwards` passes enqueued (per wrapper = per stream) and waits for the oldest one by POLLING its event
    # between 1 ms sleeps (hipEventSynchronize spins too, also on events created with blocking sync: measured 2.5 busy cores
    # per rank with it, against 1.7 unpaced).  Two passes in flight keep >= 25 ms of work queued, so the GPU never waits
    # for the host and a millisecond of polling granularity costs nothing.  0 disables the pacing.
    max_inflight_forwards = 2

    def _init_throttle(self) -> None:
        import collections
        self._inflight = collections.deque()

    def _pace(self, device) -> None:
        n = self.max_inflight_forwards
        if n <= 0 or torch.compiler.is_compiling() or torch.cuda.is_current_stream_capturing():
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._inflight.append(ev)
        while len(self._inflight) > n:
            oldest = self._inflight.popleft()
            while not oldest.query():
                time.sleep(0.001)

    def share(self) -> "WanDiffusionWrapper":
        """A second wrapper over the SAME device weights (own pointer tables / scratch), for a second
        rollout running concurrently on another HIP stream."""
        other = WanDiffusionWrapper.__new__(WanDiffusionWrapper)
        torch.nn.Module.__init__(other)
        other.uniform_timestep = self.uniform_timestep
        other.scheduler = self.scheduler
        other.model = self.model
        other.seq_len = self.seq_len
        other._evict_scratch = None
        other._init_throttle()
        return other

    # --- reference API ------------------------------------------------------------------------
    def get_scheduler(self):
        return self.scheduler

    def post_init(self):
        self.get_scheduler()

    def enable_gradient_checkpointing(self):
        raise NotImplementedError("inference-only path")

    # --- cache plumbing -----------------------------------------------------------------------
    @staticmethod
    def _read_indices(kv_cache: List[dict]):
        """Host values of (global_end, local_end).  The pipeline's integers are mirrored in the
        first layer's dict; a mirror is valid only while the dict still holds the very index
        tensors we last updated (the reference's reset REBINDS them, causal_inference.py:128-132)."""
        d = kv_cache[0]
        mir = d.get("_sf_mirror")
        if mir is not None and mir[0] is d["global_end_index"] and mir[1] is d["local_end_index"]:
            return mir[2], mir[3]
        return int(d["global_end_index"].item()), int(d["local_end_index"].item())

    @staticmethod
    def _shared_index_buffer(kv_cache: List[dict]) -> Optional[Tensor]:
        """The int64 [L, 2] buffer all layers' index tensors are views of (caches built by our pipeline,
        `_initialize_kv_cache`), while the dicts still hold those very views; None for foreign or rebound caches."""
        d0 = kv_cache[0]
        buf, views = d0.get("_sf_index_buffer"), d0.get("_sf_index_views")
        if buf is None or views is None or len(views) != len(kv_cache):
            return None
        for kv, (g, l) in zip(kv_cache, views):
            if kv["global_end_index"] is not g or kv["local_end_index"] is not l:
                return None
        return buf

    @staticmethod
    def _write_indices(kv_cache: List[dict], global_end: int, local_end: int, done_by_kernel: bool = False) -> None:
        """causal_model.py:235-236.  With the shared buffer the forward call's last kernel has written every row
        (`kv_index_out`); foreign caches get the reference's per-layer fills.  The host mirror is refreshed either way."""
        d0 = kv_cache[0]
        if not done_by_kernel:
            for kv in kv_cache:
                kv["global_end_index"].fill_(global_end)
                kv["local_end_index"].fill_(local_end)
        d0["_sf_mirror"] = (d0["global_end_index"], d0["local_end_index"], global_end, local_end)

    # --- two passes in one call ------------------------------------------------------------------
    def can_pair(self, conditional_dict: dict) -> bool:
        """`forward_pair` covers the plain text-conditioned rollout (no pose tokens / image conditioning)."""
        return conditional_dict.get("add_condition") is None and conditional_dict.get("clip_feature") is None \
            and conditional_dict.get("y") is None

    @torch.no_grad()
    def forward_pair(self, context_input: Tensor, context_timestep: Tensor, noisy_image_or_video: Tensor, timestep: Tensor,
                     conditional_dict: dict, kv_cache: List[dict], crossattn_cache: List[dict], context_start: int, current_start: int):
        """Extension (no counterpart call in the reference, which runs these back to back, causal_inference.py:226-235 then
        :190-205 of the next chunk): the context pass of one chunk -- `context_input` = its denoised latents at
        `context_timestep`, cache positions from `context_start`, only the KV cache is updated -- and the FIRST denoising
        pass of the next chunk (`noisy_image_or_video`, `timestep`, `current_start`) as one call.  Same results bit for
        bit as `forward(..., cache_only=True)` followed by `forward(...)`; returns (flow_pred, pred_x0) of the second."""
        mdl = self.model
        shape = mdl.shape
        xs = []
        for x in (context_input, noisy_image_or_video):
            assert x.dim() == 5 and x.shape[2] == shape.in_dim, "inputs must be [B, F, C, H, W] latents"
            xs.append(x.to(device=mdl.device, dtype=torch.bfloat16).contiguous())
        assert xs[0].shape == xs[1].shape, "forward_pair: both passes must have the same number of frames"
        B, F, _, H, W = xs[1].shape
        ts = []
        for t in (context_timestep, timestep):
            t = t.to(mdl.device)
            if t.dim() == 1:
                t = t.unsqueeze(1)
            ts.append((t if t.dtype == torch.int64 else t.to(torch.float32)).contiguous())
        assert ts[0].shape == ts[1].shape and ts[0].dtype == ts[1].dtype and ts[0].shape[0] == B, "forward_pair: timesteps must match in shape and dtype"
        assert len(kv_cache) == mdl.num_layers and len(crossattn_cache) == mdl.num_layers
        assert crossattn_cache[0]["is_init"], "forward_pair: the cross-attention cache is filled by the chunk's earlier passes"
        fs = (H // 2) * (W // 2)
        n_new = F * fs
        cap = kv_cache[0]["k"].shape[1]
        window = cap if mdl.local_attn_size == -1 else mdl.local_attn_size * fs
        global_end, local_end = self._read_indices(kv_cache)
        plan0 = plan_cache_update(local_end, global_end, context_start, n_new, cap, mdl.local_attn_size, mdl.sink_size * fs, window)
        plan1 = plan_cache_update(plan0.local_end, plan0.global_end, current_start, n_new, cap, mdl.local_attn_size, mdl.sink_size * fs, window)
        scratch = None
        if plan0.evict > 0 or plan1.evict > 0:
            need = B * max(plan0.keep, plan1.keep) * shape.dim * 2
            if self._evict_scratch is None or self._evict_scratch.numel() < need:
                self._evict_scratch = torch.empty(B * cap * shape.dim * 2, dtype=torch.uint8, device=mdl.device)
            scratch = self._evict_scratch
        index_buf = self._shared_index_buffer(kv_cache)
        flow, x0 = mdl.forward_pair(xs[0], ts[0], xs[1], ts[1], [kv["k"] for kv in kv_cache], [kv["v"] for kv in kv_cache],
                                    [c["k"] for c in crossattn_cache], [c["v"] for c in crossattn_cache], plan0, plan1,
                                    context_start // fs, current_start // fs, scratch, kv_index=index_buf)
        self._write_indices(kv_cache, plan1.global_end, plan1.local_end, done_by_kernel=index_buf is not None)
        self._pace(mdl.device)
        return flow, x0

    # --- the hot call --------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, noisy_image_or_video: Tensor, conditional_dict: dict, timestep: Tensor,
                kv_cache: Optional[List[dict]] = None, crossattn_cache: Optional[List[dict]] = None,
                current_start: Optional[int] = None, classify_mode: Optional[bool] = False,
                concat_time_embeddings: Optional[bool] = False, clean_x: Optional[Tensor] = None,
                aug_t: Optional[Tensor] = None, cache_start: Optional[int] = None,
                add_condition: Optional[Tensor] = None, clip_feature: Optional[Tensor] = None, y: Optional[Tensor] = None,
                cache_only: bool = False):
        """`cache_only=True` (extension): the caller only wants the KV-cache update (context pass,
        initial-latent warm-up) and gets (None, None) back; see sf_forward_args.cache_only."""
        if kv_cache is None or crossattn_cache is None:
            raise NotImplementedError("only the KV-cached inference branch is implemented (kv_cache / crossattn_cache required)")
        if classify_mode or clean_x is not None or aug_t is not None:
            raise NotImplementedError("training-only branches (classify_mode / teacher forcing) are out of scope")
        if add_condition is None:
            add_condition = conditional_dict.get("add_condition")
        if clip_feature is not None or y is not None \
                or conditional_dict.get("clip_feature") is not None or conditional_dict.get("y") is not None:
            raise NotImplementedError("image conditioning (clip_feature, y: the i2v model type) is not implemented")
        mdl = self.model
        shape = mdl.shape
        x = noisy_image_or_video
        assert x.dim() == 5, "noisy_image_or_video must be [B, F, C, H, W]"
        B, F, Cin, H, W = x.shape
        assert Cin == shape.in_dim, f"expected {shape.in_dim} latent channels, got {Cin}"
        assert len(kv_cache) == mdl.num_layers and len(crossattn_cache) == mdl.num_layers, \
            "cache lists must have one entry per transformer block"
        if current_start is None:
            current_start = 0
        x = x.to(device=mdl.device, dtype=torch.bfloat16).contiguous()
        t = timestep.to(mdl.device)
        if t.dim() == 1:
            t = t.unsqueeze(1)
        t = (t if t.dtype == torch.int64 else t.to(torch.float32)).contiguous()
        assert t.shape[0] == B, "timestep must be [B, groups]"

        fs = (H // 2) * (W // 2)
        n_new = F * fs
        cap = kv_cache[0]["k"].shape[1]
        k0 = kv_cache[0]["k"]
        assert tuple(k0.shape) == (B, cap, shape.num_heads, shape.head_dim) and k0.dtype == torch.bfloat16 and k0.is_contiguous(), \
            f"kv cache must be contiguous bf16 [B, S, {shape.num_heads}, {shape.head_dim}], got {tuple(k0.shape)} {k0.dtype}"
        c0 = crossattn_cache[0]["k"]
        assert tuple(c0.shape) == (B, shape.text_len, shape.num_heads, shape.head_dim) and c0.is_contiguous(), \
            f"cross-attention cache must be [B, {shape.text_len}, {shape.num_heads}, {shape.head_dim}]"

        global_end, local_end = self._read_indices(kv_cache)
        window = cap if mdl.local_attn_size == -1 else mdl.local_attn_size * fs
        plan = plan_cache_update(local_end, global_end, current_start, n_new, cap, mdl.local_attn_size,
                                 mdl.sink_size * fs, window)
        scratch = None
        if plan.evict > 0:
            need = B * plan.keep * shape.dim * 2
            if self._evict_scratch is None or self._evict_scratch.numel() < need:
                self._evict_scratch = torch.empty(B * cap * shape.dim * 2, dtype=torch.uint8, device=mdl.device)
            scratch = self._evict_scratch

        init_cross = not crossattn_cache[0]["is_init"]
        pe = None
        if init_cross:
            pe = conditional_dict["prompt_embeds"].to(device=mdl.device, dtype=torch.bfloat16)
            assert pe.dim() == 3 and pe.shape[0] == B and pe.shape[2] == shape.text_dim and pe.shape[1] <= shape.text_len, \
                f"prompt_embeds must be [B, <= {shape.text_len}, {shape.text_dim}], got {tuple(pe.shape)}"
            if pe.shape[1] < shape.text_len:  # zero-pad to text_len (causal_model.py:838-842)
                pe = torch.cat([pe, pe.new_zeros(B, shape.text_len - pe.shape[1], shape.text_dim)], dim=1)
            pe = pe.contiguous()

        if add_condition is not None:   # pose tokens [B, L_pose, 5120]: x += pose_proj(add_condition), the intent of
            # causal_model.py:786-819 (that branch raises in the reference snapshot: parity pinned by the oracle only)
            if not mdl.accepts_pose:   # (dim == 5120 models need none: their pose_proj is nn.Identity(), :500-501)
                raise ValueError(f"add_condition needs pose_proj weights in the state dict of a dim-{shape.dim} model")
            add_condition = add_condition.to(device=mdl.device, dtype=torch.bfloat16).contiguous()
            if add_condition.dim() != 3 or add_condition.shape[0] != B or add_condition.shape[1] != n_new:
                raise ValueError(f"add_condition spatial dim {add_condition.shape[1]} doesn't match "
                                 f"x spatial dim {n_new}. Check pose data processing.")
            assert add_condition.shape[2] == mdl.cmodel.pose_dim, "add_condition channel width must match pose_proj"
        index_buf = self._shared_index_buffer(kv_cache)
        flow, x0 = mdl.forward(x, t, pe, init_cross, [kv["k"] for kv in kv_cache], [kv["v"] for kv in kv_cache],
                               [c["k"] for c in crossattn_cache], [c["v"] for c in crossattn_cache], plan,
                               current_start // fs, scratch, cache_only=cache_only, add_condition=add_condition,
                               kv_index=index_buf)
        if init_cross:
            for c in crossattn_cache:
                c["is_init"] = True
        self._write_indices(kv_cache, plan.global_end, plan.local_end, done_by_kernel=index_buf is not None)
        self._pace(mdl.device)
        return flow, x0
