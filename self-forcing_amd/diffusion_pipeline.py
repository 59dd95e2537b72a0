"""`CausalDiffusionInferencePipeline` -- drop-in for pipeline/causal_diffusion_inference.py of the reference: the
many-step (50 by default) classifier-free-guidance sampler over the same causal generator.

Same constructor `(args, device, generator=None, text_encoder=None, vae=None, image_encoder=None)` and the same
`inference(noise, text_prompts, input_image, dwpose_data, random_ref_dwpose, initial_latent=None,
return_latents=False, start_frame_index=0)`; same attributes (`kv_cache_pos/neg`, `crossattn_cache_pos/neg`,
`sampling_steps`, `sample_solver`, `shift`, `num_frame_per_block`, ...) and cache-dict schema.

Per chunk (causal_diffusion_inference.py:370-457): a fresh FlowUniPCMultistepScheduler, then per step the generator
under the prompt and under the negative prompt (two KV / cross-attention cache sets), the guidance blend and one
scheduler step; finally a timestep-0 pass that rewrites the chunk's K/V in both caches.

What differs:
  * the two generator calls of a step are independent, so with `overlap_cfg=True` (default when the generator can
    `share()` its weights) the negative-prompt pass runs on a second HIP stream beside the prompt pass;
  * the blend and the scheduler's tensor arithmetic are `sf_lincomb_bf16` launches with host-evaluated scalars
    (`unipc.py`); timesteps come from the scheduler's host table, so the loop never waits on the device;
  * the constants the reference hard-codes (30 blocks, 1560 tokens per frame, 12x128 heads, 32760-token caches,
    :69-72, :464-487) are derived from the generator's shape and the latent size;
  * the fork's image / pose front end (CLIP image encoder, VAE-encoded `y`, the dwpose convolution stacks,
    :86-123, :154-173, :305-347) is outside the hot path: `input_image`, `dwpose_data`, `random_ref_dwpose` must be
    None; already-embedded pose tokens can be passed as `dwpose_data_emb` [B, C_pose, F_total, h, w] and reach the
    generator as `add_condition`, sliced per chunk exactly as :386-394 does;
  * the 'dpm++' solver branch (:526-536) is not implemented; the per-step `print`s are dropped.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .unipc import FlowUniPCMultistepScheduler
from .wan_wrapper import WanDiffusionWrapper


def _new_kv_cache(shape, n_layers: int, batch_size: int, cache_tokens: int, dtype, device) -> List[dict]:
    """Same dict schema as causal_diffusion_inference.py:459-487; the index tensors are views of one [L, 2] buffer."""
    index_buffer = torch.zeros(n_layers, 2, dtype=torch.long, device=device)
    kv = []
    for i in range(n_layers):
        kv.append({
            "k": torch.zeros([batch_size, cache_tokens, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
            "v": torch.zeros([batch_size, cache_tokens, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
            "global_end_index": index_buffer[i, 0:1],
            "local_end_index": index_buffer[i, 1:2],
            "_sf_index_buffer": index_buffer,
        })
    kv[0]["_sf_mirror"] = (kv[0]["global_end_index"], kv[0]["local_end_index"], 0, 0)
    kv[0]["_sf_index_views"] = [(d["global_end_index"], d["local_end_index"]) for d in kv]
    return kv


def _reset_kv_cache(kv: List[dict]) -> None:
    buf = kv[0].get("_sf_index_buffer")
    if buf is not None:
        buf.zero_()
        kv[0]["_sf_mirror"] = (kv[0]["global_end_index"], kv[0]["local_end_index"], 0, 0)
    else:  # foreign cache: rebind as the reference does (:221-231)
        dev = kv[0]["k"].device
        for d in kv:
            d["global_end_index"] = torch.tensor([0], dtype=torch.long, device=dev)
            d["local_end_index"] = torch.tensor([0], dtype=torch.long, device=dev)


def _new_crossattn_cache(shape, n_layers: int, batch_size: int, dtype, device) -> List[dict]:
    return [{
        "k": torch.zeros([batch_size, shape.text_len, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
        "v": torch.zeros([batch_size, shape.text_len, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
        "is_init": False,
    } for _ in range(n_layers)]


class CausalDiffusionInferencePipeline(torch.nn.Module):
    def __init__(self, args, device, generator=None, text_encoder=None, vae=None, image_encoder=None,
                 overlap_cfg: Optional[bool] = None):
        super().__init__()
        self.device = torch.device(device)
        self.generator = WanDiffusionWrapper(**getattr(args, "model_kwargs", {}), is_causal=True, device=device) \
            if generator is None else generator
        # as the reference (causal_inference.py:19-23): build the default components when none is injected; they load
        # the reference's default checkpoints (weights-only) and raise FileNotFoundError when those are absent
        if text_encoder is None:
            from .text_encoder import WanTextEncoder
            text_encoder = WanTextEncoder(device=device)
        if vae is None:
            from .vae import WanVAEWrapper
            vae = WanVAEWrapper(device=device)
        self.text_encoder = text_encoder
        self.vae = vae
        self.image_encoder = image_encoder      # accepted for signature parity; the CLIP front end is out of scope

        self.num_train_timesteps = args.num_train_timestep
        self.sampling_steps = 50
        self.sample_solver = "unipc"
        self.shift = args.timestep_shift

        self.num_transformer_blocks = self.generator.model.num_layers
        self.frame_seq_length = 1560  # refined from the latent size at inference()
        self.kv_cache_pos = None
        self.kv_cache_neg = None
        self.crossattn_cache_pos = None
        self.crossattn_cache_neg = None
        self.args = args
        self.torch_dtype = torch.bfloat16
        self.num_frame_per_block = getattr(args, "num_frame_per_block", 1)
        self.independent_first_frame = args.independent_first_frame
        self.local_attn_size = self.generator.model.local_attn_size
        if self.num_frame_per_block > 1:
            self.generator.model.num_frame_per_block = self.num_frame_per_block

        # the negative-prompt pass on a second stream needs its own activation workspace over the same weights
        can_share = hasattr(self.generator, "share")
        self.overlap_cfg = can_share if overlap_cfg is None else bool(overlap_cfg)
        if self.overlap_cfg and not can_share:
            raise ValueError("overlap_cfg=True needs a generator with share() (self_forcing_amd.WanDiffusionWrapper)")
        self._generator_neg = self.generator.share() if self.overlap_cfg else self.generator
        self._side_stream = None
        import inspect
        try:
            self._cache_only_kw = {"cache_only": True} if "cache_only" in inspect.signature(self.generator.forward).parameters else {}
        except (TypeError, ValueError):
            self._cache_only_kw = {}
        self._cache_key = None
        self.timesteps = None

    # ------------------------------------------------------------------------------------------
    def _initialize_sample_scheduler(self, noise):
        """:517-540."""
        if self.sample_solver != "unipc":
            raise NotImplementedError("Unsupported solver." if self.sample_solver != "dpm++"
                                      else "the 'dpm++' branch of the reference is not implemented; use 'unipc'")
        sample_scheduler = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                       use_dynamic_shifting=False)
        sample_scheduler.set_timesteps(self.sampling_steps, device=None, shift=self.shift)
        self.timesteps = sample_scheduler.timesteps
        return sample_scheduler

    def _cache_tokens(self, total_frames: int) -> int:
        if self.local_attn_size != -1:
            return self.local_attn_size * self.frame_seq_length
        return max(21, total_frames) * self.frame_seq_length

    def _initialize_kv_cache(self, batch_size, dtype, device, cache_tokens: Optional[int] = None):
        shape = self.generator.model.shape
        if cache_tokens is None:
            cache_tokens = self._cache_tokens(0)
        self.kv_cache_pos = _new_kv_cache(shape, self.num_transformer_blocks, batch_size, cache_tokens, dtype, device)
        self.kv_cache_neg = _new_kv_cache(shape, self.num_transformer_blocks, batch_size, cache_tokens, dtype, device)

    def _initialize_crossattn_cache(self, batch_size, dtype, device):
        shape = self.generator.model.shape
        self.crossattn_cache_pos = _new_crossattn_cache(shape, self.num_transformer_blocks, batch_size, dtype, device)
        self.crossattn_cache_neg = _new_crossattn_cache(shape, self.num_transformer_blocks, batch_size, dtype, device)

    # ------------------------------------------------------------------------------------------
    def _both(self, x, cond_dict, uncond_dict, timestep, current_start, cache_only: bool):
        """The generator under both conditions on the same input; returns (flow_cond, flow_uncond)."""
        kw = dict(noisy_image_or_video=x, timestep=timestep, current_start=current_start, cache_start=None)
        if cache_only:
            kw.update(self._cache_only_kw)
        if not self.overlap_cfg:
            fc, _ = self.generator(conditional_dict=cond_dict, kv_cache=self.kv_cache_pos,
                                   crossattn_cache=self.crossattn_cache_pos, **kw)
            fu, _ = self.generator(conditional_dict=uncond_dict, kv_cache=self.kv_cache_neg,
                                   crossattn_cache=self.crossattn_cache_neg, **kw)
            return fc, fu
        main = torch.cuda.current_stream(x.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=x.device)
        side = self._side_stream
        side.wait_stream(main)                       # x and the timestep tensor are produced on `main`
        fc, _ = self.generator(conditional_dict=cond_dict, kv_cache=self.kv_cache_pos,
                               crossattn_cache=self.crossattn_cache_pos, **kw)
        with torch.cuda.stream(side):
            fu, _ = self._generator_neg(conditional_dict=uncond_dict, kv_cache=self.kv_cache_neg,
                                        crossattn_cache=self.crossattn_cache_neg, **kw)
            if fu is not None:
                fu.record_stream(main)
        x.record_stream(side)
        timestep.record_stream(side)
        main.wait_stream(side)
        return fc, fu

    def inference(self, noise: torch.Tensor, text_prompts: List[str], input_image=None, dwpose_data=None,
                  random_ref_dwpose=None, initial_latent: Optional[torch.Tensor] = None, return_latents: bool = False,
                  start_frame_index: Optional[int] = 0, dwpose_data_emb: Optional[torch.Tensor] = None):
        """noise [B, F, C, H, W] -> video in [0, 1] (and the latents)."""
        if input_image is not None or dwpose_data is not None or random_ref_dwpose is not None:
            raise NotImplementedError("the image / pose front end (CLIP, VAE encode, dwpose convolutions) is outside "
                                      "this path: pass embedded pose tokens as dwpose_data_emb=, or None")
        batch_size, num_frames, num_channels, height, width = noise.shape
        if not self.independent_first_frame or (self.independent_first_frame and initial_latent is not None):
            assert num_frames % self.num_frame_per_block == 0
            num_blocks = num_frames // self.num_frame_per_block
        else:
            assert (num_frames - 1) % self.num_frame_per_block == 0
            num_blocks = (num_frames - 1) // self.num_frame_per_block
        num_input_frames = initial_latent.shape[1] if initial_latent is not None else 0
        num_output_frames = num_frames + num_input_frames
        self.frame_seq_length = (height // 2) * (width // 2)
        conditional_dict = dict(self.text_encoder(text_prompts=text_prompts))
        unconditional_dict = dict(self.text_encoder(text_prompts=[self.args.negative_prompt] * len(text_prompts)))

        output = torch.zeros([batch_size, num_output_frames, num_channels, height, width], device=noise.device, dtype=noise.dtype)

        # Step 1: caches (:203-231)
        key = (batch_size, self._cache_tokens(num_output_frames), noise.device)
        if self.kv_cache_pos is None or self._cache_key != key:
            self._initialize_kv_cache(batch_size, noise.dtype, noise.device, cache_tokens=key[1])
            self._initialize_crossattn_cache(batch_size, noise.dtype, noise.device)
            self._cache_key = key
        else:
            for block_index in range(self.num_transformer_blocks):
                self.crossattn_cache_pos[block_index]["is_init"] = False
                self.crossattn_cache_neg[block_index]["is_init"] = False
            _reset_kv_cache(self.kv_cache_pos)
            _reset_kv_cache(self.kv_cache_neg)

        # Step 2: context frames into both caches (:233-297)
        fs = self.frame_seq_length
        current_start_frame = start_frame_index
        cache_start_frame = 0
        if initial_latent is not None:
            timestep = torch.zeros([batch_size, 1], device=noise.device, dtype=torch.int64)
            if self.independent_first_frame:
                assert (num_input_frames - 1) % self.num_frame_per_block == 0
                num_input_blocks = (num_input_frames - 1) // self.num_frame_per_block
                output[:, :1] = initial_latent[:, :1]
                self._both(initial_latent[:, :1], conditional_dict, unconditional_dict, timestep,
                           current_start_frame * fs, cache_only=True)
                current_start_frame += 1
                cache_start_frame += 1
            else:
                assert num_input_frames % self.num_frame_per_block == 0
                num_input_blocks = num_input_frames // self.num_frame_per_block
            for _ in range(num_input_blocks):
                ref = initial_latent[:, cache_start_frame:cache_start_frame + self.num_frame_per_block]
                output[:, cache_start_frame:cache_start_frame + self.num_frame_per_block] = ref
                self._both(ref, conditional_dict, unconditional_dict, timestep, current_start_frame * fs, cache_only=True)
                current_start_frame += self.num_frame_per_block
                cache_start_frame += self.num_frame_per_block

        # Step 3: temporal denoising loop (:359-457)
        all_num_frames = [self.num_frame_per_block] * num_blocks
        if self.independent_first_frame and initial_latent is None:
            all_num_frames = [1] + all_num_frames
        if dwpose_data_emb is not None:
            expected_pose_frames = current_start_frame + sum(all_num_frames)
            assert dwpose_data_emb.shape[2] == expected_pose_frames, (
                f"dwpose_data_emb has {dwpose_data_emb.shape[2]} frames, "
                f"but expected {expected_pose_frames} to match the output timeline.")
        guidance = float(self.args.guidance_scale)
        for current_num_frames in all_num_frames:
            latents = noise[:, cache_start_frame - num_input_frames:
                            cache_start_frame + current_num_frames - num_input_frames].contiguous()
            if dwpose_data_emb is not None:
                start, end = current_start_frame, current_start_frame + current_num_frames
                if end > dwpose_data_emb.shape[2]:
                    raise ValueError("dwpose_data has fewer frames than required for the current block.")
                condition = dwpose_data_emb[:, :, start:end].permute(0, 2, 3, 4, 1).flatten(1, 3).contiguous()
                conditional_dict["add_condition"] = condition
                unconditional_dict["add_condition"] = condition
            else:
                conditional_dict.pop("add_condition", None)
                unconditional_dict.pop("add_condition", None)

            sample_scheduler = self._initialize_sample_scheduler(noise)
            for t in sample_scheduler.timesteps_host.tolist():
                timestep = torch.full([batch_size, current_num_frames], float(t), device=noise.device, dtype=torch.float32)
                flow_cond, flow_uncond = self._both(latents, conditional_dict, unconditional_dict, timestep,
                                                    current_start_frame * fs, cache_only=False)
                # uncond + g (cond - uncond), :423-424
                flow_pred = ops.lincomb([flow_uncond, flow_cond], [1.0 - guidance, guidance])
                latents = sample_scheduler.step(flow_pred, t, latents, return_dict=False)[0]

            output[:, cache_start_frame:cache_start_frame + current_num_frames] = latents
            # rerun at timestep zero so both caches hold the clean chunk (:438-455)
            self._both(latents, conditional_dict, unconditional_dict, torch.zeros_like(timestep),
                       current_start_frame * fs, cache_only=True)
            current_start_frame += current_num_frames
            cache_start_frame += current_num_frames

        # Step 4: decode (:461-468)
        video = self.vae.decode_to_pixel(output)
        video = (video * 0.5 + 0.5).clamp(0, 1)
        if return_latents:
            return video, output
        return video
