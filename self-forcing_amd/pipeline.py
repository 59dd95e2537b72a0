"""`CausalInferencePipeline` -- drop-in for pipeline/causal_inference.py of the reference.

Same constructor `(args, device, generator=None, text_encoder=None, vae=None)`, same
`inference(noise, text_prompts, initial_latent=None, return_latents=False, profile=False,
low_memory=False)`, same attributes read by the reference's other callers (`kv_cache1`,
`crossattn_cache`, `denoising_step_list`, `scheduler`, `frame_seq_length`, `num_frame_per_block`,
`num_transformer_blocks`; demo.py:309-404) and the same cache-dict schema.

What differs: the five constants the reference hard-codes for Wan-1.3B/480p
(30 blocks, 1560 tokens/frame, 12x128 heads, 32760-token cache; causal_inference.py:33-34,
:288-293) are derived from the generator's shape and the latent size; when no text encoder / VAE is
injected, `WanTextEncoder()` / `WanVAEWrapper()` are built from the reference's default local
checkpoint paths as causal_inference.py:19-23 does (weights-only loads; FileNotFoundError when the
files are absent -- benchmarks and tests inject the stand-ins of `harness.py` or seeded-weight
instances instead); the per-step `print` is dropped; `low_memory` is accepted and ignored (288 GB HBM).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .wan_wrapper import WanDiffusionWrapper


class CausalInferencePipeline(torch.nn.Module):
    def __init__(self, args, device, generator=None, text_encoder=None, vae=None):
        super().__init__()
        self.device_ = torch.device(device)
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

        # causal hyper-parameters (causal_inference.py:25-45)
        self.scheduler = self.generator.get_scheduler()
        self.denoising_step_list = torch.tensor(args.denoising_step_list, dtype=torch.long)
        if args.warp_denoising_step:
            timesteps = torch.cat((self.scheduler.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
            self.denoising_step_list = timesteps[1000 - self.denoising_step_list]

        self.num_transformer_blocks = self.generator.model.num_layers
        self.frame_seq_length = 1560  # refined from the latent size at inference()
        self.kv_cache1 = None
        self.crossattn_cache = None
        self.args = args
        self.num_frame_per_block = getattr(args, "num_frame_per_block", 1)
        self.independent_first_frame = args.independent_first_frame
        self.local_attn_size = self.generator.model.local_attn_size
        if self.num_frame_per_block > 1:
            self.generator.model.num_frame_per_block = self.num_frame_per_block
        # re-noise source; the reference calls torch.randn_like (causal_inference.py:208).  Tests and
        # the CPU-baseline comparison inject pre-drawn tensors here so both sides consume the same eps.
        self.noise_source: Optional[Callable[[torch.Tensor], torch.Tensor]] = None
        # context / warm-up passes only update the KV cache; a generator that advertises `cache_only`
        # (ours) may skip what nothing reads.  Foreign generators are called exactly as the reference does.
        import inspect
        try:
            self._cache_only_kw = {"cache_only": True} if "cache_only" in inspect.signature(self.generator.forward).parameters else {}
        except (TypeError, ValueError):
            self._cache_only_kw = {}
        self.last_profile = None
        self._cache_key = None
        # run a chunk's context pass together with the next chunk's first denoising pass (one call, same results; see
        # _denoise_chunks); False = one generator call per pass, as the reference
        self.pair_context_with_next = True
        # ... when a pass has at most this many tokens (batch x frames x tokens per frame): pairing pays through the GEMMs'
        # tile quantisation -- one prompt at 480p, 4680 rows: +2.0 % (84.9 -> 86.5, 85.2 -> 86.9 frames/s, same box,
        # alternating runs); at 9360 rows (two prompts per call) the GEMMs' rounds are already full: +-0.3 %, not worth
        # the second workspace
        self.pair_max_rows = 6144

    # ------------------------------------------------------------------------------------------
    def _randn_like(self, t: torch.Tensor) -> torch.Tensor:
        if self.noise_source is not None:
            return self.noise_source(t).to(device=t.device, dtype=t.dtype)
        return torch.randn_like(t)

    def inference(self, noise: torch.Tensor, text_prompts: List[str], initial_latent: Optional[torch.Tensor] = None,
                  return_latents: bool = False, profile: bool = False, low_memory: bool = False):
        """noise [B, F, C, H, W] -> video in [0, 1] (and the latents)."""
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
        conditional_dict = self.text_encoder(text_prompts=text_prompts)

        output = torch.zeros([batch_size, num_output_frames, num_channels, height, width], device=noise.device, dtype=noise.dtype)

        if profile:
            ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
            init_start, init_end, diffusion_start, diffusion_end, vae_start, vae_end = ev(), ev(), ev(), ev(), ev(), ev()
            block_events = []
            init_start.record()

        # Step 1: (re)initialise the caches (causal_inference.py:111-132)
        key = (batch_size, self._cache_tokens(num_output_frames), noise.device)
        if self.kv_cache1 is None or self._cache_key != key:
            self._initialize_kv_cache(batch_size, noise.dtype, noise.device, cache_tokens=key[1])
            self._initialize_crossattn_cache(batch_size, noise.dtype, noise.device)
            self._cache_key = key
        else:
            for block_index in range(self.num_transformer_blocks):
                self.crossattn_cache[block_index]["is_init"] = False
            self._reset_kv_indices()

        # Step 2: cache the context frames (causal_inference.py:134-169)
        gen = self.generator
        current_start_frame = 0
        if initial_latent is not None:
            timestep = torch.zeros([batch_size, 1], device=noise.device, dtype=torch.int64)
            if self.independent_first_frame:
                assert (num_input_frames - 1) % self.num_frame_per_block == 0
                num_input_blocks = (num_input_frames - 1) // self.num_frame_per_block
                output[:, :1] = initial_latent[:, :1]
                gen(noisy_image_or_video=initial_latent[:, :1], conditional_dict=conditional_dict, timestep=timestep,
                    kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache,
                    current_start=current_start_frame * self.frame_seq_length, **self._cache_only_kw)
                current_start_frame += 1
            else:
                assert num_input_frames % self.num_frame_per_block == 0
                num_input_blocks = num_input_frames // self.num_frame_per_block
            for _ in range(num_input_blocks):
                ref = initial_latent[:, current_start_frame:current_start_frame + self.num_frame_per_block]
                output[:, current_start_frame:current_start_frame + self.num_frame_per_block] = ref
                gen(noisy_image_or_video=ref, conditional_dict=conditional_dict, timestep=timestep,
                    kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache,
                    current_start=current_start_frame * self.frame_seq_length, **self._cache_only_kw)
                current_start_frame += self.num_frame_per_block

        if profile:
            init_end.record()
            diffusion_start.record()

        # Step 3: temporal denoising loop (causal_inference.py:176-244)
        all_num_frames = [self.num_frame_per_block] * num_blocks
        if self.independent_first_frame and initial_latent is None:
            all_num_frames = [1] + all_num_frames
        for chunk_idx, start_frame, denoised_pred in self._denoise_chunks(
                noise, conditional_dict, all_num_frames, current_start_frame, num_input_frames, skip_last_context=False,
                on_chunk_start=(lambda: block_events.append([ev(), ev()]) or block_events[-1][0].record()) if profile else None,
                on_chunk_end=(lambda: block_events[-1][1].record()) if profile else None):
            output[:, start_frame:start_frame + denoised_pred.shape[1]] = denoised_pred

        if profile:
            diffusion_end.record()
            vae_start.record()

        # Step 4: decode (causal_inference.py:254-256)
        video = self.vae.decode_to_pixel(output, use_cache=False)
        video = (video * 0.5 + 0.5).clamp(0, 1)

        if profile:
            vae_end.record()
            torch.cuda.synchronize()
            init_time = init_start.elapsed_time(init_end)
            diffusion_time = diffusion_start.elapsed_time(diffusion_end)
            vae_time = vae_start.elapsed_time(vae_end)
            blocks = [a.elapsed_time(b) for a, b in block_events]
            total = init_time + diffusion_time + vae_time
            self.last_profile = {"init_ms": init_time, "diffusion_ms": diffusion_time, "vae_ms": vae_time,
                                 "block_ms": blocks, "total_ms": total}
            share = lambda part, whole: 100.0 * part / max(whole, 1e-9)  # noqa: E731
            report = [f"rollout profile: {total:.1f} ms = cache setup {init_time:.1f} ms ({share(init_time, total):.1f} %) + "
                      f"denoising {diffusion_time:.1f} ms ({share(diffusion_time, total):.1f} %) + decode {vae_time:.1f} ms "
                      f"({share(vae_time, total):.1f} %)"]
            report += [f"  chunk {i}: {bt:.1f} ms ({share(bt, diffusion_time):.1f} % of denoising)" for i, bt in enumerate(blocks)]
            print("\n".join(report))

        if return_latents:
            return video, output
        return video

    # ------------------------------------------------------------------------------------------
    def _denoise_chunks(self, noise, conditional_dict, all_num_frames, current_start_frame, num_input_frames,
                        skip_last_context, on_chunk_start=None, on_chunk_end=None):
        """The chunk loop of causal_inference.py:176-244 as a generator: yields
        (chunk_index, start_frame, x0 [B, f, C, H, W]) as soon as a chunk's last denoising step is
        enqueued; the context pass that rewrites the chunk's K/V "clean" follows (and is skipped for
        the final chunk when `skip_last_context`, as demo.py:396 does: nothing reads that update)."""
        gen = self.generator
        batch_size = noise.shape[0]
        # Every timestep tensor of the rollout is built HERE, once: the reference builds `ones([B, f], int64) * t` for each
        # forward and `t_next * ones([B * f], long)` for each re-noise (causal_inference.py:190-216, :228) -- three small
        # launches per step of kernels this library does not own.  One host->device copy of the step list, one broadcast
        # per distinct chunk length; the loop below then launches nothing of torch's but `randn_like` (the reference's
        # global-RNG re-noise, :208, which must stay) and the caller's copy of the chunk into `output`.
        steps = self.denoising_step_list.to(noise.device)
        ctx_noise = getattr(self.args, "context_noise", 0)
        tables = {}
        for f in set(all_num_frames):
            ones = torch.ones([batch_size, f], device=noise.device, dtype=torch.int64)
            per_step = (ones.unsqueeze(0) * steps.reshape(-1, 1, 1)).contiguous()      # [S, B, f], dtype as `ones * t`
            tables[f] = (list(per_step.unbind(0)), [t.flatten() for t in per_step.unbind(0)],
                         torch.ones_like(per_step[0]) * ctx_noise)
        n_steps = steps.shape[0]
        # A chunk's context pass and the next chunk's first denoising pass run back to back in the reference (:226-235, then
        # :190-205) and layer l of the second needs only layer l's K / V of the first: a generator that offers `forward_pair`
        # (ours) runs them as ONE call -- bit-identical latents, twice the rows per GEMM (sf_dit_forward_pair).  The re-noise
        # draws keep their order: nothing is drawn between a chunk's last step and the next chunk's first.
        pair_ok = bool(self.pair_context_with_next and self._cache_only_kw and hasattr(gen, "forward_pair")
                       and gen.can_pair(conditional_dict)
                       and batch_size * max(all_num_frames) * self.frame_seq_length <= self.pair_max_rows)
        first_pred = None          # x0 of this chunk's first step when the previous chunk's pair has already computed it
        for chunk_idx, current_num_frames in enumerate(all_num_frames):
            if on_chunk_start is not None:
                on_chunk_start()
            noisy_input = noise[:, current_start_frame - num_input_frames:
                                current_start_frame + current_num_frames - num_input_frames]
            start_tok = current_start_frame * self.frame_seq_length
            step_ts, step_ts_flat, context_timestep = tables[current_num_frames]
            for index in range(n_steps):
                if index == 0 and first_pred is not None:
                    denoised_pred, first_pred = first_pred, None
                else:
                    _, denoised_pred = gen(noisy_image_or_video=noisy_input, conditional_dict=conditional_dict,
                                           timestep=step_ts[index], kv_cache=self.kv_cache1,
                                           crossattn_cache=self.crossattn_cache, current_start=start_tok)
                if index < n_steps - 1:
                    flat = denoised_pred.flatten(0, 1)
                    noisy_input = self.scheduler.add_noise(flat, self._randn_like(flat), step_ts_flat[index + 1]
                                                           ).unflatten(0, denoised_pred.shape[:2])
            yield chunk_idx, current_start_frame, denoised_pred
            # rerun at the context timestep so the cache holds clean K/V (causal_inference.py:226-235)
            is_last = chunk_idx == len(all_num_frames) - 1
            if not (skip_last_context and is_last):
                if pair_ok and not is_last and all_num_frames[chunk_idx + 1] == current_num_frames:
                    nxt = current_start_frame + current_num_frames
                    first_pred = gen.forward_pair(denoised_pred, context_timestep,
                                                  noise[:, nxt - num_input_frames:nxt + current_num_frames - num_input_frames], step_ts[0],
                                                  conditional_dict, self.kv_cache1, self.crossattn_cache, start_tok,
                                                  nxt * self.frame_seq_length)[1]
                else:
                    gen(noisy_image_or_video=denoised_pred, conditional_dict=conditional_dict, timestep=context_timestep,
                        kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache, current_start=start_tok,
                        **self._cache_only_kw)
            if on_chunk_end is not None:
                on_chunk_end()
            current_start_frame += current_num_frames

    def stream(self, noise: torch.Tensor, text_prompts: List[str], skip_last_context: bool = True,
               overlap_decode: bool = False):
        """Chunk-at-a-time generation (the streaming boundary; mirrors the inline loop of the
        reference's demo.py:303-468): yields `(chunk_index, latents [B, f, C, H, W], pixels)` per chunk.
        `pixels` comes from `vae.decode_chunk(latents, chunk_index)` when the injected VAE has a streaming
        decoder, else from `decode_to_pixel` on the chunk alone.

        overlap_decode=False: chunk k is decoded right after it is denoised and yielded at once (lowest
        latency to the first frame).  overlap_decode=True: the decode of chunk k runs on a second HIP stream
        while chunk k+1 is being denoised (it fills the CUs the denoiser's single-round kernels leave idle);
        chunk k is then yielded one chunk later, as soon as its decode has finished."""
        batch_size, num_frames, num_channels, height, width = noise.shape
        if self.independent_first_frame:
            assert (num_frames - 1) % self.num_frame_per_block == 0
            all_num_frames = [1] + [self.num_frame_per_block] * ((num_frames - 1) // self.num_frame_per_block)
        else:
            assert num_frames % self.num_frame_per_block == 0
            all_num_frames = [self.num_frame_per_block] * (num_frames // self.num_frame_per_block)
        self.frame_seq_length = (height // 2) * (width // 2)
        conditional_dict = self.text_encoder(text_prompts=text_prompts)
        key = (batch_size, self._cache_tokens(num_frames), noise.device)
        if self.kv_cache1 is None or self._cache_key != key:
            self._initialize_kv_cache(batch_size, noise.dtype, noise.device, cache_tokens=key[1])
            self._initialize_crossattn_cache(batch_size, noise.dtype, noise.device)
            self._cache_key = key
        else:
            for block_index in range(self.num_transformer_blocks):
                self.crossattn_cache[block_index]["is_init"] = False
            self._reset_kv_indices()
        decode_chunk = getattr(self.vae, "decode_chunk", None)

        def decode(x0, chunk_idx):
            pixels = decode_chunk(x0, chunk_idx) if decode_chunk is not None else self.vae.decode_to_pixel(x0, use_cache=False)
            return (pixels * 0.5 + 0.5).clamp(0, 1)

        chunks = self._denoise_chunks(noise, conditional_dict, all_num_frames, 0, 0, skip_last_context)
        if not overlap_decode:
            for chunk_idx, start_frame, x0 in chunks:
                yield chunk_idx, x0, decode(x0, chunk_idx)
            return
        if getattr(self, "_decode_stream", None) is None:
            self._decode_stream = torch.cuda.Stream(device=noise.device)
        side, main = self._decode_stream, torch.cuda.current_stream(noise.device)
        pending = None            # (chunk_idx, x0, pixels, done event) of the chunk being decoded on the side stream
        for chunk_idx, start_frame, x0 in chunks:
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                x0.record_stream(side)
                pixels = decode(x0, chunk_idx)
                done = torch.cuda.Event()
                done.record(side)
            if pending is not None:
                pending[3].synchronize()
                yield pending[:3]
            pending = (chunk_idx, x0, pixels, done)
        if pending is not None:
            pending[3].synchronize()
            main.wait_event(pending[3])
            yield pending[:3]

    # ------------------------------------------------------------------------------------------
    def _cache_tokens(self, total_frames: Optional[int] = None) -> int:
        """Cache capacity in tokens.  The reference uses 32760 (= 21 frames x 1560) or
        local_attn_size x 1560 (causal_inference.py:283-288); here: frames x tokens-per-frame of the
        actual latent, never less than what the rollout needs."""
        if self.local_attn_size != -1:
            return self.local_attn_size * self.frame_seq_length
        frames = max(21, total_frames or 0)
        return frames * self.frame_seq_length

    def _initialize_kv_cache(self, batch_size, dtype, device, cache_tokens: Optional[int] = None):
        """Per-GPU KV cache, same dict schema as causal_inference.py:278-298.  The 2 x L index
        tensors are views of one [L, 2] buffer so one fill updates them all."""
        shape = self.generator.model.shape
        if cache_tokens is None:
            cache_tokens = self._cache_tokens()
        n = self.num_transformer_blocks
        index_buffer = torch.zeros(n, 2, dtype=torch.long, device=device)
        kv_cache1 = []
        for i in range(n):
            kv_cache1.append({
                "k": torch.zeros([batch_size, cache_tokens, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
                "v": torch.zeros([batch_size, cache_tokens, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
                "global_end_index": index_buffer[i, 0:1],
                "local_end_index": index_buffer[i, 1:2],
                "_sf_index_buffer": index_buffer,
            })
        kv_cache1[0]["_sf_mirror"] = (kv_cache1[0]["global_end_index"], kv_cache1[0]["local_end_index"], 0, 0)
        # the forward call's last kernel writes all 2 x L indices at once while the dicts still hold THESE views
        kv_cache1[0]["_sf_index_views"] = [(kv["global_end_index"], kv["local_end_index"]) for kv in kv_cache1]
        self.kv_cache1 = kv_cache1

    def _reset_kv_indices(self):
        buf = self.kv_cache1[0].get("_sf_index_buffer")
        if buf is not None:
            buf.zero_()
            d0 = self.kv_cache1[0]
            d0["_sf_mirror"] = (d0["global_end_index"], d0["local_end_index"], 0, 0)
        else:  # foreign cache: rebind as the reference does (causal_inference.py:128-132)
            dev = self.kv_cache1[0]["k"].device
            for kv in self.kv_cache1:
                kv["global_end_index"] = torch.tensor([0], dtype=torch.long, device=dev)
                kv["local_end_index"] = torch.tensor([0], dtype=torch.long, device=dev)

    def _initialize_crossattn_cache(self, batch_size, dtype, device):
        """causal_inference.py:300-312."""
        shape = self.generator.model.shape
        self.crossattn_cache = [{
            "k": torch.zeros([batch_size, shape.text_len, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
            "v": torch.zeros([batch_size, shape.text_len, shape.num_heads, shape.head_dim], dtype=dtype, device=device),
            "is_init": False,
        } for _ in range(self.num_transformer_blocks)]
