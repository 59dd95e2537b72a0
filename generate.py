#!/usr/bin/env python
"""Batch generation driver for the hot path (counterpart of the reference's inference.py:39-196,
written fresh: the fork's own CLI passes keyword arguments `CausalInferencePipeline.inference`
does not accept).

    python generate.py --config_path cfg.yaml --data_path prompts.txt --output_folder out [--random_init_seed 0]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 generate.py ...

Keeps the reference's semantics: OmegaConf-style merge of a default config under the run config,
`DistributedSampler(shuffle=False, drop_last=True)` prompt sharding (rank r takes r, r+W, ...),
seed = `--seed + rank`, noise `[num_samples, num_output_frames, 16, 60, 104]` bf16 drawn per prompt,
one barrier after set-up.  It writes LATENTS (`<idx>-<sample>.pt`) and, when a VAE is given
(`--vae_path Wan2.1_VAE.pth`, loaded with weights_only=True, or `--vae_random_init_seed N`), the decoded
video as a uint8 tensor [T, H, W, 3] (`<idx>-<sample>.video.pt`: what the reference hands to
`write_video`, inference.py:186-196; there is no video encoder in this image).  The umT5 encoder is
outside this path, so embeddings are synthetic unless `--prompt_embeds` (a .pt dict prompt -> [L, 4096]
tensor) is given.  A config WITHOUT `denoising_step_list` selects the multi-step classifier-free-guidance sampler
(`CausalDiffusionInferencePipeline`), as inference.py:62-67 does; it needs `num_train_timestep`, `timestep_shift`,
`guidance_scale` and `negative_prompt`.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import self_forcing_amd as sfa  # noqa: E402
from self_forcing_amd.config import is_few_step, load_config  # noqa: E402
from self_forcing_amd.sharding import read_prompts, shard_indices  # noqa: E402


class TableTextEncoder:
    """prompt -> precomputed embedding (zero padded to text_len)."""

    def __init__(self, table, text_len, text_dim, device):
        self.table, self.text_len, self.text_dim, self.device = table, text_len, text_dim, device

    def __call__(self, text_prompts):
        out = torch.zeros(len(text_prompts), self.text_len, self.text_dim, dtype=torch.bfloat16)
        for i, p in enumerate(text_prompts):
            e = self.table[p].to(torch.bfloat16)
            out[i, :e.shape[0]] = e[:self.text_len]
        return {"prompt_embeds": out.to(self.device)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", required=True)
    ap.add_argument("--default_config_path", default=None)
    ap.add_argument("--checkpoint_path", default=None, help=".pt with 'generator' / 'generator_ema' state dicts, or a .safetensors file")
    ap.add_argument("--use_ema", action="store_true")
    ap.add_argument("--random_init_seed", type=int, default=None, help="seeded random weights instead of a checkpoint")
    ap.add_argument("--data_path", required=True, help="one prompt per line")
    ap.add_argument("--eval_first_n", type=int, default=0)
    ap.add_argument("--prompt_embeds", default=None)
    ap.add_argument("--output_folder", required=True)
    ap.add_argument("--num_output_frames", type=int, default=21)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--num_samples", type=int, default=1)
    ap.add_argument("--latent_height", type=int, default=60)
    ap.add_argument("--latent_width", type=int, default=104)
    ap.add_argument("--vae_path", default=None, help="Wan2.1_VAE.pth: decode the latents to pixels")
    ap.add_argument("--sampling_steps", type=int, default=0, help="multi-step sampler only: override its 50 steps")
    ap.add_argument("--vae_random_init_seed", type=int, default=None, help="seeded random VAE decoder weights instead")
    a = ap.parse_args()

    # inference.py:39-45: one process per GPU under torch.distributed.run, RCCL for the start / end barriers only
    from self_forcing_amd.distributed import RankGroup, env_rank_world
    rank, local_rank, world = env_rank_world()
    torch.cuda.set_device(local_rank)
    grp = RankGroup(backend="nccl", device=torch.device(f"cuda:{local_rank}"))
    device = torch.device(f"cuda:{local_rank}")
    torch.manual_seed(a.seed + rank)
    torch.set_grad_enabled(False)

    cfg = load_config(a.config_path, a.default_config_path)
    kwargs = dict(cfg.get("model_kwargs") or {})
    if a.checkpoint_path:
        if a.checkpoint_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(a.checkpoint_path)
        else:
            ck = torch.load(a.checkpoint_path, map_location="cpu", weights_only=True)
            sd = ck["generator_ema" if a.use_ema else "generator"] if "generator" in ck or "generator_ema" in ck else ck
        gen = sfa.WanDiffusionWrapper(**kwargs, is_causal=True, state_dict=sd, device=device)
    else:
        gen = sfa.WanDiffusionWrapper(**kwargs, is_causal=True, random_init_seed=a.random_init_seed, device=device)
    shape = gen.model.shape
    prompts = read_prompts(a.data_path, a.eval_first_n)
    if a.prompt_embeds:
        enc = TableTextEncoder(torch.load(a.prompt_embeds, map_location="cpu", weights_only=True), shape.text_len, shape.text_dim, device)
    else:
        enc = sfa.SyntheticTextEncoder(shape.text_len, shape.text_dim, device=device)
    vae = sfa.IdentityVAE()
    if a.vae_path:
        vae = sfa.WanVAEWrapper(torch.load(a.vae_path, map_location="cpu", weights_only=True), device=device)
    elif a.vae_random_init_seed is not None:
        vae = sfa.WanVAEWrapper(sfa.synth_vae_state_dict(sfa.WAN_VAE, seed=a.vae_random_init_seed), device=device)
    decode = not isinstance(vae, sfa.IdentityVAE)
    few_step = is_few_step(cfg)        # inference.py:62-67: few-step rollout iff the config has denoising_step_list
    if few_step:
        pipe = sfa.CausalInferencePipeline(cfg, device, generator=gen, text_encoder=enc, vae=vae)
    else:                              # 50-step UniPC sampler with classifier-free guidance
        for key in ("num_train_timestep", "timestep_shift", "guidance_scale", "negative_prompt"):
            if key not in cfg:
                raise SystemExit(f"config has neither denoising_step_list nor {key}: cannot build a sampler from it")
        pipe = sfa.CausalDiffusionInferencePipeline(cfg, device, generator=gen, text_encoder=enc, vae=vae)
        if a.sampling_steps:
            pipe.sampling_steps = a.sampling_steps

    if rank == 0:
        os.makedirs(a.output_folder, exist_ok=True)
    grp.barrier()

    for idx in shard_indices(len(prompts), rank, world):
        noise = torch.randn([a.num_samples, a.num_output_frames, 16, a.latent_height, a.latent_width], device=device,
                            dtype=torch.bfloat16)
        if few_step:
            video, latents = pipe.inference(noise=noise, text_prompts=[prompts[idx]] * a.num_samples, return_latents=True)
        else:
            video, latents = pipe.inference(noise, [prompts[idx]] * a.num_samples, None, None, None, return_latents=True)
        for s in range(a.num_samples):
            torch.save(latents[s].cpu(), os.path.join(a.output_folder, f"{idx}-{s}.pt"))
            if decode:   # [T, 3, H, W] in [0, 1] -> [T, H, W, 3] uint8 (inference.py:186-187)
                torch.save((255.0 * video[s].permute(0, 2, 3, 1)).to(torch.uint8).cpu(), os.path.join(a.output_folder, f"{idx}-{s}.video.pt"))
        if rank == 0:
            print(f"[generate] prompt {idx}: latents {tuple(latents.shape)}" + (f", video {tuple(video.shape)}" if decode else ""), flush=True)
    grp.finish()


if __name__ == "__main__":
    main()
