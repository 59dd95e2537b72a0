"""Synthetic stand-ins for the two components either side of the DiT rollout -- the umT5 text encoder and
the Wan VAE -- for benchmarks and tests that time / check the rollout alone (BASELINE.json's timed region ends at
the last latent).  The real ones are `self_forcing_amd.WanTextEncoder` (text_encoder.py) and
`self_forcing_amd.WanVAEWrapper` (vae.py); the pipeline constructors accept either kind injected, as the
reference's do (pipeline/causal_inference.py:14-23)."""
from __future__ import annotations

import zlib
from typing import List

import torch


class SyntheticTextEncoder:
    """prompt -> deterministic pseudo-embedding [B, text_len, text_dim]: N(0,1) rows seeded by the
    prompt text, rows >= len_i zeroed with len_i in [20, 200] (mirrors the zero padding of
    WanTextEncoder, utils/wan_wrapper.py:50-51)."""

    def __init__(self, text_len: int = 512, text_dim: int = 4096, device="cuda", dtype=torch.bfloat16):
        self.text_len, self.text_dim, self.device, self.dtype = text_len, text_dim, device, dtype
        self._cache = {}

    def embed_one(self, prompt: str) -> torch.Tensor:
        seed = zlib.crc32(prompt.encode("utf-8"))
        g = torch.Generator(device="cpu").manual_seed(seed)
        n = 20 + seed % 181
        e = torch.zeros(self.text_len, self.text_dim)
        e[:n] = torch.randn(n, self.text_dim, generator=g)
        return e

    def __call__(self, text_prompts: List[str]) -> dict:
        key = tuple(text_prompts)
        if key not in self._cache:  # embeddings stay resident on the device, like a real encoder's output
            pe = torch.stack([self.embed_one(p) for p in text_prompts]).to(self.dtype)
            self._cache[key] = pe.to(self.device)
        return {"prompt_embeds": self._cache[key]}


class FixedTextEncoder:
    """Returns a given embedding tensor (tests)."""

    def __init__(self, prompt_embeds: torch.Tensor):
        self.prompt_embeds = prompt_embeds

    def __call__(self, text_prompts: List[str]) -> dict:
        return {"prompt_embeds": self.prompt_embeds}


class IdentityVAE:
    """decode_to_pixel = identity on the latents (VAE decode is SURVEY 8f next-row 1)."""

    def decode_to_pixel(self, latents: torch.Tensor, use_cache: bool = False) -> torch.Tensor:
        return latents
