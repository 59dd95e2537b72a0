"""Flow-matching scheduler of the hot path (utils/scheduler.py:106-176 of the reference).

Only what `CausalInferencePipeline` touches: the shifted sigma / timestep tables, the nearest-entry
sigma lookup and `add_noise`.  Tables are built on the host exactly as the reference builds them
(float32 linspace, shift formula) and mirrored once on the device -- the reference re-uploads them
on every call (scheduler.py:170-171, wan_wrapper.py:218-222).
"""
from __future__ import annotations

import torch

from . import ops


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.sigma_max = sigma_max
        self.sigma_min = sigma_min
        self.inverse_timesteps = inverse_timesteps
        self.extra_one_step = extra_one_step
        self.reverse_sigmas = reverse_sigmas
        self._dev = {}
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False):
        """scheduler.py:118-141 (the training-weight table is not needed on this path)."""
        sigma_start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        if self.extra_one_step:
            sig = torch.linspace(sigma_start, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            sig = torch.linspace(sigma_start, self.sigma_min, num_inference_steps)
        if self.inverse_timesteps:
            sig = torch.flip(sig, dims=[0])
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        if self.reverse_sigmas:
            sig = 1 - sig
        self.sigmas = sig
        self.timesteps = sig * self.num_train_timesteps
        self._dev = {}

    def device_tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.sigmas.to(device=device, dtype=torch.float32).contiguous(),
                              self.timesteps.to(device=device, dtype=torch.float32).contiguous())
        return self._dev[key]

    def add_noise(self, original_samples, noise, timestep):
        """(1 - sigma) x0 + sigma eps with sigma by nearest-timestep lookup (scheduler.py:159-176),
        evaluated by the HIP kernel in fp32 and rounded to noise's dtype (bf16)."""
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        sig, ts = self.device_tables(noise.device)
        return ops.add_noise(original_samples, noise, timestep.to(noise.device), sig, ts)
