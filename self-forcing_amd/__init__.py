"""MI355X-native implementation of Self-Forcing's chunk-wise autoregressive denoising
rollout (the `WanDiffusionWrapper` / `CausalInferencePipeline` hot path).

Host code is Python; all device work is hand-written HIP for gfx950 behind the C-ABI
declared in `include/sf_hip.h` (built into `self-forcing_amd/csrc/libsf_hip.so`).
There is no CPU or eager-PyTorch fallback: importing the compute entry points without
the built library raises.
"""
from .weights import WanShape, WAN_1_3B, WAN_14B, WAN_REDUCED, NAMED_SHAPES, synth_state_dict  # noqa: F401

__all__ = ["WanShape", "WAN_1_3B", "WAN_14B", "WAN_REDUCED", "NAMED_SHAPES", "synth_state_dict"]
