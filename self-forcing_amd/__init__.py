"""MI355X-native implementation of Self-Forcing's chunk-wise autoregressive denoising
rollout (the `WanDiffusionWrapper` / `CausalInferencePipeline` hot path).

Host code is Python; all device work is hand-written HIP for gfx950 behind the C-ABI
declared in `include/sf_hip.h` (built into `self-forcing_amd/csrc/libsf_hip.so`).
There is no CPU or eager-PyTorch fallback: the compute entry points raise when the
library has not been built.
"""
from .weights import (WanShape, WAN_1_3B, WAN_14B, WAN_REDUCED, NAMED_SHAPES, synth_state_dict,  # noqa: F401
                      param_shapes, merge_lora, strip_prefix, synth_lora_state_dict, apply_lora_file,
                      load_lora_file, lora_target_linears)
from .kvcache import CachePlan, plan_cache_update  # noqa: F401
from .scheduler import FlowMatchScheduler  # noqa: F401
from .wan_wrapper import WanDiffusionWrapper  # noqa: F401
from .pipeline import CausalInferencePipeline  # noqa: F401
from .harness import SyntheticTextEncoder, FixedTextEncoder, IdentityVAE  # noqa: F401
from .concurrent import RolloutPool  # noqa: F401
from .vae_weights import VaeShape, WAN_VAE, VAE_REDUCED, synth_vae_state_dict, vae_param_shapes  # noqa: F401
from .vae import WanVAEWrapper, WanVAEDecoder, repack_conv  # noqa: F401
from .t5_weights import T5Shape, UMT5_XXL, T5_REDUCED, synth_t5_state_dict, t5_param_shapes  # noqa: F401
from .text_encoder import WanTextEncoder, UMT5Encoder, relative_position_buckets  # noqa: F401
from . import unipc  # noqa: F401
from .diffusion_pipeline import CausalDiffusionInferencePipeline  # noqa: F401
from .unipc import FlowUniPCMultistepScheduler  # noqa: F401
from . import ops, _lib, torch_ops, text_encoder, distributed  # noqa: F401

__all__ = ["WanShape", "WAN_1_3B", "WAN_14B", "WAN_REDUCED", "NAMED_SHAPES", "synth_state_dict",
           "param_shapes", "merge_lora", "strip_prefix", "synth_lora_state_dict", "apply_lora_file", "load_lora_file",
           "lora_target_linears", "CachePlan", "plan_cache_update",
           "FlowMatchScheduler", "WanDiffusionWrapper", "CausalInferencePipeline",
           "SyntheticTextEncoder", "FixedTextEncoder", "IdentityVAE", "RolloutPool", "ops", "torch_ops",
           "VaeShape", "WAN_VAE", "VAE_REDUCED", "synth_vae_state_dict", "vae_param_shapes", "WanVAEWrapper",
           "WanVAEDecoder", "repack_conv", "T5Shape", "UMT5_XXL", "T5_REDUCED", "synth_t5_state_dict", "t5_param_shapes",
           "WanTextEncoder", "UMT5Encoder", "relative_position_buckets", "FlowUniPCMultistepScheduler", "CausalDiffusionInferencePipeline"]
