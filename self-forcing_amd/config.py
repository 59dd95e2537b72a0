"""Config loading: the reference merges `configs/default_config.yaml` under a run config with
OmegaConf (inference.py:57-59).  The keys the hot path consumes are few and flat
(`denoising_step_list, warp_denoising_step, num_frame_per_block, independent_first_frame,
context_noise, model_kwargs`), so `yaml.safe_load` + a recursive dict merge is enough; the result is
a namespace with attribute access like the OmegaConf object the pipeline expects."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, Optional

import yaml

# the hot-path subset of configs/default_config.yaml:1-25
DEFAULTS: Dict[str, Any] = {
    "independent_first_frame": False,
    "warp_denoising_step": False,
    "context_noise": 0,
    "num_frame_per_block": 1,
    "model_kwargs": {},
    "seed": 0,
    "num_samples": 1,
}


def _merge(base: Dict[str, Any], over: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = v
    return out


class Config(SimpleNamespace):
    """Attribute access + `in` / `get`, so both `hasattr(cfg, 'denoising_step_list')`
    (inference.py:62) and `getattr(args, 'model_kwargs', {})` (causal_inference.py:21) work."""

    def __contains__(self, key):
        return key in self.__dict__

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def to_dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)


def load_config(path: Optional[str] = None, default_path: Optional[str] = None, overrides: Optional[Dict[str, Any]] = None) -> Config:
    """default (file or built-in) <- run config <- overrides, later wins (OmegaConf.merge order)."""
    cfg = dict(DEFAULTS)
    for p in (default_path, path):
        if p:
            with open(p, encoding="utf-8") as f:
                data = yaml.safe_load(f) or {}
            if not isinstance(data, dict):
                raise ValueError(f"{p}: expected a mapping at the top level")
            cfg = _merge(cfg, data)
    if overrides:
        cfg = _merge(cfg, overrides)
    return Config(**cfg)


def is_few_step(cfg: Config) -> bool:
    """Pipeline selection of the reference: few-step iff `denoising_step_list` is present
    (inference.py:62-67); otherwise the multi-step CFG sampler (`CausalDiffusionInferencePipeline`)."""
    return "denoising_step_list" in cfg
