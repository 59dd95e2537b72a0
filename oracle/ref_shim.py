"""Import the reference's hot-path modules on CPU (build container only).

TEST INFRASTRUCTURE.  Used by the fixture generators `oracle/make_golden*.py` only (the tests
compare against the committed fixtures; /root/reference does not exist on the GPU box).
Nothing from the reference is copied: this file only arranges for
`import` to succeed despite packages that are not installed here.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SF_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "wan", "modules"))


_loaded = None


def load():
    """Returns a namespace with the reference classes of the hot path."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    import torch

    sys.dont_write_bytecode = True  # never write __pycache__ into the reference tree
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    def stub(name, **attrs):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m

    def pkg(name, path):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m

    class ModelMixin(torch.nn.Module):
        pass

    # diffusers is used only for ModelMixin / ConfigMixin plumbing on this path
    stub("diffusers")
    stub("diffusers.models")
    stub("diffusers.configuration_utils", ConfigMixin=type("ConfigMixin", (), {}),
         register_to_config=lambda f: f)
    stub("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    stub("ftfy", fix_text=lambda s: s)
    # skip the eager package __init__ files (they import T5/CLIP/VAE + absent deps)
    pkg("wan", os.path.join(REFERENCE_ROOT, "wan"))
    pkg("wan.modules", os.path.join(REFERENCE_ROOT, "wan", "modules"))
    pkg("pipeline", os.path.join(REFERENCE_ROOT, "pipeline"))
    # evaluated at import time in demo_utils/memory.py:9 and wan/modules/t5.py:478
    torch.cuda.current_device = lambda: 0

    import wan.modules.attention as wa
    import wan.modules.model as wm
    import wan.modules.causal_model as wcm
    from utils.wan_wrapper import WanDiffusionWrapper
    from utils.scheduler import FlowMatchScheduler
    from pipeline.causal_inference import CausalInferencePipeline

    ns = types.SimpleNamespace(
        attention_mod=wa, model_mod=wm, causal_mod=wcm,
        CausalWanModel=wcm.CausalWanModel, WanDiffusionWrapper=WanDiffusionWrapper,
        FlowMatchScheduler=FlowMatchScheduler, CausalInferencePipeline=CausalInferencePipeline,
    )
    _orig_attention = wa.attention

    def set_attention_dtype(mode: str):
        """'bf16': the reference's CPU path as shipped (SDPA fallback casts q/k/v to bf16,
        attention.py:194-196).  'input': additionally pass dtype=q.dtype so an fp32 model
        runs (the fp32 'math oracle' of SURVEY 8c item 7)."""
        if mode == "bf16":
            fn = _orig_attention
        else:
            def fn(q, k, v, *a, **kw):
                kw["dtype"] = q.dtype
                return _orig_attention(q, k, v, *a, **kw)
        # flash_attention asserts q.device.type == 'cuda' before reaching its own
        # fallback (attention.py:62, :68-82); rebind both call sites to that fallback.
        wm.flash_attention = fn
        wcm.attention = fn

    ns.set_attention_dtype = set_attention_dtype
    set_attention_dtype("bf16")
    _loaded = ns
    return ns


def build_wrapper(ns, model, shift: float):
    """Assemble a WanDiffusionWrapper around an already-constructed CausalWanModel
    (its __init__ calls from_pretrained on a local directory that does not exist here,
    utils/wan_wrapper.py:137-145)."""
    import torch
    w = ns.WanDiffusionWrapper.__new__(ns.WanDiffusionWrapper)
    torch.nn.Module.__init__(w)
    w.model = model
    w.uniform_timestep = False
    w.scheduler = ns.FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
    w.scheduler.set_timesteps(1000, training=True)
    w.seq_len = 32760
    w.post_init()
    return w


def load_sampler():
    """Additionally import the 50-step classifier-free-guidance sampler: `wan/utils/fm_solvers_unipc.py` and
    `pipeline/causal_diffusion_inference.py`.

    Both inherit from diffusers' `SchedulerMixin` / `ConfigMixin`, which is not installed here.  The stand-ins below
    carry NO numerical content: `register_to_config` records the constructor's keyword arguments (with defaults) as
    `self.config.<name>`, which is all the scheduler reads from the mixins; `deprecate` is a no-op; the scheduler's
    arithmetic comes entirely from the reference's own file.  `wan.modules.clip` (needs torchvision) is replaced by
    a placeholder because the pipeline only instantiates it when no image encoder is injected."""
    ns = load()
    if hasattr(ns, "CausalDiffusionInferencePipeline"):
        return ns
    import dataclasses
    import enum
    import functools
    import inspect
    import torch

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *args, **kwargs):
            bound = inspect.signature(init).bind(self, *args, **kwargs)
            bound.apply_defaults()
            self.config = types.SimpleNamespace(**{k: v for k, v in bound.arguments.items() if k != "self"})
            init(self, *args, **kwargs)
        return inner

    class ConfigMixin:
        def register_to_config(self, **kw):
            self.config.__dict__.update(kw)

    class SchedulerMixin:
        pass

    @dataclasses.dataclass
    class SchedulerOutput:
        prev_sample: torch.Tensor

    class KarrasDiffusionSchedulers(enum.Enum):
        UniPCMultistepScheduler = 1

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.schedulers")
    mod("diffusers.schedulers.scheduling_utils", KarrasDiffusionSchedulers=KarrasDiffusionSchedulers,
        SchedulerMixin=SchedulerMixin, SchedulerOutput=SchedulerOutput)
    mod("diffusers.utils", deprecate=lambda *a, **k: None, is_scipy_available=lambda: False)
    mod("diffusers.utils.torch_utils", randn_tensor=lambda shape, generator=None, device=None, dtype=None:
        torch.randn(shape, generator=generator, device=device, dtype=dtype))
    mod("wan.modules.clip", CLIPModel=type("CLIPModel", (), {}))
    if "wan.utils" not in sys.modules:
        m = types.ModuleType("wan.utils")
        m.__path__ = [os.path.join(REFERENCE_ROOT, "wan", "utils")]
        sys.modules["wan.utils"] = m
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from pipeline.causal_diffusion_inference import CausalDiffusionInferencePipeline
    ns.FlowUniPCMultistepScheduler = FlowUniPCMultistepScheduler
    ns.CausalDiffusionInferencePipeline = CausalDiffusionInferencePipeline
    return ns
