"""Wan VAE decode on the GPU behind the reference's `WanVAEWrapper` surface.

Mirrors `utils/wan_wrapper.py:56-117` for the decode direction only:

    vae = WanVAEWrapper(state_dict=..., device="cuda")
    video = vae.decode_to_pixel(latent, use_cache=False)      # [B, F, 16, h, w] -> [B, 1+4(F-1), 3, 8h, 8w]
    vae.model.clear_cache()                                    # inference.py:183

Every kernel is in csrc/ (conv_igemm.hip, vae_elementwise.hip, gemm_bf16.hip) and one latent frame is ONE
C call (`sf_vae_decode_frames`).  There is no eager/CPU fallback.  `encode_to_latent` (used only by
`--i2v`, inference.py:145) is not on this path and raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib, torch_ops
from .vae_weights import (LATENT_MEAN, LATENT_STD, ResBlockSpec, ResampleSpec, VaeShape, WAN_VAE, decoder_layout,
                          vae_param_shapes)

Tensor = torch.Tensor


def repack_conv(w: Tensor, cin_pad: Optional[int] = None) -> Tensor:
    """Conv3d / Conv2d weight [Cout, Cin, (kt,) kh, kw] -> the implicit-GEMM layout of `sf_conv_args.w`:
    [Cout][Kpad] with k = ((dt*kh + dh)*kw + dw)*Cin_pad + ci, Cin padded to a multiple of 32 and K to a
    multiple of 64 (zeros)."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kt, kh, kw = w.shape
    cp = cin_pad or ((cin + 31) // 32) * 32
    t = torch.zeros(cout, kt, kh, kw, cp, dtype=w.dtype, device=w.device)
    t[..., :cin] = w.permute(0, 2, 3, 4, 1)
    k = kt * kh * kw * cp
    kpad = ((k + 63) // 64) * 64
    out = torch.zeros(cout, kpad, dtype=w.dtype, device=w.device)
    out[:, :k] = t.reshape(cout, k)
    return out


class _StreamPos:
    """Where one latent size's stream stands since its last reset."""
    __slots__ = ("fresh", "nframes", "slot")

    def __init__(self):
        self.fresh = True      # no chunk decoded since the last clear_cache
        self.nframes = 0       # latent frames decoded since then
        self.slot = 0          # ... of them in the current lap of the sliding history windows (sf_vae_decode_frames)


class WanVAEDecoder:
    """Device-resident decoder: repacked bf16 weights, the C model descriptor, and one decode state
    (the convolution histories of one stream).  Counterpart of `WanVAE_` (wan/modules/vae.py:478-617)
    for `decode` / `cached_decode` / `clear_cache`."""

    def __init__(self, shape: VaeShape, state_dict: Dict[str, Tensor], device, frames_per_call: int = 4):
        if not 1 <= frames_per_call <= 63:
            raise ValueError("frames_per_call must be in 1..63")
        self.shape = shape
        self.device = torch.device(device)
        self.frames_per_call = frames_per_call          # latent frames handed to one C call (any value gives the same bits)
        self.window_frames = frames_per_call + 1        # slots of the sliding history windows (the first chunk + one group)
        self._keep: List[Tensor] = []
        self._state: Dict[tuple, Tensor] = {}
        self._scratch: Dict[tuple, Tensor] = {}
        self._pos: Dict[tuple, _StreamPos] = {}     # per latent size, beside its state: where that stream stands
        self._load(state_dict)

    # ---------------------------------------------------------------------------------
    def _dev(self, t: Tensor, dtype=torch.bfloat16) -> Tensor:
        t = t.detach().to(device=self.device, dtype=dtype).contiguous()
        self._keep.append(t)
        return t

    def _conv(self, dst: _lib.VaeConv, sd, name: str, cin_pad: Optional[int] = None) -> None:
        w = sd[name + ".weight"]
        if w.dim() == 4:
            w = w.unsqueeze(2)
        cout, cin, kt, kh, kw = w.shape
        rp = self._dev(repack_conv(w.float(), cin_pad))
        dst.w, dst.bias = rp.data_ptr(), self._dev(sd[name + ".bias"]).data_ptr()
        dst.cin = cin_pad or ((cin + 31) // 32) * 32
        dst.cout, dst.kt, dst.kh, dst.kw, dst.ldw = cout, kt, kh, kw, rp.shape[1]

    def _res(self, dst: _lib.VaeResBlock, sd, spec: ResBlockSpec) -> None:
        p = spec.prefix
        dst.gamma1 = self._dev(sd[p + "residual.0.gamma"].flatten()).data_ptr()
        dst.gamma2 = self._dev(sd[p + "residual.3.gamma"].flatten()).data_ptr()
        self._conv(dst.conv1, sd, p + "residual.2")
        self._conv(dst.conv2, sd, p + "residual.6")
        if spec.in_dim != spec.out_dim:
            self._conv(dst.shortcut, sd, p + "shortcut")

    def _load(self, sd: Dict[str, Tensor]) -> None:
        s = self.shape
        need = vae_param_shapes(s)
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"VAE state dict lacks {len(missing)} decoder tensors, e.g. {missing[:4]}")
        for k, shp in need.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
        if any(d % 32 for d in s.dims) or s.dims[0] % 64:
            raise ValueError(f"decoder widths {s.dims} must be multiples of 32 (first: 64)")
        if len(s.dim_mult) > _lib.VAE_MAX_STAGES:
            raise ValueError("at most 4 decoder stages")
        m = _lib.VaeModel()
        m.z_dim, m.n_stages, m.res_per_stage = s.z_dim, len(s.dim_mult), s.num_res_blocks + 1
        for i, t in enumerate(s.temperal_upsample):
            m.temporal_up[i] = 1 if t else 0
        m.latent_mean = self._dev(torch.tensor(LATENT_MEAN[:s.z_dim]), torch.float32).data_ptr()
        m.latent_std = self._dev(torch.tensor(LATENT_STD[:s.z_dim]), torch.float32).data_ptr()
        m.conv2_w = self._dev(sd["conv2.weight"].reshape(s.z_dim, s.z_dim)).data_ptr()
        m.conv2_b = self._dev(sd["conv2.bias"]).data_ptr()
        self._conv(m.conv1, sd, "decoder.conv1")
        middle, ups = decoder_layout(s)
        self._res(m.mid0, sd, middle[0])
        self._res(m.mid2, sd, middle[2])
        a, c = middle[1], s.dims[0]
        qkv_w, qkv_b = sd[a + "to_qkv.weight"].reshape(3 * c, c), sd[a + "to_qkv.bias"]
        m.attn_gamma = self._dev(sd[a + "norm.gamma"].flatten()).data_ptr()
        m.attn_qk_w, m.attn_qk_b = self._dev(qkv_w[:2 * c]).data_ptr(), self._dev(qkv_b[:2 * c]).data_ptr()
        m.attn_v_w, m.attn_v_b = self._dev(qkv_w[2 * c:]).data_ptr(), self._dev(qkv_b[2 * c:]).data_ptr()
        m.attn_proj_w = self._dev(sd[a + "proj.weight"].reshape(c, c)).data_ptr()
        m.attn_proj_b = self._dev(sd[a + "proj.bias"]).data_ptr()
        blocks = [u for u in ups if isinstance(u, ResBlockSpec)]
        res = (_lib.VaeResBlock * len(blocks))()
        for i, spec in enumerate(blocks):
            self._res(res[i], sd, spec)
        self._res_array = res
        m.res_host = C.cast(res, C.POINTER(_lib.VaeResBlock))
        for stage, spec in enumerate(u for u in ups if isinstance(u, ResampleSpec)):
            self._conv(m.up_conv[stage], sd, spec.prefix + "resample.1")
            if spec.mode == "upsample3d":
                self._conv(m.time_conv[stage], sd, spec.prefix + "time_conv")
        m.head_gamma = self._dev(sd["decoder.head.0.gamma"].flatten()).data_ptr()
        self._conv(m.head_conv, sd, "decoder.head.2")
        self.cmodel = m
        self._handle = torch_ops.register_model(self)

    def param_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)

    # ---------------------------------------------------------------------------------
    def _buffers(self, h: int, w: int):
        key = (h, w)
        sf, tf = self.shape.spatial_factor, self.shape.temporal_factor
        if (2 + self.window_frames * tf) * (sf * h) * (sf * w) * self.cmodel.head_conv.cin * 2 >= 0xFFFFFF00:
            raise ValueError(f"frames_per_call={self.frames_per_call} at {sf * h}x{sf * w}: a convolution's input volume would pass 4 GiB "
                             "(the kernels address a volume through one 32-bit-ranged buffer descriptor); use fewer frames per call")
        if key not in self._state:
            n = _lib.lib().sf_vae_state_bytes(C.byref(self.cmodel), h, w, self.window_frames)
            if n == 0:
                _lib.check(-1, "sf_vae_state_bytes")
            self._state[key] = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._pos[key] = _StreamPos()          # every latent size streams on its own: counters live beside its state
        skey = (h, w, torch.cuda.current_stream(self.device).cuda_stream)
        if skey not in self._scratch:
            n = _lib.lib().sf_vae_scratch_bytes(C.byref(self.cmodel), h, w, self.window_frames)
            self._scratch[skey] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._state[key], self._scratch[skey]

    def clear_cache(self) -> None:
        """`WanVAE_.clear_cache` (vae.py:610-617): forget every convolution's history."""
        stream = torch.cuda.current_stream(self.device).cuda_stream if self._state else None
        for (h, w), st in self._state.items():
            _lib.check(_lib.lib().sf_vae_reset(C.byref(self.cmodel), st.data_ptr(), st.numel(), h, w, self.window_frames, stream), "sf_vae_reset")
            self._pos[(h, w)] = _StreamPos()

    def frames_out(self, latent_frames: int, h: Optional[int] = None, w: Optional[int] = None) -> int:
        """Pixel frames the next `cached_decode` of `latent_frames` frames of an h x w latent returns (the size may be
        omitted while the decoder has seen at most one)."""
        tf = self.shape.temporal_factor
        if h is None:
            if len(self._pos) > 1:
                raise ValueError("frames_out: this decoder streams several latent sizes; pass h and w")
            pos = next(iter(self._pos.values()), None)
        else:
            pos = self._pos.get((h, w))
        fresh = pos is None or pos.fresh
        return (1 + tf * (latent_frames - 1)) if fresh else tf * latent_frames

    def cached_decode(self, z: Tensor) -> Tensor:
        """`WanVAE_.cached_decode` (vae.py:579-593) for one sample: z [F, z_dim, h, w] bf16 -> float32
        pixels [T, 3, 8h, 8w] in [-1, 1]; continues from the state the previous call left."""
        if z.dim() != 4 or z.shape[1] != self.shape.z_dim:
            raise ValueError(f"expected latents [F, {self.shape.z_dim}, h, w], got {tuple(z.shape)}")
        z = z.to(device=self.device, dtype=torch.bfloat16).contiguous()
        F, _, h, w = z.shape
        state, scratch = self._buffers(h, w)
        pos = self._pos[(h, w)]
        sf, tf = self.shape.spatial_factor, self.shape.temporal_factor
        out = torch.empty(self.frames_out(F, h, w), 3, sf * h, sf * w, dtype=torch.float32, device=self.device)
        t0 = i = 0
        while i < F:
            # the frame that follows a reset is decoded alone (one output frame); after it, groups of up to
            # frames_per_call latent frames per C call -- bit-identical to one call per frame, but the low-resolution
            # stages fill the chip and every launch has a shorter tail
            g = 1 if pos.fresh else min(self.frames_per_call, F - i)
            if pos.slot + g > self.window_frames:
                window, history_at = 0, pos.slot
            else:
                window = history_at = pos.slot
            torch.ops.sf_hip.vae_decode_frames(self._handle, state, scratch, z[i:i + g], out[t0:], h, w, self.window_frames,
                                               pos.nframes, window, history_at)
            pos.slot = window + g
            pos.nframes += g
            t0 += 1 if pos.fresh else tf * g
            i += g
            pos.fresh = False
        return out

    def decode(self, z: Tensor) -> Tensor:
        """`WanVAE_.decode` (vae.py:556-578): cleared caches before and after."""
        self.clear_cache()
        out = self.cached_decode(z)
        self.clear_cache()
        return out


VAE_CHECKPOINT = "wan_models/Wan2.1-T2V-1.3B/Wan2.1_VAE.pth"   # utils/wan_wrapper.py:74


class WanVAEWrapper(torch.nn.Module):
    """Drop-in for the reference's `WanVAEWrapper` (utils/wan_wrapper.py:56-117), decode side.

    `state_dict`: the tensors of `Wan2.1_VAE.pth` (or the seeded stand-in of `vae_weights.synth_vae_state_dict`);
    encoder tensors, if present, are ignored.  Without one the reference's default checkpoint path is loaded
    (weights-only); FileNotFoundError when it is absent."""

    def __init__(self, state_dict: Optional[Dict[str, Tensor]] = None, device="cuda", shape: VaeShape = WAN_VAE,
                 checkpoint_path: str = VAE_CHECKPOINT, frames_per_call: int = 4):
        super().__init__()
        if state_dict is None:   # `WanVAEWrapper()` as the reference's pipelines construct it (wan_wrapper.py:72-76)
            import os
            if not os.path.exists(checkpoint_path):
                raise FileNotFoundError(f"VAE checkpoint {checkpoint_path!r} not found: download it as the reference's README "
                                        "describes, or construct WanVAEWrapper(state_dict=...) / inject a vae= into the pipeline")
            state_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        self.mean = torch.tensor(LATENT_MEAN, dtype=torch.float32)
        self.std = torch.tensor(LATENT_STD, dtype=torch.float32)
        self.model = WanVAEDecoder(shape, state_dict, device, frames_per_call=frames_per_call)

    def encode_to_latent(self, pixel: Tensor) -> Tensor:
        raise NotImplementedError("the VAE encoder (image-to-video conditioning, inference.py:145) is outside this path")

    def decode_to_pixel(self, latent: Tensor, use_cache: bool = False) -> Tensor:
        """latent [B, F, C, h, w] -> float32 [B, T, 3, 8h, 8w] clamped to [-1, 1] (wan_wrapper.py:95-117)."""
        if use_cache:
            assert latent.shape[0] == 1, "Batch size must be 1 when using cache"
        fn = self.model.cached_decode if use_cache else self.model.decode
        return torch.stack([fn(u) for u in latent], dim=0)

    def decode_chunk(self, latent: Tensor, chunk_index: int) -> Tensor:
        """Streaming decode used by `CausalInferencePipeline.stream`: chunk 0 starts from cleared caches,
        later chunks continue the stream (demo.py:399-427 does the same with `use_cache=True`)."""
        if chunk_index == 0:
            self.model.clear_cache()
        return self.decode_to_pixel(latent, use_cache=True)
