"""Vocos vocoder object: stands where the reference passes `vocoder` (F/infer/utils_infer.py:92-115,472).

`decode(mel[b, 100, T]) -> wave[b, 256 (T - 1)]` runs in libf5hip; state_dict keys are vocos 0.1.0's
(`backbone.embed.weight`, `backbone.convnext.{i}.*`, `backbone.final_layer_norm.*`, `head.out.*`)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, torch_ops


class F5HipVocos:
    def __init__(self, state_dict: dict, in_channels=100, dim=512, intermediate_dim=1536, num_layers=8, n_fft=1024,
                 hop_length=256, gemm_planes: int = 2, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.F5HipError("F5HipVocos needs a HIP device (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self.hop_length = hop_length
        self._lib = _lib.lib()
        cfg = _lib.VocosConfig(in_channels, dim, intermediate_dim, num_layers, n_fft, hop_length, gemm_planes)
        self._h = self._lib.f5hip_vocos_create(C.byref(cfg))
        if not self._h:
            raise _lib.F5HipError("f5hip_vocos_create: " + self._lib.f5hip_last_error().decode())
        for k, v in state_dict.items():
            if k.startswith("feature_extractor."):
                continue   # mel front-end buffers of the checkpoint; decode() does not use them
            a = np.ascontiguousarray(v.detach().to(torch.float32).cpu().numpy())
            _lib.check(self._lib.f5hip_vocos_load_param(self._h, k.encode(), C.c_void_p(a.ctypes.data), a.size),
                       "vocos load_param " + k)
        _lib.check(self._lib.f5hip_vocos_finalize(self._h), "f5hip_vocos_finalize")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.f5hip_vocos_destroy(h)

    def eval(self):
        return self

    def to(self, device):
        return self

    @torch.no_grad()
    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        b, c, t = mel.shape
        mel = mel.to(self.device, torch.float32).contiguous()
        if torch_ops.load():   # TORCH_LIBRARY operator over the same C entry point
            return torch_ops.ops().vocos_decode(int(self._h), mel, self.hop_length)
        wave = torch.empty(b, self.hop_length * (t - 1), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.f5hip_vocos_decode(self._h, b, t, C.c_void_p(mel.data_ptr()), C.c_void_p(wave.data_ptr()),
                                                _lib.current_stream_ptr()), "f5hip_vocos_decode")
        return wave


class F5HipBigVGAN:
    """BigVGAN v2 generator object: stands where the reference passes `vocoder` for mel_spec_type="bigvgan"
    (F/infer/utils_infer.py:116-129,474): `vocoder(mel[b, 100, T]) -> wave[b, 1, 256 T]`.  state_dict keys are the generator's
    after `remove_weight_norm()`; weight-norm'ed checkpoints (`weight_g` / `weight_v`) are folded here the same way."""

    def __init__(self, state_dict: dict, num_mels=100, upsample_rates=(4, 4, 2, 2, 2, 2), upsample_kernel_sizes=(8, 8, 4, 4, 4, 4),
                 upsample_initial_channel=1536, resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), gemm_planes: int = 2, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.F5HipError("F5HipBigVGAN needs a HIP device (no CPU fallback)")
        torch.cuda.set_device(self.device)
        self.total_up = int(np.prod(upsample_rates))
        self._lib = _lib.lib()
        cfg = _lib.BigVGANConfig()
        cfg.num_mels, cfg.num_upsamples, cfg.upsample_initial_channel, cfg.gemm_planes = num_mels, len(upsample_rates), upsample_initial_channel, gemm_planes
        for i, (r, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = r, k
        for j, k in enumerate(resblock_kernel_sizes):
            cfg.resblock_kernel_sizes[j] = k
            for d, dil in enumerate(resblock_dilation_sizes[j]):
                cfg.resblock_dilations[j * 3 + d] = dil
        self._h = self._lib.f5hip_bigvgan_create(C.byref(cfg))
        if not self._h:
            raise _lib.F5HipError("f5hip_bigvgan_create: " + self._lib.f5hip_last_error().decode())
        sd = dict(state_dict)
        for k in [k for k in sd if k.endswith(".weight_g")]:   # fold weight norm: w = g * v / ||v|| over all dims but 0
            base = k[: -len("_g")]
            g, v = sd.pop(k), sd.pop(base + "_v")
            sd[base] = v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.ndim - 1))))
        for k, v in sd.items():
            if k.endswith(".filter"):
                continue   # anti-aliasing FIR buffers: recomputed in the library
            a = np.ascontiguousarray(v.detach().to(torch.float32).cpu().numpy())
            _lib.check(self._lib.f5hip_bigvgan_load_param(self._h, k.encode(), C.c_void_p(a.ctypes.data), a.size), "bigvgan load_param " + k)
        _lib.check(self._lib.f5hip_bigvgan_finalize(self._h), "f5hip_bigvgan_finalize")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.f5hip_bigvgan_destroy(h)

    def eval(self):
        return self

    def to(self, device):
        return self

    def remove_weight_norm(self):
        return None

    @torch.no_grad()
    def __call__(self, mel: torch.Tensor) -> torch.Tensor:
        b, c, t = mel.shape
        mel = mel.to(self.device, torch.float32).contiguous()
        if torch_ops.load():
            return torch_ops.ops().bigvgan_forward(int(self._h), mel, self.total_up)
        wave = torch.empty(b, 1, self.total_up * t, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.f5hip_bigvgan_forward(self._h, b, t, C.c_void_p(mel.data_ptr()), C.c_void_p(wave.data_ptr()),
                                                   _lib.current_stream_ptr()), "f5hip_bigvgan_forward")
        return wave
