"""Mel front-end (MelSpec with mel_spec_type="vocos", F/model/modules.py:75-101,104-143) on the device."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


@torch.no_grad()
def mel_spectrogram(wave: torch.Tensor, n_fft=1024, hop_length=256, n_mel_channels=100, target_sample_rate=24000):
    """wave [b, nw] (device fp32) -> log-mel [b, n_mels, 1 + nw // hop]."""
    if wave.ndim == 3:
        wave = wave.squeeze(1)
    assert wave.ndim == 2
    if wave.device.type != "cuda":
        raise _lib.F5HipError("mel_spectrogram needs a HIP device tensor (no CPU fallback)")
    wave = wave.to(torch.float32).contiguous()
    b, nw = wave.shape
    mel = torch.empty(b, n_mel_channels, 1 + nw // hop_length, device=wave.device, dtype=torch.float32)
    _lib.check(_lib.lib().f5hip_mel_spectrogram(b, nw, C.c_void_p(wave.data_ptr()), C.c_void_p(mel.data_ptr()), n_fft,
                                                hop_length, n_mel_channels, target_sample_rate,
                                                _lib.current_stream_ptr()), "f5hip_mel_spectrogram")
    return mel


@torch.no_grad()
def mel_spectrogram_bigvgan(wave: torch.Tensor, n_fft=1024, hop_length=256, n_mel_channels=100, target_sample_rate=24000):
    """get_bigvgan_mel_spectrogram (F/model/modules.py:30-72): wave [b, nw] (device fp32) -> log-mel [b, n_mels, nw // hop]."""
    if wave.ndim == 3:
        wave = wave.squeeze(1)
    assert wave.ndim == 2
    if wave.device.type != "cuda":
        raise _lib.F5HipError("mel_spectrogram_bigvgan needs a HIP device tensor (no CPU fallback)")
    wave = wave.to(torch.float32).contiguous()
    b, nw = wave.shape
    pad = (n_fft - hop_length) // 2
    frames = (nw + 2 * pad - n_fft) // hop_length + 1
    mel = torch.empty(b, n_mel_channels, frames, device=wave.device, dtype=torch.float32)
    _lib.check(_lib.lib().f5hip_mel_spectrogram_bigvgan(b, nw, C.c_void_p(wave.data_ptr()), C.c_void_p(mel.data_ptr()), n_fft,
                                                        hop_length, n_mel_channels, target_sample_rate,
                                                        _lib.current_stream_ptr()), "f5hip_mel_spectrogram_bigvgan")
    return mel
