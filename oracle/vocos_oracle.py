"""fp32 CPU oracle of the Vocos decoder and the torchaudio-style mel front-end (TEST INFRASTRUCTURE ONLY).

Both are third-party to the reference (vocos==0.1.0 and torchaudio==2.6.0 are pinned in
/root/reference/pyproject.toml:182,19 but absent from /root/reference and from this image), so these are
restatements of the published algorithms -- PARITY UNPINNED by the reference; tests/test_oracle_vocos.py pins
them with analytic known-answer tests (istft(stft(x)) == x, filterbank shape properties, lengths).
Call sites in the reference: F/infer/utils_infer.py:92-115,472 (Vocos), F/model/modules.py:75-101 (mel).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def vocos_decode(sd: dict, mel: torch.Tensor, num_layers: int = 8, n_fft: int = 1024, hop: int = 256) -> torch.Tensor:
    """Vocos.decode = ISTFTHead(VocosBackbone(mel)) for mel [b, 100, T] -> wave [b, hop (T-1)]."""
    dim = sd["backbone.embed.weight"].shape[0]
    x = F.conv1d(mel, sd["backbone.embed.weight"], sd["backbone.embed.bias"], padding=3)
    x = F.layer_norm(x.transpose(1, 2), (dim,), sd["backbone.norm.weight"], sd["backbone.norm.bias"], eps=1e-6).transpose(1, 2)
    for i in range(num_layers):
        p = f"backbone.convnext.{i}."
        r = x
        y = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=dim).transpose(1, 2)
        y = F.layer_norm(y, (dim,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
        y = F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"]))
        y = F.linear(y, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
        y = sd[p + "gamma"] * y
        x = r + y.transpose(1, 2)
    x = F.layer_norm(x.transpose(1, 2), (dim,), sd["backbone.final_layer_norm.weight"],
                     sd["backbone.final_layer_norm.bias"], eps=1e-6)
    y = F.linear(x, sd["head.out.weight"], sd["head.out.bias"]).transpose(1, 2)
    mag, phase = y.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    spec = mag * (torch.cos(phase) + 1j * torch.sin(phase))
    return torch.istft(spec, n_fft, hop, n_fft, torch.hann_window(n_fft), center=True)


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, f_max, n_freqs)   # torchaudio: linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def vocos_mel_spectrogram(wave: torch.Tensor, n_fft=1024, hop=256, n_mels=100, sr=24000) -> torch.Tensor:
    """get_vocos_mel_spectrogram (F/model/modules.py:75-101): MelSpectrogram(power=1, center=True, reflect,
    norm=None, htk) then clamp(1e-5).log().  wave [b, nw] -> [b, n_mels, 1 + nw // hop]."""
    spec = torch.stft(wave, n_fft, hop, n_fft, torch.hann_window(n_fft), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs()
    fb = melscale_fbanks_htk(n_fft // 2 + 1, 0.0, float(sr // 2), n_mels)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.clamp(min=1e-5).log()
