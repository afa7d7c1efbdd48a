"""GPU parity (through the C ABI) of the BigVGAN v2 generator and its mel front-end vs the CPU oracle
(third-party leaf, parity unpinned by the reference).  Tolerance: 1e-4 on waveform samples (BASELINE north_star)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bigvgan_oracle as B  # noqa: E402
from tts_indic_server_f5_amd import synth  # noqa: E402


def _report(tag, got, ref):
    d = got.float().cpu() - ref.float().cpu()
    print(f"[parity] {tag}: rms_err {d.pow(2).mean().sqrt():.3e} max_err {d.abs().max():.3e} ref_rms {ref.float().pow(2).mean().sqrt():.3e}")
    return d.abs().max().item(), d.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("b,t", [(1, 40), (2, 13), (1, 130)])
def test_bigvgan_small_config(b, t):
    """Reduced-width generator (initial channel 256 -> 128 ... 4): every stage incl. the padded-channel tail."""
    from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
    sd = synth.bigvgan_state_dict(upsample_initial_channel=256)
    voc = F5HipBigVGAN(sd, upsample_initial_channel=256)
    g = torch.Generator().manual_seed(200 + t)
    mel = torch.randn(b, 100, t, generator=g) * 1.5 - 1.0
    ref = B.bigvgan_forward(sd, B.BigVGANConfig(upsample_initial_channel=256), mel)
    got = voc(mel)
    assert got.shape == ref.shape == (b, 1, 256 * t)
    mx, rms = _report(f"bigvgan c0=256 b{b} t{t}", got, ref)
    assert mx < 1e-4


@pytest.mark.parametrize("b,t", [(1, 40), (2, 13)])
def test_bigvgan_small_config_fp16_fast_mode(b, t):
    """gemm_planes=3 (one fp16 operand plane: a third of the MFMA work).  NOT a parity mode: 11-bit operands through 36 residual
    convolutions per stage measure ~1e-3 max / 2e-4 rms on the waveform (tools/bigvgan_modes.py), ten times the 1e-4 bound that
    the default split-bf16 mode meets; the bound asserted here is that measured level with 2x head room, so a regression shows."""
    from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
    sd = synth.bigvgan_state_dict(upsample_initial_channel=256)
    voc = F5HipBigVGAN(sd, upsample_initial_channel=256, gemm_planes=3)
    g = torch.Generator().manual_seed(200 + t)
    mel = torch.randn(b, 100, t, generator=g) * 1.5 - 1.0
    ref = B.bigvgan_forward(sd, B.BigVGANConfig(upsample_initial_channel=256), mel)
    mx, rms = _report(f"bigvgan fp16 fast mode c0=256 b{b} t{t}", voc(mel), ref)
    assert mx < 2.5e-3 and rms < 5e-4


def test_bigvgan_full_config():
    """bigvgan_v2_24khz_100band_256x geometry (112 M parameters), 48 frames."""
    from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
    sd = synth.bigvgan_state_dict()
    voc = F5HipBigVGAN(sd)
    g = torch.Generator().manual_seed(77)
    mel = torch.randn(1, 100, 48, generator=g) * 1.5 - 1.0
    ref = B.bigvgan_forward(sd, B.BIGVGAN_V2_24K_100B_256X, mel)
    got = voc(mel)
    mx, rms = _report("bigvgan full t48", got, ref)
    clipped = (ref.abs() >= 1.0).float().mean().item()
    print(f"[parity] clipped fraction {clipped:.4f}")
    assert mx < 1e-4


@pytest.mark.parametrize("b,nw", [(1, 120_000), (2, 24_000 + 77)])
def test_mel_spectrogram_bigvgan(b, nw):
    from tts_indic_server_f5_amd.mel import mel_spectrogram_bigvgan
    wave = torch.cat([synth.ref_audio(nw, seed=1234 + i) for i in range(b)], dim=0)
    ref = B.bigvgan_mel_spectrogram(wave)
    got = mel_spectrogram_bigvgan(wave.cuda())
    assert got.shape == ref.shape
    mx, rms = _report(f"bigvgan-mel b{b} nw{nw}", got, ref)
    assert rms < 1e-3 and mx < 5e-3
