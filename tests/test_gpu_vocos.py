"""GPU parity (through the C ABI) of the Vocos decoder and the mel front-end vs the CPU oracle.
Tolerances from BASELINE.json north_star: 1e-4 on waveform samples, 1e-3 RMS on mel frames."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vocos_oracle as V  # noqa: E402
from tts_indic_server_f5_amd import synth  # noqa: E402


def _report(tag, got, ref):
    d = got.float().cpu() - ref.float().cpu()
    print(f"[parity] {tag}: rms_err {d.pow(2).mean().sqrt():.3e} max_err {d.abs().max():.3e} ref_rms {ref.float().pow(2).mean().sqrt():.3e}")
    return d.abs().max().item(), d.pow(2).mean().sqrt().item()


@pytest.fixture(scope="module")
def vocos():
    from tts_indic_server_f5_amd.vocoder import F5HipVocos
    return F5HipVocos(synth.vocos_state_dict())


@pytest.mark.parametrize("b,t", [(1, 936), (3, 77), (1, 2), (2, 129)])
def test_vocos_decode(vocos, b, t):
    g = torch.Generator().manual_seed(100 + t)
    mel = torch.randn(b, 100, t, generator=g) * 1.5 - 1.0
    ref = V.vocos_decode(synth.vocos_state_dict(), mel)
    got = vocos.decode(mel)
    assert got.shape == ref.shape == (b, 256 * (t - 1))
    mx, rms = _report(f"vocos b{b} t{t}", got, ref)
    assert mx < 1e-4


@pytest.mark.parametrize("b,nw", [(1, 120_000), (2, 24_000 + 77), (1, 1024)])
def test_mel_spectrogram(b, nw):
    from tts_indic_server_f5_amd.mel import mel_spectrogram
    wave = torch.cat([synth.ref_audio(nw, seed=1234 + i) for i in range(b)], dim=0)
    ref = V.vocos_mel_spectrogram(wave)
    got = mel_spectrogram(wave.cuda())
    assert got.shape == ref.shape == (b, 100, 1 + nw // 256)
    mx, rms = _report(f"mel b{b} nw{nw}", got, ref)
    assert rms < 1e-3 and mx < 5e-3


def test_mel_then_vocos_roundtrip_lengths(vocos):
    from tts_indic_server_f5_amd.mel import mel_spectrogram
    wave = synth.ref_audio(24_000).cuda()
    mel = mel_spectrogram(wave)
    out = vocos.decode(mel)
    assert out.shape[-1] == 256 * (mel.shape[-1] - 1) and torch.isfinite(out).all()
