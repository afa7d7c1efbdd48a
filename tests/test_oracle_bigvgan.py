"""CPU known-answer tests for the BigVGAN restatement (third-party, parity unpinned by the reference): SURVEY C.3."""
import torch

from oracle import bigvgan_oracle as B
from tts_indic_server_f5_amd import synth

SMALL = dict(upsample_initial_channel=64)


def test_aa_filter_dc_gain_and_symmetry():
    f = B.aa_filter()
    assert f.shape == (12,) and abs(float(f.sum()) - 1.0) < 1e-6
    assert torch.allclose(f, f.flip(0), atol=1e-7)


def test_up_down_length_and_dc():
    x = torch.full((1, 3, 40), 0.7)
    u = B.upsample2(x)
    assert u.shape == (1, 3, 80) and torch.allclose(u, torch.full_like(u, 0.7), atol=1e-5)   # DC gain 1 incl. the x2
    d = B.downsample2(u)
    assert d.shape == (1, 3, 40) and torch.allclose(d, x, atol=1e-5)


def test_snake_beta_identity_points():
    x = torch.zeros(1, 4, 8)
    assert torch.equal(B.snake_beta(x, torch.zeros(4), torch.zeros(4)), x)
    y = B.snake_beta(torch.full((1, 1, 1), math_pi_half()), torch.zeros(1), torch.zeros(1))
    assert abs(float(y) - (math_pi_half() + 1.0)) < 1e-5     # x + sin^2(x) / 1


def math_pi_half():
    import math
    return math.pi / 2


def test_forward_length_and_range():
    cfg = B.BigVGANConfig(upsample_initial_channel=64)
    sd = synth.bigvgan_state_dict(**SMALL)
    mel = torch.randn(2, 100, 9)
    w = B.bigvgan_forward(sd, cfg, mel)
    assert w.shape == (2, 1, 9 * 256) and torch.isfinite(w).all() and w.abs().max() <= 1.0
    assert w.std() > 1e-3


def test_bigvgan_mel_shape():
    wave = synth.ref_audio(24000)
    m = B.bigvgan_mel_spectrogram(wave)
    assert m.shape == (1, 100, 24000 // 256) and torch.isfinite(m).all()
    fb = B.librosa_slaney_mel(24000, 1024, 100)
    assert fb.shape == (100, 513) and (fb >= 0).all() and (fb.sum(1) > 0).all()
