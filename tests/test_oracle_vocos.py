"""CPU known-answer tests for the third-party leaves the reference does not pin (vocos ISTFT head geometry,
torchaudio mel filterbank): SURVEY Appendix C.3."""
import torch

from oracle import vocos_oracle as V
from tts_indic_server_f5_amd import synth


def test_vocos_output_length_and_finite():
    sd = synth.vocos_state_dict()
    mel = torch.randn(2, 100, 40)
    w = V.vocos_decode(sd, mel)
    assert w.shape == (2, 256 * 39) and torch.isfinite(w).all()
    assert w.abs().max() < 50


def test_istft_of_stft_is_identity_on_head_geometry():
    x = torch.randn(1, 256 * 30)
    win = torch.hann_window(1024)
    s = torch.stft(x, 1024, 256, 1024, win, center=True, return_complex=True)
    y = torch.istft(s, 1024, 256, 1024, win, center=True)
    assert y.shape[-1] == 256 * (s.shape[-1] - 1)
    assert torch.allclose(y, x[:, : y.shape[-1]], atol=1e-5)


def test_mel_filterbank_properties():
    fb = V.melscale_fbanks_htk(513, 0.0, 12000.0, 100)
    assert fb.shape == (513, 100) and (fb >= 0).all() and fb.max() <= 1.0 + 1e-6
    assert (fb.sum(0) > 0).all()                       # every mel bin sees some frequency
    peak = fb.argmax(0)
    assert (peak[1:] >= peak[:-1]).all()               # centres increase monotonically
    for j in (3, 40, 99):                              # triangular: rises then falls
        col = fb[:, j]
        nz = col.nonzero().flatten()
        k = int(col.argmax())
        assert (col[nz[0]:k + 1].diff() >= -1e-7).all() and (col[k:nz[-1] + 1].diff() <= 1e-7).all()


def test_mel_spectrogram_shape_and_tone():
    sr, nw = 24000, 24000
    t = torch.arange(nw) / sr
    x = 0.5 * torch.sin(2 * torch.pi * 1000.0 * t)[None]
    mel = V.vocos_mel_spectrogram(x)
    assert mel.shape == (1, 100, 1 + nw // 256)
    fb = V.melscale_fbanks_htk(513, 0.0, 12000.0, 100)
    k = round(1000.0 / (12000.0 / 512))
    assert int(mel[0, :, 40].argmax()) == int(fb[k].argmax())   # energy lands in the bin covering 1 kHz
    assert mel.min() >= torch.log(torch.tensor(1e-5)) - 1e-6
