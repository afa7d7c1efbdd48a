"""CPU: the host-side mirror of the reference's inference driver (chunk_text, infer_batch_process glue: rms gain,
duration rule, ref-frame strip, cross-fade) against fixtures produced by running the reference's own
F/infer/utils_infer.py (tests/golden/gen_golden.py, stub sampler/vocoder with closed-form outputs)."""
import json
import os

import numpy as np
import pytest
import torch

from tts_indic_server_f5_amd import infer


def test_chunk_text_matches_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "chunk_text.json"), encoding="utf-8"))
    assert len(cases) >= 30
    for c in cases:
        assert infer.chunk_text(c["text"], max_chars=c["max_chars"]) == c["chunks"], (c["text"][:40], c["max_chars"])


def test_defaults_match_reference_constants():
    assert (infer.target_sample_rate, infer.n_mel_channels, infer.hop_length, infer.win_length, infer.n_fft) == (24000, 100, 256, 1024, 1024)
    assert (infer.target_rms, infer.cross_fade_duration, infer.nfe_step, infer.cfg_strength, infer.sway_sampling_coef, infer.speed) == (0.1, 0.15, 32, 2.0, -1.0, 1.0)


class _StubModel:   # same closed-form stand-ins as tests/golden/gen_golden.py::gen_glue_fixtures
    def __init__(self):
        self.calls = []

    def sample(self, cond, text, duration, steps, cfg_strength, sway_sampling_coef):
        self.calls.append(dict(nw=int(cond.shape[-1]), text="".join(text[0]), duration=int(duration), steps=steps))
        n = duration
        base = torch.arange(n * 100, dtype=torch.float32).reshape(1, n, 100) / (n * 100)
        return base * (1 + len(self.calls)), None


class _StubVocoder:
    def decode(self, mel):
        t = mel.shape[-1]
        k = torch.arange(256 * (t - 1), dtype=torch.float32)
        return (torch.sin(k * 0.01) * mel.mean())[None]


@pytest.mark.parametrize("idx,amp", [(0, 0.3), (1, 0.02)])
def test_infer_batch_process_glue(golden_dir, idx, amp):
    meta = json.load(open(os.path.join(golden_dir, "glue_calls.json")))[idx]
    z = np.load(os.path.join(golden_dir, f"glue_case_amp{amp}.npz"))
    audio = torch.from_numpy(z["audio"])
    ref_text = "Some call me nature, others call me mother nature. "
    gens = ["I do not care.", "I have been a silent spectator, watching.", "Short."]
    m = _StubModel()
    wave, sr, spec = infer.infer_batch_process((audio, meta["sr"]), ref_text, gens, m, _StubVocoder(), nfe_step=4)
    assert sr == 24000
    assert m.calls == meta["calls"]                      # duration rule, token text, nfe plumbed through
    assert str(wave.dtype) == meta["wave_dtype"] and len(wave) == meta["n"]   # float64 after cross-fade
    assert list(spec.shape) == meta["spec_shape"]
    np.testing.assert_allclose(wave, z["wave"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(spec, z["spec"], rtol=0, atol=1e-6)


def test_text_to_tokens_rules():
    assert infer.text_to_tokens(["a;b“c”"]) == [list('a,b"c"')]
    with pytest.raises(NotImplementedError):
        infer.text_to_tokens(["中文"])
    assert infer.text_to_tokens(["ಕನ್ನಡ."]) == [list("ಕನ್ನಡ.")]


def test_load_wav_roundtrip(tmp_path):
    import wave
    x = (np.sin(np.arange(2400) * 0.05) * 12000).astype("<i2")
    p = str(tmp_path / "r.wav")
    with wave.open(p, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(24000); f.writeframes(x.tobytes())
    a, sr = infer.load_wav(p)
    assert sr == 24000 and a.shape == (1, 2400) and abs(float(a[0, 10]) - x[10] / 32768.0) < 1e-7


def test_resample_known_answers():
    """torchaudio-style sinc resampler (third-party leaf, parity unpinned): length rule, tone preservation, identity."""
    import math
    sr = 16000
    n = 16000
    t = torch.arange(n) / sr
    x = 0.5 * torch.sin(2 * math.pi * 440.0 * t)[None]
    y = infer.resample_sinc_hann(x, sr, 24000)
    assert y.shape == (1, math.ceil(24000 * n / sr))
    ref = 0.5 * torch.sin(2 * math.pi * 440.0 * torch.arange(y.shape[1]) / 24000.0)
    assert (y[0, 200:-200] - ref[200:-200]).abs().max() < 2e-3        # band-limited tone survives, away from the edges
    assert infer.resample_sinc_hann(x, 24000, 24000) is x
    z = infer.resample_sinc_hann(torch.randn(2, 44100), 44100, 24000)
    assert z.shape == (2, 24000) and torch.isfinite(z).all()
