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
    # the reference's space-before-a-multi-character-ASCII-segment rule (F/model/utils.py:153-156) under jieba's segmentation
    # (known answers derived from jieba 0.42.1's published algorithm; the package is absent here: parity unpinned)
    T = lambda s: "".join(infer.text_to_tokens([s])[0])
    assert T("hello world") == "hello world"                 # a segment behind a space gets no second one
    assert T("well-known") == "well- known"
    assert T("(hello") == "( hello"
    assert T("a,b2") == "a, b2"
    assert T("x: yes 'no' \"ok\"") == "x: yes 'no' \"ok\""   # colon / quotes / space in front: no insertion
    assert T("pi is 3.14 or 50%!") == "pi is 3.14 or 50%!"   # digits with a decimal part and a percent sign stay one segment
    assert T("ಕ123 ಕabc") == "ಕ 123 ಕ abc"                  # ASCII run directly behind an Indic character
    assert T("a...b") == "a ...b"                             # a run of dots is one (multi-character, ASCII) segment; the single "b" gets none
    assert T("ಕನ್ನಡ ಪಠ್ಯ.") == "ಕನ್ನಡ ಪಠ್ಯ."
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


# ---------------------------------------------------------------- reference-audio pre-step (audio_prep.py, F/infer/utils_infer.py:263-350)
def _tone(ms, rate=24000, amp=8000, f=220.0):
    import numpy as np
    n = int(rate * ms / 1000)
    return (amp * np.sin(2 * np.pi * f * np.arange(n) / rate)).astype(np.int16)


def _zeros(ms, rate=24000):
    import numpy as np
    return np.zeros(int(rate * ms / 1000), dtype=np.int16)


def _write_wav(path, x, rate=24000):
    import wave
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(rate)
        f.writeframes(x.astype("<i2").tobytes())


def test_pcm_segment_rms_matches_audioop():
    """`rms` restates audioop.rms (floor of the root mean square over all interleaved samples): checked against the stdlib."""
    import warnings
    import numpy as np
    from tts_indic_server_f5_amd.audio_prep import PcmSegment
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import audioop
    rng = np.random.default_rng(5)
    x = rng.integers(-20000, 20000, size=(24000, 2)).astype(np.int16)
    seg = PcmSegment(x, 24000)
    for a, b in [(0, 1000), (10, 20), (333, 777), (999, 1000)]:
        fa, fb = int(a * 24.0), int(b * 24.0)
        assert seg.rms_ms(a, b) == audioop.rms(x[fa:fb].tobytes(), 2)
    assert len(seg) == 1000 and abs(seg.dbfs_ms(0, 1000) - 20 * np.log10(seg.rms / 32768.0)) < 1e-12


def test_detect_and_split_on_silence_known_answers():
    import numpy as np
    from tts_indic_server_f5_amd import audio_prep as AP
    seg = AP.PcmSegment(np.concatenate([_tone(2000), _zeros(1500), _tone(2000)]), 24000)
    assert AP.detect_silence(seg, min_silence_len=1000, silence_thresh=-50, seek_step=10) == [[2000, 3500]]
    assert AP.detect_nonsilent(seg, min_silence_len=1000, silence_thresh=-50, seek_step=10) == [[0, 2000], [3500, 5500]]
    chunks = AP.split_on_silence(seg, min_silence_len=1000, silence_thresh=-50, keep_silence=1000, seek_step=10)
    assert [len(c) for c in chunks] == [2750, 2750]          # padded ranges overlap -> split at the midpoint (3000 + 2500) // 2
    assert AP.detect_silence(seg, min_silence_len=2000, silence_thresh=-50, seek_step=10) == []   # pause shorter than the window
    lead = AP.PcmSegment(np.concatenate([_zeros(300), _tone(1000), _zeros(200)]), 24000)
    assert AP.detect_leading_silence(lead, silence_threshold=-42) == 300
    trimmed = AP.remove_silence_edges(lead)
    assert len(trimmed) in (999, 1000) and abs(int(trimmed.frames[1, 0]) - int(_tone(10)[1])) <= 1


def test_preprocess_ref_audio_text(tmp_path):
    import numpy as np
    from tts_indic_server_f5_amd import audio_prep as AP
    from tts_indic_server_f5_amd.infer import load_wav, preprocess_ref_audio_text
    msgs = []
    # (1) 20 s with a 1.2 s pause at 8 s: clipped at the pause, edges trimmed, 50 ms of silence appended
    p1 = tmp_path / "pause.wav"
    _write_wav(p1, np.concatenate([_tone(8000), _zeros(1200), _tone(10800)]))
    out, text = preprocess_ref_audio_text(str(p1), "hello there", show_info=msgs.append)
    w, sr = load_wav(out)
    assert sr == 24000 and abs(w.shape[-1] - 24 * 8050) <= 48 and text == "hello there. "
    assert msgs and "(1)" in msgs[0]
    assert float(w[0, -1200:].abs().max()) == 0.0 and float(w[0, :240].abs().max()) > 0.1
    # (3) 20 s without any pause: hard cut at 15 s (+ 50 ms)
    msgs.clear()
    p2 = tmp_path / "long.wav"
    _write_wav(p2, _tone(20000))
    out2, text2 = preprocess_ref_audio_text(str(p2), "Already ends.", show_info=msgs.append)
    w2, _ = load_wav(out2)
    assert abs(w2.shape[-1] - 24 * 15050) <= 48 and text2 == "Already ends. " and any("(3)" in m for m in msgs)
    # short clip with silent edges: only trimmed; CJK full stop and ". " endings are left alone; empty text needs the ASR leaf
    p3 = tmp_path / "short.wav"
    _write_wav(p3, np.concatenate([_zeros(400), _tone(3000), _zeros(300)]))
    out3, text3 = preprocess_ref_audio_text(str(p3), "你好。", show_info=msgs.append)
    w3, _ = load_wav(out3)
    assert abs(w3.shape[-1] - 24 * 3050) <= 48 and text3 == "你好。"
    assert preprocess_ref_audio_text(str(p3), "x. ", show_info=msgs.append)[1] == "x. "
    import pytest
    with pytest.raises(NotImplementedError):
        preprocess_ref_audio_text(str(p3), "  ", show_info=msgs.append)
    with pytest.raises(ValueError):
        _write_wav(tmp_path / "lo.wav", _tone(1000, rate=8000), rate=8000)
        preprocess_ref_audio_text(str(tmp_path / "lo.wav"), "a", show_info=msgs.append)
