"""`/v1/audio/speech` over a stand-in sampler (CPU): route contract of S/routes/speech.py:19-41 and S/utils/tts_utils.py:38-65 --
status codes, messages, WAV response -- plus the manager's once-per-voice reference pre-step."""
import io
import wave

import numpy as np
import pytest
import torch

from tts_indic_server_f5_amd import serve


class FakeModel:
    """CFM.sample stand-in: returns a ramp mel so the glue (duration rule, ref strip, chunking) is exercised without a GPU."""
    def __init__(self):
        self.calls = []

    def sample(self, cond, text, duration, steps, cfg_strength, sway_sampling_coef):
        self.calls.append((tuple(cond.shape), text, duration, steps))
        return torch.zeros(1, duration, 100), None


class FakeVocoder:
    def decode(self, mel):
        n = mel.shape[-1] * 256
        return 0.25 * torch.sin(torch.arange(n) * 0.05)[None]


def _voice(tmp_path):
    x = (6000 * np.sin(2 * np.pi * 200 * np.arange(24000 * 3) / 24000)).astype(np.int16)
    p = tmp_path / "prompt.wav"
    with wave.open(str(p), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(24000)
        f.writeframes(x.tobytes())
    return str(p)


@pytest.fixture()
def client(tmp_path):
    from fastapi.testclient import TestClient
    reg = serve.VoiceRegistry()
    reg.add("KAN_F (Happy)", _voice(tmp_path), "reference words")
    mgr = serve.TTSManager(nfe_step=4)
    model = FakeModel()
    app = serve.create_app(mgr, reg)
    return TestClient(app), mgr, model


def test_speech_route_contract(client):
    c, mgr, model = client
    r = c.post("/v1/audio/speech", json={"text": "hello"})
    assert r.status_code == 503 and r.json()["detail"] == "TTS model not loaded"
    mgr.load(model, FakeVocoder())
    r = c.post("/v1/audio/speech", json={"text": "   "})
    assert r.status_code == 400 and r.json()["detail"] == "Text to synthesize cannot be empty."
    r = c.post("/v1/audio/speech/voice", json={"text": "hi", "ref_audio_name": "nobody"})
    assert r.status_code == 400 and r.json()["detail"] == "Invalid reference audio name."
    r = c.post("/v1/audio/speech", json={"text": "hello world, this is a test."})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    assert "synthesized_kannada_speech.wav" in r.headers["content-disposition"]
    with wave.open(io.BytesIO(r.content), "rb") as f:
        assert f.getframerate() == 24000 and f.getnchannels() == 1 and f.getsampwidth() == 2 and f.getnframes() > 0
    # the glue ran: reference text got its ". " (preprocess rule), cond is the trimmed clip + 50 ms, 4 NFE as configured
    shape, text, duration, steps = model.calls[-1]
    # ("reference words" -> ". " by the pre-step, then one more " " by infer_batch_process for a 1-byte last character: both are the reference's rules)
    assert steps == 4 and "".join(text[0]).startswith("reference words.  hello world")
    assert abs(shape[-1] - 24 * 3050) <= 48 and duration > shape[-1] // 256
    # second request with the same voice reuses the pre-processed clip
    n = len(mgr._prep_cache)
    assert c.post("/v1/audio/speech", json={"text": "again"}).status_code == 200 and len(mgr._prep_cache) == n


def test_wav_bytes_and_manager_errors():
    buf = serve.wav_bytes(np.array([0.0, 0.5, -1.0, 1.0], dtype=np.float32))
    with wave.open(buf, "rb") as f:
        pcm = np.frombuffer(f.readframes(4), dtype="<i2")
    assert pcm.tolist() == [0, 16384, -32768, 32767]
    with pytest.raises(ValueError):
        serve.TTSManager().synthesize("x", "p.wav", "t")
    with pytest.raises(serve.HTTPError) as e:
        serve.synthesize_speech(serve.TTSManager(), serve.VoiceRegistry(), "x", "none", None)
    assert e.value.status_code == 400
