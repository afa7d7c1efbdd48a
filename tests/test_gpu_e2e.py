"""GPU end-to-end through the level-2 boundary: infer_process(ref_audio, ref_text, gen_text, model_obj, vocoder) on the HIP
objects vs the same host glue driving the CPU oracle (mel front-end -> CFM.sample -> Vocos), multi-chunk text incl. the
cross-fade, quiet reference (rms gain branch) and string tokenisation through a vocab."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit_oracle as O  # noqa: E402
from oracle import vocos_oracle as V  # noqa: E402
from tts_indic_server_f5_amd import infer, synth  # noqa: E402
from tts_indic_server_f5_amd.tokenizer import list_str_to_idx  # noqa: E402

ARCH = dict(dim=256, depth=4, heads=4, ff_mult=2, text_dim=64, conv_layers=2, text_num_embeds=96)
VOCAB = {chr(32 + i): i for i in range(96)}   # printable ASCII, " " -> 0


class OracleModel:
    """CFM.sample semantics on the CPU oracle (raw-wave cond -> oracle mel; list[str] text -> vocab lookup)."""

    def __init__(self, sd):
        self.sd, self.cfg = sd, O.DiTConfig(**ARCH)

    def sample(self, cond, text, duration, steps, cfg_strength, sway_sampling_coef):
        mel = V.vocos_mel_spectrogram(cond.cpu()).permute(0, 2, 1)
        ids = list_str_to_idx(text, VOCAB)
        out, _ = O.cfm_sample(self.sd, self.cfg, mel, ids, duration, steps=steps, cfg_strength=cfg_strength,
                              sway_sampling_coef=sway_sampling_coef, seed=None, keep_trajectory=False)
        return out, None


class OracleVocoder:
    def __init__(self, sd):
        self.sd = sd

    def decode(self, mel):
        return V.vocos_decode(self.sd, mel.cpu())


@pytest.mark.parametrize("amp", [0.15, 0.03])
def test_infer_process_matches_oracle_pipeline(amp):
    from tts_indic_server_f5_amd.model import DiTArch, F5HipModel
    from tts_indic_server_f5_amd.vocoder import F5HipVocos
    sd, vsd = synth.dit_state_dict(**ARCH), synth.vocos_state_dict()
    ref_audio = (synth.ref_audio(24000 * 2, amp=amp), 24000)
    ref_text = "Some call me nature."
    gen_text = "I do not care what you call me. I have been a silent spectator, watching species evolve. Always remember, I endure."
    kw = dict(nfe_step=8, cfg_strength=2.0, sway_sampling_coef=-1.0)
    hip_model = F5HipModel(DiTArch(**ARCH), sd, vocab_char_map=VOCAB)
    hip_voc = F5HipVocos(vsd)
    torch.manual_seed(123)   # the reference draws the noise from the global CPU generator (cfm.py:181-186)
    w_hip, sr, spec_hip = infer.infer_process(ref_audio, ref_text, gen_text, hip_model, hip_voc, device="cuda", **kw)
    torch.manual_seed(123)
    w_ref, _, spec_ref = infer.infer_process(ref_audio, ref_text, gen_text, OracleModel(sd), OracleVocoder(vsd), **kw)
    assert sr == 24000 and w_hip.shape == w_ref.shape and spec_hip.shape == spec_ref.shape
    assert spec_hip.shape[0] == 100 and len(infer.chunk_text(gen_text, max_chars=int(len(ref_text.encode()) / 2 * 23))) >= 1
    mel_rms = float(np.sqrt(np.mean((spec_hip - spec_ref) ** 2)))
    wav_max = float(np.max(np.abs(w_hip - w_ref)))
    print(f"[parity] e2e amp={amp}: mel rms err {mel_rms:.3e}  wave max err {wav_max:.3e}  wave rms {np.sqrt(np.mean(w_ref ** 2)):.3e}  n={len(w_ref)}")
    assert mel_rms < 1e-3
    assert wav_max < 1e-4   # north_star waveform bound, end to end (vocoder-only parity is 1e-6)


def test_speech_endpoint_on_hip_path(tmp_path):
    """POST /v1/audio/speech (serve.py, S/routes/speech.py:19-41) through TTSManager -> preprocess_ref_audio_text -> infer_process on
    the HIP objects: the WAV in the response is the 16-bit quantisation of the direct infer_process output, and that output matches
    the CPU oracle driven by the same glue."""
    import io
    import wave
    from fastapi.testclient import TestClient
    from tts_indic_server_f5_amd import serve
    from tts_indic_server_f5_amd.model import DiTArch, F5HipModel
    from tts_indic_server_f5_amd.vocoder import F5HipVocos
    sd, vsd = synth.dit_state_dict(**ARCH), synth.vocos_state_dict()
    x = np.concatenate([np.zeros(4800), synth.ref_audio(24000 * 2, amp=0.12)[0].numpy(), np.zeros(2400)])
    prompt = tmp_path / "voice.wav"
    with wave.open(str(prompt), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(24000)
        f.writeframes(np.clip(np.rint(x * 32768), -32768, 32767).astype("<i2").tobytes())
    reg = serve.VoiceRegistry()
    reg.add("KAN_F (Happy)", str(prompt), "Some call me nature")
    mgr = serve.TTSManager(nfe_step=8).load(F5HipModel(DiTArch(**ARCH), sd, vocab_char_map=VOCAB), F5HipVocos(vsd))
    client = TestClient(serve.create_app(mgr, reg))
    text = "I have been a silent spectator. Always remember, I endure."
    torch.manual_seed(7)
    r = client.post("/v1/audio/speech", json={"text": text})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    with wave.open(io.BytesIO(r.content), "rb") as f:
        assert f.getframerate() == 24000
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
    wav_path, ref_text = infer.preprocess_ref_audio_text(str(prompt), "Some call me nature", show_info=lambda *_: None)
    assert ref_text == "Some call me nature. "
    kw = dict(nfe_step=8, cfg_strength=2.0, sway_sampling_coef=-1.0, show_info=lambda *_: None)
    torch.manual_seed(7)
    w_ref, _, _ = infer.infer_process(wav_path, ref_text, text, OracleModel(sd), OracleVocoder(vsd), **kw)
    assert len(pcm) == len(w_ref)
    err = float(np.max(np.abs(pcm.astype(np.float64) / 32768.0 - w_ref)))
    print(f"[parity] /v1/audio/speech vs oracle pipeline: max err {err:.3e} (16-bit step 3.1e-5), n={len(pcm)}")
    assert err < 1e-4 + 1.0 / 32768


def test_infer_requests_two_voices_one_batch():
    """`infer.infer_requests` on the HIP objects: requests with DIFFERENT reference voices (different prompt lengths, one below the
    rms floor) are sampled as one library call, every unit with its own prompt length (`lens`) and batch-1 semantics; each result equals
    the request run alone on the same objects (2e-5: batch composition does not leak) and the CPU oracle pipeline (1e-3 / 1e-4).
    Also the multi-voice `[tag]` front-end (F/infer/infer_cli.py:181-208) over the same batch path."""
    from tts_indic_server_f5_amd.model import DiTArch, F5HipModel
    from tts_indic_server_f5_amd.vocoder import F5HipVocos
    sd, vsd = synth.dit_state_dict(**ARCH), synth.vocos_state_dict()
    kw = dict(nfe_step=8, cfg_strength=2.0, sway_sampling_coef=-1.0)
    va = (synth.ref_audio(24000 * 2, amp=0.15), 24000)
    vb = (synth.ref_audio(int(24000 * 1.4), seed=9, amp=0.03), 24000)
    reqs = [(va, "Some call me nature.", "I do not care what you call me. I have been a silent spectator, watching species evolve."),
            (vb, "Others say mother.", "Always remember, I endure."),
            (va, "Some call me nature.", "Short.")]
    model, voc = F5HipModel(DiTArch(**ARCH), sd, vocab_char_map=VOCAB), F5HipVocos(vsd)
    torch.manual_seed(77)
    batch = infer.infer_requests(reqs, model, voc, device="cuda", **kw)
    torch.manual_seed(77)
    alone = [infer.infer_process(a, rt, gt, model, voc, device="cuda", **kw) for a, rt, gt in reqs]
    torch.manual_seed(77)
    oracle = [infer.infer_process(a, rt, gt, OracleModel(sd), OracleVocoder(vsd), **kw) for a, rt, gt in reqs]
    for i, ((w, sr, s), (w1, _, s1), (w0, _, s0)) in enumerate(zip(batch, alone, oracle)):
        assert sr == 24000 and w.shape == w1.shape == w0.shape and s.shape == s1.shape == s0.shape
        d_alone, d_mel, d_wav = float(np.abs(w - w1).max()), float(np.sqrt(np.mean((s - s0) ** 2))), float(np.abs(w - w0).max())
        print(f"[parity] infer_requests item {i}: vs alone wave {d_alone:.3e}; vs oracle mel rms {d_mel:.3e} wave max {d_wav:.3e}")
        assert d_alone < 2e-5 and float(np.sqrt(np.mean((s - s1) ** 2))) < 2e-5
        assert d_mel < 1e-3 and d_wav < 1e-4
    voices = {"main": dict(ref_audio=va, ref_text="Some call me nature."), "b": dict(ref_audio=vb, ref_text="Others say mother.")}
    script = "I do not care. [b] Always remember, I endure. [main] Short."
    torch.manual_seed(78)
    w, sr, specs = infer.infer_multi_voice(script, voices, model, voc, device="cuda", **kw)
    torch.manual_seed(78)
    parts = [infer.infer_process(voices[v]["ref_audio"], voices[v]["ref_text"], t, model, voc, device="cuda", **kw)[0]
             for v, t in infer.split_voice_tags(script, voices)]
    assert len(specs) == 3 and float(np.abs(w - np.concatenate(parts)).max()) < 2e-5
