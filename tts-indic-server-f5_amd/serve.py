"""`/v1/audio/speech` over the HIP path: the reference's route, request model, manager and helper with
the same names, status codes and response shape, and a local voice registry instead of a GitHub fetch per request.

Mirrors: `S/routes/speech.py:19-41` (route), `S/utils/tts_utils.py:22-65` (`KannadaSynthesizeRequest`,
`SynthesizeRequest`, `synthesize_speech`), `S/core/managers.py:62-85` (`TTSManager`: `.model`, `.load()`,
`.synthesize(text, ref_audio_path, ref_text)`).  Only this path of the server is built: no auth, rate limiting, logging
configuration, ASR / LLM / translation routes (out of scope).

Differences, explicit:
  * reference voices come from `VoiceRegistry` (name -> local 16-bit WAV + transcript); the reference downloads the prompt WAV
    from GitHub on every request (`tts_utils.py:40-46`), which this deployment target (no egress) cannot and should not do;
  * the response body is 16-bit PCM WAV at 24 kHz written with the stdlib (`soundfile`'s default WAV subtype for float input
    is PCM_16 as well);
  * `TTSManager.load()` takes the model / vocoder objects (or a loader callable): checkpoints are not fetched from the hub.
"""

import io
import wave as _wave
from dataclasses import dataclass, field
from typing import Callable

import numpy as np

from . import infer


@dataclass
class Voice:
    audio_path: str
    ref_text: str


@dataclass
class VoiceRegistry:
    """Local replacement of the reference's EXAMPLES table (`S/utils/tts_utils.py:12-19`): audio_name -> prompt clip + transcript."""
    voices: dict = field(default_factory=dict)
    default_voice: str = "KAN_F (Happy)"

    def add(self, name: str, audio_path: str, ref_text: str) -> None:
        self.voices[name] = Voice(audio_path, ref_text)

    def get(self, name: str):
        return self.voices.get(name)


class TTSManager:
    """`S/core/managers.py:62-85`.  `model` is what `synthesize` calls: (text, ref_audio_path=..., ref_text=...) -> waveform."""

    def __init__(self, loader: Callable[[], tuple] | None = None, nfe_step: int = infer.nfe_step, cfg_strength: float = infer.cfg_strength,
                 sway_sampling_coef: float = infer.sway_sampling_coef, speed: float = infer.speed, mel_spec_type: str = "vocos"):
        self.loader = loader
        self.model = None
        self.model_obj = None
        self.vocoder = None
        self.opts = dict(nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed)
        self.mel_spec_type = mel_spec_type
        self._prep_cache: dict = {}   # prompt path -> (processed wav path, ref_text): the pre-step runs once per voice

    def load(self, model_obj=None, vocoder=None):
        """Attach the sampler / vocoder objects (F5HipModel, F5HipVocos | F5HipBigVGAN), or build them with `loader`."""
        if not self.model:
            if model_obj is None:
                if self.loader is None:
                    raise ValueError("TTSManager.load needs a model object or a loader")
                model_obj, vocoder = self.loader()
            self.model_obj, self.vocoder = model_obj, vocoder
            self.model = self._call
        return self

    def _call(self, text, ref_audio_path, ref_text):
        key = (ref_audio_path, ref_text)
        if key not in self._prep_cache:
            self._prep_cache[key] = infer.preprocess_ref_audio_text(ref_audio_path, ref_text, show_info=lambda *_: None)
        wav_path, ref_text_n = self._prep_cache[key]
        wave, _, _ = infer.infer_process(wav_path, ref_text_n, text, self.model_obj, self.vocoder, mel_spec_type=self.mel_spec_type,
                                         show_info=lambda *_: None, **self.opts)
        return np.asarray(wave, dtype=np.float32)

    def synthesize(self, text, ref_audio_path, ref_text):
        if not self.model:
            raise ValueError("TTS model not loaded")
        return self.model(text, ref_audio_path=ref_audio_path, ref_text=ref_text)


def wav_bytes(audio: np.ndarray, sample_rate: int = infer.target_sample_rate) -> io.BytesIO:
    a = np.asarray(audio)
    if a.dtype != np.int16:
        a = np.clip(np.rint(a.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    buf = io.BytesIO()
    with _wave.open(buf, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sample_rate)
        f.writeframes(a.astype("<i2").tobytes())
    buf.seek(0)
    return buf


class HTTPError(Exception):
    """Carries (status_code, detail) out of `synthesize_speech`; the route turns it into fastapi.HTTPException."""

    def __init__(self, status_code: int, detail: str):
        super().__init__(detail)
        self.status_code, self.detail = status_code, detail


def synthesize_speech(tts_manager: TTSManager, registry: VoiceRegistry, text: str, ref_audio_name: str, ref_text: str | None):
    """`S/utils/tts_utils.py:38-65` with the same checks in the same order and the same messages."""
    voice = registry.get(ref_audio_name)
    if voice is not None and not ref_text:
        ref_text = voice.ref_text
    if voice is None:
        raise HTTPError(400, "Invalid reference audio name.")
    if not text.strip():
        raise HTTPError(400, "Text to synthesize cannot be empty.")
    if not ref_text or not ref_text.strip():
        raise HTTPError(400, "Reference text cannot be empty.")
    audio = tts_manager.synthesize(text, ref_audio_path=voice.audio_path, ref_text=ref_text)
    return wav_bytes(audio)


def create_app(tts_manager: TTSManager, registry: VoiceRegistry):
    """FastAPI app with the reference's `/v1/audio/speech` route (`S/routes/speech.py:19-41`)."""
    from fastapi import APIRouter, FastAPI, HTTPException
    from pydantic import BaseModel
    from starlette.responses import StreamingResponse

    class KannadaSynthesizeRequest(BaseModel):       # S/utils/tts_utils.py:27-28
        text: str

    class SynthesizeRequest(BaseModel):              # S/utils/tts_utils.py:22-25
        text: str
        ref_audio_name: str
        ref_text: str | None = None

    router = APIRouter(prefix="/v1", tags=["speech"])

    def _run(text, name, ref_text, filename):
        if not tts_manager.model:
            raise HTTPException(status_code=503, detail="TTS model not loaded")
        if not text.strip():
            raise HTTPException(status_code=400, detail="Text to synthesize cannot be empty.")
        try:
            buf = synthesize_speech(tts_manager, registry, text=text, ref_audio_name=name, ref_text=ref_text)
        except HTTPError as e:
            raise HTTPException(status_code=e.status_code, detail=e.detail)
        return StreamingResponse(buf, media_type="audio/wav", headers={"Content-Disposition": f"attachment; filename={filename}"})

    @router.post("/audio/speech", response_class=StreamingResponse)
    async def synthesize_kannada(request: KannadaSynthesizeRequest):
        return _run(request.text, registry.default_voice, None, "synthesized_kannada_speech.wav")

    @router.post("/audio/speech/voice", response_class=StreamingResponse)
    async def synthesize_with_voice(request: SynthesizeRequest):     # the generic form the reference's helper already supports
        return _run(request.text, request.ref_audio_name, request.ref_text, "synthesized_speech.wav")

    app = FastAPI(title="F5-TTS on MI355X (HIP path)")
    app.include_router(router)
    return app
