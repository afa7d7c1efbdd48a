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
  * `TTSManager.load()` takes the model / vocoder objects (or a loader callable): checkpoints are not fetched from the hub;
  * `TTSManager(micro_batch=dict(max_requests=16, max_wait_ms=5))`: concurrent requests are collected for a few milliseconds and synthesized as ONE
    sampler batch (`infer.infer_requests`); with `ShardedSampler` as the model object that batch is dealt over the GPUs of the node
    (rank 0 serves HTTP and owns the queue, the other ranks sit in `rank_worker_loop`).  The reference serves one request at a
    time on one GPU (`S/routes/speech.py:19-41`).  A request's result does not depend on its batch: `load()` puts the model handle into
    the library's shape-invariant attention mode (`batch_invariant=True`); without it the same utterance alone and inside a batch can take
    different attention kernels, which agree to the last bits per launch but -- in the mixed GEMM mode -- drift apart to that mode's
    rounding-noise floor over a sample (measured 4.4e-4 rms after two Euler steps, `profiles/r03_attn_mode_tapdiff.txt`).
"""

import io
import queue
import threading
import time
import wave as _wave
from concurrent.futures import Future
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


class MicroBatcher:
    """Collects concurrent synthesis requests into one sampler batch.

    `submit((ref_audio, ref_text, gen_text))` returns a `concurrent.futures.Future`; a single worker thread takes the first waiting
    request, keeps collecting for at most `max_wait_ms` or until `max_requests` are waiting, runs `run_batch(list_of_requests)` (one
    `infer.infer_requests` call = one library call over all chunks of all requests) and resolves the futures in order.  One batch is in
    flight at a time -- the library allows one call per handle -- and the next one forms while it runs, so under load the batch size
    grows by itself.  A failing batch is retried request by request so one bad request cannot fail its neighbours -- unless the error
    says the backend itself is gone (`no_retry`, e.g. a rank of a sharded job failed): then the whole batch fails at once.
    `close()` resolves every request that is still queued with RuntimeError("MicroBatcher is closed"); nothing can be enqueued behind it."""

    def __init__(self, run_batch: Callable[[list], list], max_requests: int = 16, max_wait_ms: float = 5.0):
        self.run_batch, self.max_requests, self.max_wait = run_batch, int(max_requests), max_wait_ms / 1e3
        self._q: queue.Queue = queue.Queue()
        self._closed = False
        self._gate = threading.Lock()             # orders submit() against close(): no item can land behind the shutdown mark
        self.batch_sizes: list[int] = []          # observability: sizes of the batches run so far
        self._thread = threading.Thread(target=self._loop, name="f5hip-microbatcher", daemon=True)
        self._thread.start()

    def submit(self, request) -> Future:
        f: Future = Future()
        with self._gate:
            if self._closed:
                raise RuntimeError("MicroBatcher is closed")
            self._q.put((request, f))
        return f

    def close(self, timeout: float = 30.0):
        with self._gate:
            if self._closed:
                return
            self._closed = True
            self._q.put(None)                     # the shutdown mark: everything in front of it is still served
        self._thread.join(timeout=timeout)
        self._fail_pending()                      # (only non-empty if the worker thread did not get there: join timed out)

    def _fail_pending(self):
        while True:
            try:
                item = self._q.get_nowait()
            except queue.Empty:
                return
            if item is not None:
                item[1].set_exception(RuntimeError("MicroBatcher is closed"))

    def _collect(self):
        first = self._q.get()
        if first is None:
            return None
        batch, deadline = [first], time.monotonic() + self.max_wait
        while len(batch) < self.max_requests:
            left = deadline - time.monotonic()
            try:
                item = self._q.get(timeout=left) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                break
            if item is None:
                self._q.put(None)   # leave the shutdown mark for the loop (nothing can follow it: submit() is closed)
                break
            batch.append(item)
        return batch

    def _loop(self):
        while True:
            batch = self._collect()
            if batch is None:       # the shutdown mark: every request submitted before close() has been served
                break
            self.batch_sizes.append(len(batch))
            try:
                results = self.run_batch([r for r, _ in batch])
                if len(results) != len(batch):
                    raise RuntimeError(f"run_batch returned {len(results)} results for {len(batch)} requests")
                for (_, f), res in zip(batch, results):
                    f.set_result(res)
            except Exception as e:   # noqa: BLE001 -- isolate the failing request
                if len(batch) == 1 or getattr(e, "no_retry", False):
                    for _, f in batch:
                        f.set_exception(e)
                    continue
                for r, f in batch:
                    try:
                        f.set_result(self.run_batch([r])[0])
                    except Exception as e1:   # noqa: BLE001
                        f.set_exception(e1)
        self._fail_pending()


class TTSManager:
    """`S/core/managers.py:62-85`.  `model` is what `synthesize` calls: (text, ref_audio_path=..., ref_text=...) -> waveform."""

    def __init__(self, loader: Callable[[], tuple] | None = None, nfe_step: int = infer.nfe_step, cfg_strength: float = infer.cfg_strength,
                 sway_sampling_coef: float = infer.sway_sampling_coef, speed: float = infer.speed, mel_spec_type: str = "vocos",
                 micro_batch: dict | None = None, batch_invariant: bool = True):
        self.loader = loader
        self.batch_invariant = batch_invariant   # False: leave the model's attention mode alone (fastest kernel per launch shape)
        self.micro_batch = micro_batch            # e.g. dict(max_requests=16, max_wait_ms=5): batch concurrent requests
        self.batcher: MicroBatcher | None = None
        self.model = None
        self.model_obj = None
        self.vocoder = None
        self.opts = dict(nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed)
        self.mel_spec_type = mel_spec_type
        self._prep_cache: dict = {}   # prompt path -> (PreparedVoice, ref_text): clip / trim / resample / mel run once per voice
        self._prep_lock = threading.Lock()   # route handlers run in a thread pool: concurrent first requests of a voice prepare it once
        # The library allows ONE call in flight per handle (include/f5hip.h), the sampler keeps per-call state and noise comes from torch's
        # global generator: every entry into the device path takes this lock.  Without a batcher, concurrent HTTP requests therefore run one
        # after the other, like the reference's blocking `async def` handlers (S/routes/speech.py:19-41).
        self._device_lock = threading.Lock()
        self.request_timeout_s = 600.0       # a request never waits for its batch for ever

    def load(self, model_obj=None, vocoder=None):
        """Attach the sampler / vocoder objects (F5HipModel, F5HipVocos | F5HipBigVGAN), or build them with `loader`."""
        if not self.model:
            if model_obj is None:
                if self.loader is None:
                    raise ValueError("TTSManager.load needs a model object or a loader")
                model_obj, vocoder = self.loader()
            self.model_obj, self.vocoder = model_obj, vocoder
            # A served request must not depend on what it happened to be batched with: this handle runs the shape-invariant attention
            # arithmetic (per-handle setting of the library, ~3 % at batch 1; other handles of the process keep theirs).
            setter = getattr(getattr(model_obj, "local", model_obj), "set_attention_shape_invariant", None)
            if callable(setter) and self.batch_invariant:
                setter(True)
            self.model = self._call
            if self.micro_batch is not None:
                self.batcher = MicroBatcher(self._run_batch, **self.micro_batch)
        return self

    def _run_batch(self, requests):
        with self._device_lock:
            res = infer.infer_requests(requests, self.model_obj, self.vocoder, mel_spec_type=self.mel_spec_type, **self.opts)
        return [np.asarray(w, dtype=np.float32) for w, _, _ in res]

    def close(self):
        """Unload: stop the batcher (requests already queued are served, later ones refused) and drop the model objects."""
        if self.batcher is not None:
            self.batcher.close()
            self.batcher = None
        closer = getattr(self.model_obj, "close", None)
        if callable(closer):
            closer()                             # ShardedSampler: releases the worker ranks
        self.model = self.model_obj = self.vocoder = None

    def _voice(self, ref_audio_path, ref_text):
        """Once per voice: the reference's pre-step (clip to 15 s / trim silence, `preprocess_ref_audio_text`), then the prologue of
        every `infer_process` call for it (mono, rms gain, 24 kHz) and -- on first use -- its mel on the device."""
        key = (ref_audio_path, ref_text)
        with self._prep_lock:
            if key not in self._prep_cache:
                wav_path, ref_text_n = infer.preprocess_ref_audio_text(ref_audio_path, ref_text, show_info=lambda *_: None)
                self._prep_cache[key] = (infer.PreparedVoice(wav_path), ref_text_n)
            return self._prep_cache[key]

    def _call(self, text, ref_audio_path, ref_text):
        voice, ref_text_n = self._voice(ref_audio_path, ref_text)
        if self.batcher is not None:   # wait for the batch this request rides in (the route runs in a worker thread, see create_app)
            return self.batcher.submit((voice, ref_text_n, text)).result(timeout=self.request_timeout_s)
        return self._run_batch([(voice, ref_text_n, text)])[0]

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

    # The reference's handlers are `async def` around a blocking call, i.e. one request at a time.  Here the blocking part runs in
    # starlette's thread pool, so concurrent requests overlap and meet in the MicroBatcher queue (when the manager has one).
    from starlette.concurrency import run_in_threadpool

    @router.post("/audio/speech", response_class=StreamingResponse)
    async def synthesize_kannada(request: KannadaSynthesizeRequest):
        return await run_in_threadpool(_run, request.text, registry.default_voice, None, "synthesized_kannada_speech.wav")

    @router.post("/audio/speech/voice", response_class=StreamingResponse)
    async def synthesize_with_voice(request: SynthesizeRequest):     # the generic form the reference's helper already supports
        return await run_in_threadpool(_run, request.text, request.ref_audio_name, request.ref_text, "synthesized_speech.wav")

    app = FastAPI(title="F5-TTS on MI355X (HIP path)")
    app.include_router(router)
    return app


# ---------------------------------------------------------------------------------------------------------------- multi-GPU backend
class ShardedJobError(RuntimeError):
    """A rank of a sharded job failed.  Every rank still took part in the job's collectives (so nobody hangs), but the job has no
    result and the batch must not be retried request by request on a backend in an unknown state: `no_retry` tells MicroBatcher so, the
    sampler refuses further jobs, and the serving process should exit non-zero so that its supervisor starts fresh ranks."""
    no_retry = True


class ShardedSampler:
    """The model object of rank 0 in a one-process-per-GPU serving job: `sample_units` deals the units of a batch over the ranks
    (`sharding.shard_units`: longest-processing-time dealing with the SURVEY 8(d) cost model), every rank -- this one included --
    samples its share on its own GPU with its own replica of the weights, and the mels come back to rank 0 in unit order.
    There is no collective inside the ODE loop: one job broadcast (tokens, frame counts, the reference mels of the voices in the
    batch: 188 KB per voice) and one gather per batch, RCCL over xGMI when the process group's backend is "nccl".
    Ranks > 0 run `rank_worker_loop(local_model)`; `close()` on rank 0 releases them.
    Failure: a rank whose local `sample_units` raises still joins the gather with a failure header; rank 0 then raises
    `ShardedJobError` after the collective has completed on every rank, and refuses later jobs (`failed`).
    Noise: every rank draws the noise of ITS units from its own generator (like the reference's per-call `torch.randn`, unseeded in
    `infer_batch_process`), so an unseeded result is not reproducible across world sizes; pass `seed=` in the knobs for that."""

    def __init__(self, local_model, device=None):
        import torch
        import torch.distributed as dist
        self.local, self.dist, self.torch = local_model, dist, torch
        self.device = device if device is not None else getattr(local_model, "device", torch.device("cpu"))
        self.vocab_char_map = getattr(local_model, "vocab_char_map", None)
        self.failed: str | None = None

    # what infer.* needs from a model object
    def cond_mel(self, audio):
        return self.local.cond_mel(audio)

    def sample_units(self, audio, units, **knobs):
        if self.failed:
            raise ShardedJobError(f"sharded backend is down: {self.failed}")
        torch = self.torch
        b = len(units)
        audios = list(audio) if isinstance(audio, (list, tuple)) else [audio] * b
        voices, voice_of = [], []
        for a in audios:                                   # distinct voices of the batch, by object identity
            for k, v in enumerate(voices):
                if v is a:
                    voice_of.append(k)
                    break
            else:
                voices.append(a)
                voice_of.append(len(voices) - 1)
        mels = [(self.local.cond_mel(a) if a.ndim == 2 else a)[0].to(torch.float32) for a in voices]
        job = dict(units=[(list(t), int(f)) for t, f in units], voice_of=voice_of, mel_shapes=[tuple(m.shape) for m in mels], knobs=knobs)
        try:
            return _run_sharded_job(self.local, job, mels, self.device)
        except ShardedJobError as e:
            self.failed = str(e)
            raise

    def close(self):
        if self.dist.is_initialized() and self.dist.get_world_size() > 1:
            self.dist.broadcast_object_list([None], src=0)


def _run_sharded_job(local_model, job, mels, device):
    """Collective part shared by rank 0 (`job`, `mels` given) and the workers (both None): returns the mels of all units on rank 0.
    Every rank that entered the job's broadcast also enters its gather, whatever its local sampler did."""
    import sys
    import traceback

    import torch
    import torch.distributed as dist
    from .sharding import gather_waves, shard_units
    multi = dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    if multi:
        box = [job]
        dist.broadcast_object_list(box, src=0)
        job = box[0]
        if job is None:
            return None
        flat = torch.cat([m.reshape(-1) for m in mels]).to(device) if rank == 0 else torch.empty(sum(a * b for a, b in job["mel_shapes"]), device=device)
        dist.broadcast(flat, src=0)
        mels, k = [], 0
        for a, b in job["mel_shapes"]:
            mels.append(flat[k:k + a * b].view(a, b))
            k += a * b
    units = job["units"]
    shards = shard_units([f for _, f in units], world)
    mine = shards[rank]
    # payload of a rank: a status word (number of units, or -1: the local sampler failed), the row count of each unit, then their rows
    # (a unit's final duration can exceed the planned frames: sample() raises it to lens + 1 like the reference, cfm.py:136)
    local_error = None
    try:
        outs = local_model.sample_units([mels[job["voice_of"][i]][None] for i in mine], [units[i] for i in mine], **job["knobs"]) if mine else []
        if len(outs) != len(mine):
            raise RuntimeError(f"sample_units returned {len(outs)} mels for {len(mine)} units")
        head = torch.tensor([float(len(outs))] + [float(o.shape[0]) for o in outs], dtype=torch.float32, device=device)
        packed = torch.cat([head] + [o.reshape(-1).to(device, torch.float32) for o in outs])
    except Exception as e:   # noqa: BLE001 -- reported through the collective, never by leaving it
        local_error = e
        traceback.print_exc(file=sys.stderr)
        packed = torch.tensor([-1.0], dtype=torch.float32, device=device)
    got = gather_waves(packed, dst=0)
    if rank != 0:
        return []
    bad = [r for r, flat_r in enumerate(got) if float(flat_r[0]) < 0]
    if bad:
        err = ShardedJobError(f"sample_units failed on rank(s) {bad} of {world}" + (f": {local_error!r}" if local_error is not None else ""))
        raise err from local_error
    mel_dim = mels[0].shape[1]
    result = [None] * len(units)
    for r, flat_r in enumerate(got):
        k = 1 + len(shards[r])
        for j, i in enumerate(shards[r]):
            n = int(flat_r[1 + j])
            result[i] = flat_r[k:k + n * mel_dim].view(n, mel_dim)
            k += n * mel_dim
    return result


def rank_worker_loop(local_model, device=None):
    """What ranks > 0 of a serving job run: take part in every job rank 0's `ShardedSampler` broadcasts until it closes.  A job whose
    local sampler raised is reported to rank 0 inside the job's gather (`_run_sharded_job`) and the loop goes on."""
    import torch
    device = device if device is not None else getattr(local_model, "device", torch.device("cpu"))
    n = 0
    while _run_sharded_job(local_model, None, None, device) is not None:
        n += 1
    return n
