"""Drop-in mirror of the reference's inference driver (F/infer/utils_infer.py): module constants (:40-53),
`chunk_text` (:61-88), `infer_process` (:357-400) and `infer_batch_process` (:406-524), with the same
signatures, defaults, return triple and quirks (UTF-8 byte budgets, `ref_audio_len = nw // 256`, float64
cross-fade ramps), running the sampler and vocoder on the HIP objects (`F5HipModel`, `F5HipVocos`).

Host-side differences, all explicit:
  * reference audio is read with the stdlib `wave` module (16-bit PCM WAV) or passed as a `(tensor, sr)` pair:
    torchaudio is not part of this image;
  * resampling to 24 kHz restates torchaudio.transforms.Resample (sinc interpolation, Hann window, width 6, rolloff
    0.99; third-party leaf, parity unpinned) on the host, like the reference does before `.to(device)`;
  * `preprocess_ref_audio_text` (silence clipping of the reference clip, ". " rule) is restated without pydub in `audio_prep.py`;
  * `convert_char_to_pinyin` (jieba + pypinyin) is replaced by `text_to_tokens`, which reproduces the reference's
    behaviour for text without CJK characters (per-character tokens, the same punctuation translation table)
    and rejects CJK input instead of silently mis-tokenising it.
"""
from __future__ import annotations

import re
import wave as _wave

import numpy as np
import torch

from .audio_prep import preprocess_ref_audio_text, remove_silence_edges  # noqa: F401  (F/infer/utils_infer.py:263-350)
from .loaders import DiT, MMDiT, UNetT, load_checkpoint, load_model, load_vocoder  # noqa: F401  (F/infer/utils_infer.py:92-130,175-260)

# ----------------------------------------- F/infer/utils_infer.py:40-53
target_sample_rate = 24000
n_mel_channels = 100
hop_length = 256
win_length = 1024
n_fft = 1024
mel_spec_type = "vocos"
target_rms = 0.1
cross_fade_duration = 0.15
ode_method = "euler"
nfe_step = 32
cfg_strength = 2.0
sway_sampling_coef = -1.0
speed = 1.0
fix_duration = None


def chunk_text(text, max_chars=135):
    """F/infer/utils_infer.py:61-88: split at punctuation, greedily pack sentences by UTF-8 byte budget."""
    chunks = []
    current = ""
    for sentence in re.split(r"(?<=[;:,.!?])\s+|(?<=[；：，。！？])", text):
        piece = sentence + " " if sentence and len(sentence[-1].encode("utf-8")) == 1 else sentence
        if len(current.encode("utf-8")) + len(sentence.encode("utf-8")) <= max_chars:
            current += piece
        else:
            if current:
                chunks.append(current.strip())
            current = piece
    if current:
        chunks.append(current.strip())
    return chunks


_CUSTOM_TRANS = str.maketrans({";": ",", "“": '"', "”": '"', "‘": "'", "’": "'"})   # F/model/utils.py:142-144

# jieba 0.42.1 `cut(text)` (default mode, HMM on) restated for text WITHOUT CJK characters (third-party leaf, absent here: parity
# unpinned, known-answer tests in tests/test_host_glue.py).  Its published algorithm: blocks matching re_han_default go to the
# dictionary cutter, everything else is split at whitespace and yielded character by character; inside a block every character that
# starts no dictionary word is buffered and the buffer goes through finalseg.cut, whose non-Han path splits at re_skip -- runs of
# [a-zA-Z0-9]+(.digits)?%? stay whole and so do the runs of "+#&._%-" between them.  (jieba's dictionary holds a handful of entries
# with Latin letters, e.g. "AT&T", "C++": those would come out as one segment there and as several here.)
_RE_HAN_DEFAULT = re.compile(r"([\u4E00-\u9FD5a-zA-Z0-9+#&\._%\-]+)")
_RE_SKIP_DEFAULT = re.compile(r"(\r\n|\s)")
_RE_SKIP_FINAL = re.compile(r"([a-zA-Z0-9]+(?:\.\d+)?%?)")


def _segments_non_cjk(text):
    for blk in _RE_HAN_DEFAULT.split(text):
        if not blk:
            continue
        if _RE_HAN_DEFAULT.match(blk):
            if len(blk) == 1:
                yield blk
            else:
                yield from (x for x in _RE_SKIP_FINAL.split(blk) if x)
        else:
            for x in _RE_SKIP_DEFAULT.split(blk):
                if _RE_SKIP_DEFAULT.match(x):
                    yield x
                else:
                    yield from x


def text_to_tokens(text_list):
    """convert_char_to_pinyin (F/model/utils.py:140-177) for text without CJK characters: translation table, jieba-style
    segmentation, then the reference's rule per segment -- a pure-ASCII segment longer than one character gets a space in front
    unless the previous token is one of space, colon, quote (:153-156), and is spelled out character by character; every other
    segment (Indic scripts arrive one character per segment, and pypinyin returns non-Han characters unchanged) passes through
    character by character.  CJK input is rejected instead of silently mis-tokenised (the pinyin front-end needs jieba's dictionary
    and pypinyin's tables, which are not available offline)."""
    out = []
    for text in text_list:
        text = text.translate(_CUSTOM_TRANS)
        if any("\u3100" <= c <= "\u9fff" for c in text):
            raise NotImplementedError("CJK text needs the pinyin front-end (jieba/pypinyin), which is not on this path yet")
        chars = []
        for seg in _segments_non_cjk(text):
            if len(seg.encode("utf-8")) == len(seg) and chars and len(seg) > 1 and chars[-1] not in " :'\"":
                chars.append(" ")
            chars.extend(seg)
        out.append(chars)
    return out


def resample_sinc_hann(wave: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                       rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.transforms.Resample(orig_freq, new_freq) (call site F/infer/utils_infer.py:430-432), default
    "sinc_interp_hann" method of torchaudio 2.6: polyphase windowed-sinc kernel applied as a strided conv1d.
    wave [channels, n] -> [channels, ceil(n * new / orig)]."""
    import math
    if orig_freq == new_freq:
        return wave
    g = math.gcd(int(orig_freq), int(new_freq))
    of, nf = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(of, nf) * rolloff
    width = math.ceil(lowpass_filter_width * of / base_freq)
    idx = torch.arange(-width, width + of, dtype=torch.float64)[None, None] / of
    t = torch.arange(0, -nf, -1, dtype=torch.float64)[:, None, None] / nf + idx
    t = (t * base_freq).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base_freq / of)
    kernels = kernels.to(torch.float32)
    shape = wave.shape
    w = wave.reshape(-1, shape[-1]).to(torch.float32)
    length = w.shape[-1]
    w = torch.nn.functional.pad(w, (width, width + of))
    out = torch.nn.functional.conv1d(w[:, None], kernels, stride=of)
    out = out.transpose(1, 2).reshape(w.shape[0], -1)
    target = math.ceil(nf * length / of)
    return out[..., :target].reshape(*shape[:-1], target)


def load_wav(path):
    """16-bit PCM WAV -> (float32 tensor [channels, samples] in [-1, 1), sample_rate) like torchaudio.load."""
    with _wave.open(path, "rb") as f:
        sr, ch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if sw != 2:
        raise ValueError("only 16-bit PCM WAV reference audio is supported")
    a = np.frombuffer(raw, dtype="<i2").reshape(-1, ch).T.astype(np.float32) / 32768.0
    return torch.from_numpy(np.ascontiguousarray(a)), sr


def infer_process(ref_audio, ref_text, gen_text, model_obj, vocoder, mel_spec_type=mel_spec_type, show_info=print,
                  progress=None, target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step,
                  cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed,
                  fix_duration=fix_duration, device=None):
    """F/infer/utils_infer.py:357-400."""
    audio, sr = ref_audio if isinstance(ref_audio, tuple) else load_wav(ref_audio)
    max_chars = int(len(ref_text.encode("utf-8")) / (audio.shape[-1] / sr) * (25 - audio.shape[-1] / sr))
    gen_text_batches = chunk_text(gen_text, max_chars=max_chars)
    return infer_batch_process((audio, sr), ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type=mel_spec_type,
                               progress=progress, target_rms=target_rms, cross_fade_duration=cross_fade_duration,
                               nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef,
                               speed=speed, fix_duration=fix_duration, device=device)


def _prepare_reference(audio, sr, rms_floor, device):
    """Prologue of infer_batch_process (F/infer/utils_infer.py:423-433): mono mix, gain up to `rms_floor`, resample to 24 kHz.
    Returns (audio [1, nw] fp32, measured rms)."""
    if audio.shape[0] > 1:
        audio = torch.mean(audio, dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < rms_floor:
        audio = audio * rms_floor / rms
    if sr != target_sample_rate:
        audio = resample_sinc_hann(audio, sr, target_sample_rate)
    if device is not None:
        audio = audio.to(device)
    return audio, rms


def plan_units(ref_text, gen_text_batches, ref_frames, speed=1.0, fix_duration=None, tokenizer=None):
    """One sampling unit per text chunk: (tokens of ref_text + chunk, total frames).  Duration rule of the reference
    (F/infer/utils_infer.py:446-454): UTF-8 byte lengths, `ref_frames = n_samples // hop` (one less than the mel has: SURVEY B2)."""
    tokenizer = tokenizer or text_to_tokens
    ref_bytes = len(ref_text.encode("utf-8"))
    units = []
    for chunk in gen_text_batches:
        if fix_duration is not None:
            frames = int(fix_duration * target_sample_rate / hop_length)
        else:
            frames = ref_frames + int(ref_frames / ref_bytes * len(chunk.encode("utf-8")) / speed)
        units.append((tokenizer([ref_text + chunk])[0], frames))
    return units


def cross_fade_concat(waves, fade_seconds, sample_rate=target_sample_rate):
    """Joins the chunk waveforms (F/infer/utils_infer.py:485-519): plain concatenation for a non-positive fade, else a linear
    cross-fade over min(fade, len(prev), len(next)) samples.  The ramps are float64 (np.linspace), so the result is float64 from the
    second chunk on, exactly like the reference's."""
    if fade_seconds <= 0:
        return np.concatenate(waves)
    out = waves[0]
    for nxt in waves[1:]:
        n = min(int(fade_seconds * sample_rate), len(out), len(nxt))
        if n <= 0:
            out = np.concatenate([out, nxt])
            continue
        ramp = np.linspace(0, 1, n)
        out = np.concatenate([out[:-n], out[-n:] * ramp[::-1] + nxt[:n] * ramp, nxt[n:]])
    return out


class PreparedVoice:
    """The per-voice part of `infer_process` done once: the reference wave after mono mix / rms gain / resampling
    (F/infer/utils_infer.py:423-433), its measured rms, its duration in seconds before resampling (the `max_chars` rule, :379) and --
    filled in by the first request that uses it -- the reference mel on the device (the "reference latents": 188 KB for a 5 s
    prompt).  `serve.TTSManager` keeps one per voice; `infer_requests` accepts it wherever a `ref_audio` is expected."""

    def __init__(self, ref_audio, target_rms=0.1, device=None):
        wav, sr = ref_audio if isinstance(ref_audio, tuple) else load_wav(ref_audio)
        self.seconds = wav.shape[-1] / sr
        self.audio, self.rms = _prepare_reference(wav, sr, target_rms, device)
        self.ref_frames = self.audio.shape[-1] // hop_length
        self.mel = None

    def cond(self, model_obj):
        """What to hand the sampler as the prompt: the cached mel [1, n, mel] when the model object can compute one, else the wave."""
        if not hasattr(model_obj, "cond_mel"):
            return self.audio
        if self.mel is None:
            self.mel = model_obj.cond_mel(self.audio)
        return self.mel


def _plan_request(ref_audio, ref_text, gen_text_batches, target_rms, speed, fix_duration, device, tokenizer):
    """Host prologue of infer_batch_process for one request (F/infer/utils_infer.py:423-454): prepared reference wave, its rms,
    the frame count the reference strips afterwards, and one sampling unit per text chunk."""
    voice = ref_audio if isinstance(ref_audio, PreparedVoice) else PreparedVoice(ref_audio, target_rms, device)
    if len(ref_text[-1].encode("utf-8")) == 1:
        ref_text = ref_text + " "
    return voice, plan_units(ref_text, gen_text_batches, voice.ref_frames, speed, fix_duration, tokenizer)


def _vocode_and_join(mels, ref_frames, rms, vocoder, mel_spec_type, target_rms, cross_fade_duration):
    """Tail of infer_batch_process (F/infer/utils_infer.py:468-524): strip the reference frames, vocode, restore the rms, cross-fade."""
    if mel_spec_type not in ("vocos", "bigvgan"):
        raise ValueError(mel_spec_type)
    waves, specs = [], []
    for mel in mels:
        spec = mel.to(torch.float32)[ref_frames:, :].t()[None]                     # [1, mel, T] (:468-470)
        wave = vocoder.decode(spec) if mel_spec_type == "vocos" else vocoder(spec)
        if rms < target_rms:
            wave = wave * rms / target_rms
        waves.append(wave.squeeze().cpu().numpy())
        specs.append(spec[0].cpu().numpy())
    return cross_fade_concat(waves, cross_fade_duration), target_sample_rate, np.concatenate(specs, axis=1)


def infer_batch_process(ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type="vocos", progress=None,
                        target_rms=0.1, cross_fade_duration=0.15, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1,
                        speed=1, fix_duration=None, device=None, tokenizer=text_to_tokens):
    """F/infer/utils_infer.py:406-524, same signature and return triple.

    The reference loops over the chunks and calls `sample()` / the vocoder once per chunk with batch 1.  The chunks are independent
    units, so here they are planned first and sampled in ONE `sample_units()` call when the model object offers it (F5HipModel: all
    chunks packed back to back, each with the reference's batch-1 semantics, the reference-audio mel computed once instead of once
    per chunk); any other object with the reference's `.sample()` is driven chunk by chunk like the reference does."""
    voice, units = _plan_request(ref_audio, ref_text, gen_text_batches, target_rms, speed, fix_duration, device, tokenizer)
    knobs = dict(steps=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef)
    if hasattr(model_obj, "sample_units"):
        mels = model_obj.sample_units(voice.cond(model_obj), units, **knobs)        # list of [frames_i, mel] incl. the reference frames
    else:
        mels = [model_obj.sample(cond=voice.audio, text=[tokens], duration=frames, **knobs)[0][0] for tokens, frames in units]
    return _vocode_and_join(mels, voice.ref_frames, voice.rms, vocoder, mel_spec_type, target_rms, cross_fade_duration)


def infer_requests(requests, model_obj, vocoder, mel_spec_type=mel_spec_type, target_rms=target_rms,
                   cross_fade_duration=cross_fade_duration, nfe_step=nfe_step, cfg_strength=cfg_strength,
                   sway_sampling_coef=sway_sampling_coef, speed=speed, fix_duration=fix_duration, device=None, tokenizer=text_to_tokens):
    """Several `infer_process()` calls as ONE sampler batch: `requests` = [(ref_audio, ref_text, gen_text)], each with its own
    reference voice (a path, a (wave, sr) pair or a `PreparedVoice`); returns one (wave, sample_rate, spectrogram) triple per request, each what `infer_process` returns for
    that request alone: units keep the reference's batch-1 semantics (no padding against each other, no shared mask), and noise is drawn
    unit by unit in request order, i.e. the draws of the sequential calls.  Bit for bit this holds when the model handle runs the
    shape-invariant attention arithmetic (`F5HipModel(attn_shape_invariant=True)`, what `serve.TTSManager.load` sets); in the default
    mode a lone unit and the same unit inside a batch may take different attention kernels (same values to the last bits per launch, but
    in the mixed GEMM mode last-bit differences grow to that mode's rounding-noise floor, 4.4e-4 rms after two Euler steps:
    `profiles/r03_attn_mode_tapdiff.txt`; either result is within the 1e-3 bound of the reference).  This is what the serving queue (`serve.MicroBatcher`) and the
    multi-voice front-end hand to the GPU: the chunks of all waiting requests are packed back to back in one library call."""
    plans, flat_units, flat_cond, flat_audio = [], [], [], []
    for ref_audio, ref_text, gen_text in requests:
        voice = ref_audio if isinstance(ref_audio, PreparedVoice) else PreparedVoice(ref_audio, target_rms, device)
        max_chars = int(len(ref_text.encode("utf-8")) / voice.seconds * (25 - voice.seconds))                 # utils_infer.py:379
        voice, units = _plan_request(voice, ref_text, chunk_text(gen_text, max_chars=max_chars), target_rms, speed, fix_duration, device, tokenizer)
        plans.append((voice, len(units)))
        flat_units += units
        flat_cond += [voice.cond(model_obj)] * len(units)
        flat_audio += [voice.audio] * len(units)
    knobs = dict(steps=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef)
    if hasattr(model_obj, "sample_units"):
        mels = model_obj.sample_units(flat_cond, flat_units, **knobs)
    else:
        mels = [model_obj.sample(cond=a, text=[tokens], duration=frames, **knobs)[0][0] for a, (tokens, frames) in zip(flat_audio, flat_units)]
    out, k = [], 0
    for voice, n in plans:
        out.append(_vocode_and_join(mels[k:k + n], voice.ref_frames, voice.rms, vocoder, mel_spec_type, target_rms, cross_fade_duration))
        k += n
    return out


_VOICE_TAG = re.compile(r"\[(\w+)\]")


def split_voice_tags(text_gen, voices):
    """Multi-voice script -> [(voice, text)] (F/infer/infer_cli.py:181-197): the text is cut in front of every `[tag]`; a piece without a
    tag, or with a tag that is not in `voices`, is spoken by "main"; empty pieces are dropped; the tag itself is not spoken."""
    pieces = []
    for piece in re.split(r"(?=\[\w+\])", text_gen):
        if not piece.strip():
            continue
        m = _VOICE_TAG.match(piece)
        voice = m.group(1) if m and m.group(1) in voices else "main"
        text = _VOICE_TAG.sub("", piece).strip()
        if text:   # (a tag with nothing behind it would hand the reference an empty gen_text; dropped here)
            pieces.append((voice, text))
    return pieces


def infer_multi_voice(text_gen, voices, model_obj, vocoder, **kw):
    """The multi-voice loop of the reference's CLI (F/infer/infer_cli.py:181-208): `voices` = {"main": {"ref_audio": path | (wave, sr),
    "ref_text": str}, "<tag>": {...}}; every `[tag]` piece is synthesized with its voice and the pieces are concatenated (no
    cross-fade between voices, like the reference).  All pieces go to the GPU as ONE `infer_requests` batch instead of one
    `infer_process` call after the other.  Returns (wave, sample_rate, [spectrogram per piece])."""
    if "main" not in voices:
        raise ValueError('voices needs a "main" entry')
    pieces = split_voice_tags(text_gen, voices)
    if not pieces:
        raise ValueError("nothing to synthesize")
    res = infer_requests([(voices[v]["ref_audio"], voices[v]["ref_text"], t) for v, t in pieces], model_obj, vocoder, **kw)
    return np.concatenate([w for w, _, _ in res]), target_sample_rate, [s for _, _, s in res]
