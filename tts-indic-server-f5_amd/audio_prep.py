"""Reference-audio pre-step of the reference's inference driver, restated without pydub.

Mirrors `preprocess_ref_audio_text` / `remove_silence_edges` (F/infer/utils_infer.py:263-350): clip a long reference
clip at a pause (>= 1 s below -50 dBFS, else >= 0.1 s below -40 dBFS, else a hard cut at 15 s), strip leading / trailing
silence (-42 dBFS), append 50 ms of silence, write a temporary 16-bit WAV, and normalise the end of the reference text
(". " rule).  Returns `(wav_path, ref_text)` like the reference, so `infer_process(*preprocess_ref_audio_text(...), ...)`
reads the same.

pydub (0.25.1) is a third-party dependency that is absent here; its published algorithms are restated on int16 numpy
arrays: millisecond slicing (`frame = int(ms * rate / 1000)`), `rms` = floor(sqrt(mean(x^2))) over all interleaved samples
(audioop.rms), `dBFS` = 20 log10(rms / 32768), `silence.detect_silence` / `detect_nonsilent` / `split_on_silence` /
`detect_leading_silence`.  PARITY UNPINNED by the reference (no fixture of it exists); pinned by known-answer tests on
synthetic tone / pause signals and against the stdlib `audioop.rms` (tests/test_host_glue.py).  Differences, explicit:
  * input is a 16-bit PCM WAV (stdlib `wave`); pydub would hand other containers to ffmpeg;
  * sample rates below 11 025 Hz are rejected (pydub would up-sample them to its 11 025 Hz "silent" segment's rate);
  * an empty `ref_text` raises: the reference transcribes with a Whisper pipeline that is not part of this path.
"""
from __future__ import annotations

import hashlib
import math
import tempfile
import wave as _wave

import numpy as np

_MAX_AMP = 32768.0   # AudioSegment.max_possible_amplitude for 16-bit samples


class PcmSegment:
    """The subset of pydub.AudioSegment the pre-step uses: int16 frames [n, channels] at `rate`, sliced in milliseconds."""

    def __init__(self, frames: np.ndarray, rate: int):
        frames = np.asarray(frames, dtype=np.int16)
        self.frames = frames.reshape(-1, 1) if frames.ndim == 1 else frames
        self.rate = int(rate)
        self._csq = None

    @classmethod
    def from_wav(cls, path: str) -> "PcmSegment":
        with _wave.open(path, "rb") as f:
            rate, ch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            raw = f.readframes(n)
        if sw != 2:
            raise ValueError("only 16-bit PCM WAV reference audio is supported")
        return cls(np.frombuffer(raw, dtype="<i2").reshape(-1, ch), rate)

    @classmethod
    def silent(cls, duration_ms: int, rate: int, channels: int = 1) -> "PcmSegment":
        return cls(np.zeros((int(rate * duration_ms / 1000.0), channels), dtype=np.int16), rate)

    def __len__(self) -> int:                      # pydub: round(1000 * frame_count / frame_rate)
        return int(round(1000.0 * self.frames.shape[0] / self.rate))

    def _frame(self, ms) -> int:                   # pydub _parse_position: int(ms * frame_rate / 1000.0), clipped
        return min(max(int(ms * (self.rate / 1000.0)), 0), self.frames.shape[0])

    def slice_ms(self, start_ms, end_ms) -> "PcmSegment":
        start_ms = max(0, min(start_ms, len(self)))
        end_ms = max(0, min(end_ms, len(self)))
        return PcmSegment(self.frames[self._frame(start_ms):self._frame(end_ms)], self.rate)

    def __add__(self, other: "PcmSegment") -> "PcmSegment":
        if other.rate != self.rate or other.frames.shape[1] != self.frames.shape[1]:
            raise ValueError("segments must share rate and channel count")
        return PcmSegment(np.concatenate([self.frames, other.frames]), self.rate)

    def _cum_squares(self) -> np.ndarray:          # exact in int64: 2^30 per sample
        if self._csq is None:
            sq = (self.frames.astype(np.int64) ** 2).sum(axis=1)
            self._csq = np.concatenate([[0], np.cumsum(sq)])
        return self._csq

    def rms_ms(self, start_ms, end_ms) -> int:
        """audioop.rms of the millisecond window: floor(sqrt(sum x^2 / n_samples)) over all interleaved samples."""
        a, b = self._frame(max(0, min(start_ms, len(self)))), self._frame(max(0, min(end_ms, len(self))))
        n = (b - a) * self.frames.shape[1]
        if n <= 0:
            return 0
        c = self._cum_squares()
        return int(math.sqrt(float(c[b] - c[a]) / n))

    @property
    def rms(self) -> int:
        return self.rms_ms(0, len(self) + 1)

    def dbfs_ms(self, start_ms, end_ms) -> float:
        r = self.rms_ms(start_ms, end_ms)
        return -math.inf if r == 0 else 20.0 * math.log10(r / _MAX_AMP)

    @property
    def duration_seconds(self) -> float:
        return self.frames.shape[0] / self.rate

    def export_wav(self, path: str) -> None:
        with _wave.open(path, "wb") as f:
            f.setnchannels(self.frames.shape[1]); f.setsampwidth(2); f.setframerate(self.rate)
            f.writeframes(np.ascontiguousarray(self.frames, dtype="<i2").tobytes())


def detect_silence(seg: PcmSegment, min_silence_len=1000, silence_thresh=-16, seek_step=1):
    """pydub.silence.detect_silence: [start_ms, end_ms] ranges whose every `min_silence_len` window has rms <= threshold."""
    seg_len = len(seg)
    if seg_len < min_silence_len:
        return []
    thresh = (10 ** (silence_thresh / 20.0)) * _MAX_AMP
    last = seg_len - min_silence_len
    starts = list(range(0, last + 1, seek_step))
    if last % seek_step:
        starts.append(last)
    silent = [i for i in starts if seg.rms_ms(i, i + min_silence_len) <= thresh]
    if not silent:
        return []
    ranges = []
    prev = silent[0]
    cur = prev
    for i in silent[1:]:
        if i != prev + seek_step and i > prev + min_silence_len:
            ranges.append([cur, prev + min_silence_len])
            cur = i
        prev = i
    ranges.append([cur, prev + min_silence_len])
    return ranges


def detect_nonsilent(seg: PcmSegment, min_silence_len=1000, silence_thresh=-16, seek_step=1):
    silent = detect_silence(seg, min_silence_len, silence_thresh, seek_step)
    n = len(seg)
    if not silent:
        return [[0, n]]
    if silent[0][0] == 0 and silent[0][1] == n:
        return []
    out, prev_end, end = [], 0, 0
    for start, end in silent:
        out.append([prev_end, start])
        prev_end = end
    if end != n:
        out.append([prev_end, n])
    if out[0] == [0, 0]:
        out.pop(0)
    return out


def split_on_silence(seg: PcmSegment, min_silence_len=1000, silence_thresh=-16, keep_silence=100, seek_step=1):
    """pydub.silence.split_on_silence (0.25.1): non-silent chunks padded by `keep_silence` ms, overlaps split at the midpoint."""
    ranges = [[s - keep_silence, e + keep_silence] for s, e in detect_nonsilent(seg, min_silence_len, silence_thresh, seek_step)]
    for a, b in zip(ranges, ranges[1:]):
        if b[0] < a[1]:
            a[1] = (a[1] + b[0]) // 2
            b[0] = a[1]
    return [seg.slice_ms(max(s, 0), min(e, len(seg))) for s, e in ranges]


def detect_leading_silence(seg: PcmSegment, silence_threshold=-50.0, chunk_size=10) -> int:
    trim = 0
    while seg.dbfs_ms(trim, trim + chunk_size) < silence_threshold and trim < len(seg):
        trim += chunk_size
    return min(trim, len(seg))


def remove_silence_edges(seg: PcmSegment, silence_threshold=-42) -> PcmSegment:
    """F/infer/utils_infer.py:263-277: leading silence in 10 ms chunks, trailing silence one millisecond at a time."""
    seg = seg.slice_ms(detect_leading_silence(seg, silence_threshold), len(seg) + 1)
    end = seg.duration_seconds
    for ms in range(len(seg) - 1, -1, -1):
        if seg.dbfs_ms(ms, ms + 1) > silence_threshold:
            break
        end -= 0.001
    return seg.slice_ms(0, int(end * 1000))


def _clip_at_pause(seg: PcmSegment, min_silence_len: int, silence_thresh: int, show_info, tag: str) -> PcmSegment:
    out = PcmSegment(np.zeros((0, seg.frames.shape[1]), dtype=np.int16), seg.rate)
    for chunk in split_on_silence(seg, min_silence_len=min_silence_len, silence_thresh=silence_thresh, keep_silence=1000, seek_step=10):
        if len(out) > 6000 and len(out + chunk) > 15000:
            show_info(f"Audio is over 15s, clipping short. ({tag})")
            break
        out = out + chunk
    return out


_ref_text_cache: dict = {}   # audio md5 -> transcription (the reference caches ASR output only; kept for interface parity)


def preprocess_ref_audio_text(ref_audio_orig: str, ref_text: str, clip_short: bool = True, show_info=print, device=None):
    """F/infer/utils_infer.py:282-350.  Returns (path of the processed temporary WAV, normalised ref_text)."""
    seg = PcmSegment.from_wav(ref_audio_orig)
    if seg.rate < 11025:
        raise ValueError("reference audio below 11025 Hz is not supported on this path")
    if clip_short:
        clipped = _clip_at_pause(seg, 1000, -50, show_info, "1")
        if len(clipped) > 15000:
            clipped = _clip_at_pause(seg, 100, -40, show_info, "2")
        seg = clipped
        if len(seg) > 15000:
            seg = seg.slice_ms(0, 15000)
            show_info("Audio is over 15s, clipping short. (3)")
    seg = remove_silence_edges(seg) + PcmSegment.silent(50, seg.rate, seg.frames.shape[1])
    with tempfile.NamedTemporaryFile(delete=False, suffix=".wav") as f:
        path = f.name
    seg.export_wav(path)
    with open(path, "rb") as f:
        audio_hash = hashlib.md5(f.read()).hexdigest()
    if not ref_text.strip():
        if audio_hash in _ref_text_cache:
            show_info("Using cached reference text...")
            ref_text = _ref_text_cache[audio_hash]
        else:
            raise NotImplementedError("empty ref_text: the reference transcribes the clip with a Whisper ASR pipeline, which is not on this path")
    if not ref_text.endswith(". ") and not ref_text.endswith("。"):
        ref_text += " " if ref_text.endswith(".") else ". "
    return path, ref_text
