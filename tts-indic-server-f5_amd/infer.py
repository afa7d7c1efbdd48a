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


def text_to_tokens(text_list):
    """Non-CJK subset of convert_char_to_pinyin (F/model/utils.py:140-177): translation table, then one token per
    character.  (jieba's word segmentation only changes the output for CJK text and for the
    space-before-a-Latin-word rule after CJK; neither applies to text without CJK characters.)"""
    out = []
    for text in text_list:
        text = text.translate(_CUSTOM_TRANS)
        if any("㄀" <= c <= "鿿" for c in text):
            raise NotImplementedError("CJK text needs the pinyin front-end (jieba/pypinyin), which is not on this path yet")
        out.append(list(text))
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


def infer_batch_process(ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type="vocos", progress=None,
                        target_rms=0.1, cross_fade_duration=0.15, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1,
                        speed=1, fix_duration=None, device=None, tokenizer=text_to_tokens):
    """F/infer/utils_infer.py:406-524."""
    audio, sr = ref_audio
    if audio.shape[0] > 1:
        audio = torch.mean(audio, dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < target_rms:
        audio = audio * target_rms / rms
    if sr != target_sample_rate:
        audio = resample_sinc_hann(audio, sr, target_sample_rate)
    if device is not None:
        audio = audio.to(device)

    generated_waves = []
    spectrograms = []
    if len(ref_text[-1].encode("utf-8")) == 1:
        ref_text = ref_text + " "
    for gen_text in gen_text_batches:
        final_text_list = tokenizer([ref_text + gen_text])
        ref_audio_len = audio.shape[-1] // hop_length
        if fix_duration is not None:
            duration = int(fix_duration * target_sample_rate / hop_length)
        else:
            ref_text_len = len(ref_text.encode("utf-8"))
            gen_text_len = len(gen_text.encode("utf-8"))
            duration = ref_audio_len + int(ref_audio_len / ref_text_len * gen_text_len / speed)
        generated, _ = model_obj.sample(cond=audio, text=final_text_list, duration=duration, steps=nfe_step,
                                        cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef)
        generated = generated.to(torch.float32)
        generated = generated[:, ref_audio_len:, :]
        generated_mel_spec = generated.permute(0, 2, 1)
        if mel_spec_type == "vocos":
            generated_wave = vocoder.decode(generated_mel_spec)
        elif mel_spec_type == "bigvgan":
            generated_wave = vocoder(generated_mel_spec)
        else:
            raise ValueError(mel_spec_type)
        if rms < target_rms:
            generated_wave = generated_wave * rms / target_rms
        generated_waves.append(generated_wave.squeeze().cpu().numpy())
        spectrograms.append(generated_mel_spec[0].cpu().numpy())

    if cross_fade_duration <= 0:
        final_wave = np.concatenate(generated_waves)
    else:
        final_wave = generated_waves[0]
        for i in range(1, len(generated_waves)):
            prev_wave, next_wave = final_wave, generated_waves[i]
            n = min(int(cross_fade_duration * target_sample_rate), len(prev_wave), len(next_wave))
            if n <= 0:
                final_wave = np.concatenate([prev_wave, next_wave])
                continue
            fade_out = np.linspace(1, 0, n)   # float64 ramps: the result is float64 from the 2nd chunk on
            fade_in = np.linspace(0, 1, n)
            mixed = prev_wave[-n:] * fade_out + next_wave[:n] * fade_in
            final_wave = np.concatenate([prev_wave[:-n], mixed, next_wave[n:]])
    return final_wave, target_sample_rate, np.concatenate(spectrograms, axis=1)
