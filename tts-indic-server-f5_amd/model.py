"""Host-side mirror of the reference's model objects for the inference path.

`F5HipModel` stands where the reference passes `model_obj` (a `CFM` wrapping a `DiT`):
`sample()` has the signature and semantics of `CFM.sample` (F/model/cfm.py:82-210) and
`transformer_forward()` those of `DiT.forward` (F/model/backbones/dit.py:130-163).  All arithmetic runs in
libf5hip (HIP kernels); torch is used only to own device buffers and the stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, torch_ops
from .tokenizer import list_str_to_idx


@dataclass(frozen=True)
class DiTArch:
    """model.arch of F/configs/F5TTS_*_train.yaml:24-30."""
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    ff_mult: int = 2
    text_dim: int = 512
    conv_layers: int = 4
    mel_dim: int = 100
    text_num_embeds: int = 2545


F5TTS_BASE = DiTArch()
F5TTS_SMALL = DiTArch(dim=768, depth=18, heads=12)


@dataclass(frozen=True)
class UNetTArch:
    """model.arch of F/configs/E2TTS_*_train.yaml:24-28: flat-UNet transformer, text_dim = mel_dim, no text conv."""
    dim: int = 1024
    depth: int = 24
    heads: int = 16
    ff_mult: int = 4
    mel_dim: int = 100
    text_num_embeds: int = 2545

    @property
    def text_dim(self):
        return self.mel_dim

    @property
    def conv_layers(self):
        return 0


E2TTS_BASE = UNetTArch()
E2TTS_SMALL = UNetTArch(dim=768, depth=20, heads=12)


@dataclass(frozen=True)
class MMDiTArch:
    """MMDiT.__init__ arguments (F/model/backbones/mmdit.py:84-95): dual-stream blocks over the audio frames and the text tokens, joint
    attention, the last block context-pre-only; the text is embedded at `dim` (no ConvNeXt).  No YAML of the reference uses it."""
    dim: int = 512
    depth: int = 16
    heads: int = 16
    ff_mult: int = 2
    mel_dim: int = 100
    text_num_embeds: int = 256

    @property
    def text_dim(self):
        return self.dim

    @property
    def conv_layers(self):
        return 0


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _ptr(t):
    if isinstance(t, np.ndarray):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(t.data_ptr())


class F5HipModel:
    def __init__(self, arch: DiTArch | UNetTArch | MMDiTArch, state_dict: dict, vocab_char_map: dict | None = None, gemm_planes: int = 3,
                 device: str | torch.device = "cuda:0", mel_spec_type: str = "vocos", odeint_kwargs: dict | None = None,
                 attn_shape_invariant: bool | None = None):
        # odeint_kwargs: CFM's constructor argument (F/model/cfm.py:37-41), dict(method="euler") by default; "midpoint" is the other
        # fixed-grid solver the reference names.  Adaptive torchdiffeq solvers are not offered.
        self.odeint_kwargs = dict(odeint_kwargs) if odeint_kwargs is not None else dict(method="euler")
        method = self.odeint_kwargs.get("method", "euler")
        if method not in ("euler", "midpoint") or set(self.odeint_kwargs) - {"method"}:
            raise ValueError(f"odeint_kwargs={self.odeint_kwargs!r}: only method='euler' or 'midpoint' on the fixed grid is supported")
        self.arch = arch
        self.device = torch.device(device)
        self.vocab_char_map = vocab_char_map
        self.mel_spec_type = mel_spec_type
        self.num_channels = arch.mel_dim
        self.gemm_planes = gemm_planes   # 3 = mixed parity mode (default), 2 = bf16x3 everywhere, 1 = plain bf16 (include/f5hip.h)
        self._lib = _lib.lib()
        if self.device.type != "cuda":
            raise _lib.F5HipError("F5HipModel needs a HIP device (no CPU fallback)")
        torch.cuda.set_device(self.device)
        cfg = _lib.DitConfig(arch.dim, arch.depth, arch.heads, arch.ff_mult, arch.text_dim, arch.conv_layers,
                             arch.mel_dim, arch.text_num_embeds, gemm_planes, 1 if isinstance(arch, UNetTArch) else (2 if isinstance(arch, MMDiTArch) else 0))
        self._h = self._lib.f5hip_dit_create(C.byref(cfg))
        if not self._h:
            raise _lib.F5HipError("f5hip_dit_create: " + self._lib.f5hip_last_error().decode())
        # reference checkpoint keys: strip the EMA prefix like load_checkpoint does (F/infer/utils_infer.py:198-202)
        for k, v in state_dict.items():
            k = k.replace("ema_model.", "")
            if not k.startswith("transformer."):
                continue
            a = np.ascontiguousarray(v.detach().to(torch.float32).cpu().numpy())
            _lib.check(self._lib.f5hip_dit_load_param(self._h, k.encode(), _ptr(a), a.size), "load_param " + k)
        _lib.check(self._lib.f5hip_dit_finalize(self._h), "f5hip_dit_finalize")
        _lib.check(self._lib.f5hip_dit_set_ode_method(self._h, 1 if method == "midpoint" else 0), "f5hip_dit_set_ode_method")
        self.set_attention_shape_invariant(attn_shape_invariant)

    def set_attention_shape_invariant(self, on: bool | None):
        """This handle's attention arithmetic (include/f5hip.h): True = a sequence's output does not depend on what it is batched with,
        False = the fastest kernel per launch shape, None = follow the process default (f5hip_set_attention_shape_invariant)."""
        self.attn_shape_invariant = on
        _lib.check(self._lib.f5hip_dit_set_attention_shape_invariant(self._h, -1 if on is None else int(bool(on))), "f5hip_dit_set_attention_shape_invariant")

    def set_profiling(self, enabled: bool):
        """HIP-event timing of this handle's launches, per kernel class (its own spans and totals: other handles are not counted)."""
        _lib.check(self._lib.f5hip_dit_set_profiling(self._h, int(bool(enabled))), "f5hip_dit_set_profiling")

    def get_profile(self) -> dict:
        out = {}
        for cls in ("gemm", "attn", "ln", "other"):
            ms, n = C.c_double(0), C.c_int64(0)
            _lib.check(self._lib.f5hip_dit_get_profile(self._h, cls.encode(), C.byref(ms), C.byref(n)), "f5hip_dit_get_profile")
            out[cls] = {"total_ms": ms.value, "launches": n.value}
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.f5hip_dit_destroy(h)

    def eval(self):
        return self

    # ------------------------------------------------------------------ DiT.forward
    def transformer_forward(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None, n_blocks=-1):
        """DiT.forward for x/cond [b, n, mel] (device fp32), text int [b, nt], scalar time.  With `mask`
        ([b, n] bool) the reference's padded-batch semantics are reproduced (key-padding + zeroed rows)."""
        b, n, mel = x.shape
        x = x.to(self.device, torch.float32).contiguous()
        cond = cond.to(self.device, torch.float32).contiguous()
        text = _i32(text.cpu().numpy() if isinstance(text, torch.Tensor) else text).reshape(b, -1)
        seq_len = _i32([n] * b)
        kv_len = _i32(mask.sum(-1).cpu().numpy()) if mask is not None else seq_len
        da = np.full(b, 1 if drop_audio_cond else 0, dtype=np.uint8)
        dt = np.full(b, 1 if drop_text else 0, dtype=np.uint8)
        width = mel if n_blocks < 0 else self.arch.dim
        out = torch.empty(b, n, width, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.f5hip_dit_forward(
            self._h, b, _ptr(seq_len), _ptr(kv_len), _ptr(x), _ptr(cond), _ptr(text), text.shape[1], float(time),
            _ptr(da), _ptr(dt), n_blocks, _ptr(out) if n_blocks < 0 else None, _ptr(out) if n_blocks >= 0 else None,
            _lib.current_stream_ptr()), "f5hip_dit_forward")
        return out

    def read_tap(self, name: str, rows: int, width: int):
        out = torch.empty(rows, width, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.f5hip_dit_read_tap(self._h, name.encode(), _ptr(out), out.numel(), _lib.current_stream_ptr()),
                   "f5hip_dit_read_tap")
        return out

    # ------------------------------------------------------------------ CFM.sample
    def cond_mel(self, audio):
        """The mel front-end CFM.sample applies to a raw-wave `cond` (F/model/cfm.py:103-106, modules.py:123-143): [b, nw] -> [b, n, mel].
        infer_batch_process computes it once per request and hands the mel to every chunk (the reference recomputes it per chunk)."""
        from .mel import mel_spectrogram, mel_spectrogram_bigvgan
        fe = mel_spectrogram_bigvgan if self.mel_spec_type == "bigvgan" else mel_spectrogram
        cond = fe(audio.to(self.device, torch.float32)).permute(0, 2, 1)
        assert cond.shape[-1] == self.num_channels
        return cond

    @torch.no_grad()
    def sample_units(self, audio, units, *, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=None):
        """Independent sampling units in ONE sampler call: the text chunks of one request (independent `sample()` calls in the
        reference, F/infer/utils_infer.py:441-466), or the chunks of several requests with different voices (`infer.infer_requests`).
        `audio`: the reference wave [1, nw] (or its mel [1, n, mel]) shared by all units, or a list with one such tensor per unit;
        `units` = [(tokens, frames)].  Returns one [frames_i, mel] tensor per unit.  Every unit keeps batch-1 semantics with its own
        prompt length (`lens`), and noise is drawn unit by unit in order, i.e. the same draws the reference's sequential calls make
        from the global generator."""
        b = len(units)
        frames = torch.tensor([int(f) for _, f in units], dtype=torch.long)
        lens = None
        if isinstance(audio, (list, tuple)):
            assert len(audio) == b, "one reference per unit"
            mels, cache = [], {}
            for a in audio:   # the mel of a voice is computed once however many units share it
                if id(a) not in cache:
                    cache[id(a)] = (self.cond_mel(a) if a.ndim == 2 else a.to(self.device, torch.float32))[0]
                mels.append(cache[id(a)])
            lens = torch.tensor([m.shape[0] for m in mels], dtype=torch.long)
            cond = torch.nn.utils.rnn.pad_sequence(mels, batch_first=True)
        else:
            cond = (self.cond_mel(audio) if audio.ndim == 2 else audio.to(self.device, torch.float32)).expand(b, -1, -1)
        out, _ = self.sample(cond, [t for t, _ in units], frames, lens=lens, steps=steps, cfg_strength=cfg_strength,
                             sway_sampling_coef=sway_sampling_coef, seed=seed)
        # (sample() raises a duration to lens + 1 like the reference does, cfm.py:136: the rows of unit i are its FINAL duration)
        return [out[i, :self._last_min_frames[i]] for i in range(b)]

    @torch.no_grad()
    def sample(self, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None,
               seed=None, max_duration=4096, vocoder=None, no_ref_audio=False, duplicate_test=False, t_inter=0.1,
               edit_mask=None, y0=None, padded_batch=False):
        """CFM.sample (F/model/cfm.py:82-210).  Returns (out [b, n, mel] on the device, None): the trajectory is
        not materialised (its only in-tree consumer drops it, F/infer/utils_infer.py:459).

        Batch semantics.  Default: every item is sampled with the reference's batch-1 semantics (mask=None, no padding) -- what
        `infer_batch_process` uses, and independent of the batch composition.  `padded_batch=True` reproduces what the reference
        computes when it is handed b > 1 items itself (cfm.py:151-154: every item padded to the longest, key-padding mask, zeroed
        attention rows, unmasked convolutions over the padding), padded rows included.
        `y0` ([b, n, mel] or list of [dur_i, mel]; host or device) overrides the noise, which is otherwise drawn exactly like the
        reference's CPU path (per item `torch.manual_seed(seed)`; `torch.randn(dur, mel)` from the global CPU generator)."""
        if cond.ndim == 2:   # raw wave -> mel (cfm.py:103-106) with the extractor of mel_spec_type (modules.py:123-126)
            cond = self.cond_mel(cond)
        cond = cond.to(self.device, torch.float32)
        batch, cond_seq_len = cond.shape[:2]
        if lens is None:
            lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
        lens = lens.cpu()
        if isinstance(text, list):
            assert self.vocab_char_map is not None, "string text needs a vocab_char_map"
            text = list_str_to_idx(text, self.vocab_char_map)
            assert text.shape[0] == batch
        text = text.cpu()
        text_lens = (text != -1).sum(dim=-1)
        lens = torch.maximum(text_lens, lens)                                     # cfm.py:123-125
        cond_mask = torch.arange(int(lens.amax()))[None, :] < lens[:, None]       # lens_to_mask
        if edit_mask is not None:
            cond_mask = cond_mask & edit_mask.cpu()
        if isinstance(duration, int):
            duration = torch.full((batch,), duration, dtype=torch.long)
        duration = torch.maximum(lens + 1, duration.cpu()).clamp(max=max_duration)  # cfm.py:136-137
        nmax = int(duration.amax())
        test_cond = None
        if duplicate_test:   # cfm.py:139-141: the prompt mel repeated once right behind itself
            test_cond = torch.nn.functional.pad(cond, (0, 0, cond_seq_len, nmax - 2 * cond_seq_len), value=0.0)
        cond = torch.nn.functional.pad(cond, (0, 0, 0, nmax - cond_seq_len), value=0.0)
        cond_mask = torch.nn.functional.pad(cond_mask, (0, nmax - cond_mask.shape[-1]), value=False)

        durs = [int(d) for d in duration]
        self._last_min_frames = durs
        padded = bool(padded_batch) and batch > 1
        lay = [nmax] * batch if padded else durs          # rows laid out per item

        # noise (cfm.py:181-186): per item randn(dur_i), zero padded to the laid-out length
        ys = []
        for i, dur in enumerate(durs):
            if y0 is None:
                if seed is not None:
                    torch.manual_seed(seed)
                yi = torch.randn(dur, self.num_channels).to(self.device)
            else:
                yi = y0[i][:dur].to(self.device, torch.float32)
            if lay[i] > dur:
                yi = torch.nn.functional.pad(yi, (0, 0, 0, lay[i] - dur))
            ys.append(yi)

        t_start = 0.0
        if duplicate_test:   # cfm.py:190-194
            t_start = float(t_inter)
            ys = [(1 - t_start) * ys[i] + t_start * test_cond[i, :lay[i]] for i in range(batch)]
            steps = int(steps * (1 - t_start))
        t = torch.linspace(t_start, 1, steps + 1, dtype=torch.float32)            # cfm.py:196-198
        if sway_sampling_coef is not None:
            t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)

        cond_packed = torch.cat([cond[i, :lay[i]] for i in range(batch)], dim=0).contiguous()
        mask_packed = np.ascontiguousarray(
            torch.cat([cond_mask[i, :lay[i]] for i in range(batch)]).numpy().astype(np.uint8))
        y0_packed = torch.cat(ys, dim=0).contiguous()
        out_packed = torch.empty_like(y0_packed)
        text_np = _i32(text.numpy())
        tg = np.ascontiguousarray(t.numpy().astype(np.float32))
        d_np, kv_np = _i32(lay), _i32(durs)
        if torch_ops.load():   # the TORCH_LIBRARY operator over the same C entry point (csrc/torch_ops.cpp)
            try:
                out_packed = torch_ops.ops().cfm_sample(int(self._h), torch.from_numpy(d_np), torch.from_numpy(kv_np) if padded else None, cond_packed,
                                                        torch.from_numpy(mask_packed), torch.from_numpy(text_np), y0_packed, torch.from_numpy(tg), float(cfg_strength))
            except RuntimeError as e:
                raise _lib.F5HipError(str(e).split("\n")[0]) from None
        else:
            _lib.check(self._lib.f5hip_cfm_sample_masked(
                self._h, batch, _ptr(d_np), _ptr(kv_np) if padded else None, _ptr(cond_packed), _ptr(mask_packed), _ptr(text_np),
                text_np.shape[1], _ptr(y0_packed), _ptr(tg), steps, float(cfg_strength), _ptr(out_packed), _lib.current_stream_ptr()),
                "f5hip_cfm_sample")
        if all(n == nmax for n in lay):
            out = out_packed.view(batch, nmax, self.num_channels)
        else:
            out = torch.zeros(batch, nmax, self.num_channels, device=self.device, dtype=torch.float32)
            o = 0
            for i in range(batch):
                out[i, :lay[i]] = out_packed[o:o + lay[i]]
                o += lay[i]
        if no_ref_audio:   # cfm.py:157-158: the final overwrite then copies zeros
            out = torch.where(cond_mask[..., None].to(self.device), torch.zeros_like(out), out)
        if vocoder is not None:
            out = vocoder(out.permute(0, 2, 1))
        return out, None
