"""fp32 CPU oracle of the DiT backbone + flow-matching sampler (TEST INFRASTRUCTURE ONLY).

Functional restatement over a plain ``state_dict`` (reference key names, i.e.
what ``load_checkpoint`` produces after stripping ``ema_model.``:
F/infer/utils_infer.py:195-209).  Shapes: b batch, n frames, d model dim.

Citations use F/ = /root/reference/src/server/f5_tts/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class DiTConfig:
    """model.arch block of F/configs/F5TTS_{Base,Small}_train.yaml:24-30."""
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    ff_mult: int = 2
    text_dim: int = 512
    conv_layers: int = 4
    mel_dim: int = 100          # F/infer/utils_infer.py:41
    text_num_embeds: int = 2545  # len(vocab.txt), F/infer/utils_infer.py:240-242
    dim_head: int = 64          # F/model/backbones/dit.py:100


F5_BASE = DiTConfig()
F5_SMALL = DiTConfig(dim=768, depth=18, heads=12)


# ----------------------------------------------------------------------------
# third-party leaves (restated; "parity unpinned" by the reference)
# ----------------------------------------------------------------------------

def rotary_freqs(seq_len: int, dim_head: int = 64, base: float = 10000.0) -> torch.Tensor:
    """x-transformers 2.2.8 RotaryEmbedding.forward_from_seq_len (call site F/model/backbones/dit.py:117,149).

    inv_freq_i = base^(-2i/dim), freqs[n, 2i] = freqs[n, 2i+1] = n * inv_freq_i (interleaved pairs).
    Returns [1, seq_len, dim_head] fp32; xpos scale is 1.
    """
    inv_freq = 1.0 / (base ** (torch.arange(0, dim_head, 2).float() / dim_head))
    t = torch.arange(seq_len).float()
    fr = torch.einsum("i,j->ij", t, inv_freq)
    fr = torch.stack((fr, fr), dim=-1).reshape(seq_len, dim_head)
    return fr.unsqueeze(0)


def rotate_half_interleaved(u: torch.Tensor) -> torch.Tensor:
    """x-transformers rotate_half: (u0,u1,u2,u3,..) -> (-u1,u0,-u3,u2,..)."""
    u = u.reshape(*u.shape[:-1], -1, 2)
    a, b = u.unbind(dim=-1)
    return torch.stack((-b, a), dim=-1).reshape(*u.shape[:-2], -1)


def apply_rotary(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """x-transformers 2.2.8 apply_rotary_pos_emb with scale=1 (call site F/model/modules.py:418-419).

    Only the first freqs.shape[-1] (=64) channels of the UN-split [b, n, d]
    tensor are rotated, i.e. head 0 only."""
    rot = freqs.shape[-1]
    tr, rest = t[..., :rot], t[..., rot:]
    tr = tr * freqs.cos() + rotate_half_interleaved(tr) * freqs.sin()
    return torch.cat((tr, rest), dim=-1)


def euler_odeint(fn, y0: torch.Tensor, t: torch.Tensor, keep_trajectory: bool = True):
    """torchdiffeq 0.2.5 odeint(method='euler') on the fixed grid t (call site F/model/cfm.py:200).

    y_{i+1} = y_i + (t_{i+1} - t_i) * fn(t_i, y_i); fn receives t_i as a 0-dim tensor."""
    ys = [y0]
    y = y0
    for i in range(t.numel() - 1):
        y = y + (t[i + 1] - t[i]) * fn(t[i], y)
        if keep_trajectory:
            ys.append(y)
    if keep_trajectory:
        return torch.stack(ys)
    return y


def midpoint_odeint(fn, y0: torch.Tensor, t: torch.Tensor, keep_trajectory: bool = True):
    """torchdiffeq 0.2.5 odeint(method='midpoint') on the fixed grid t (the alternative named at F/model/cfm.py:40 and passed
    through load_model(ode_method=...), F/infer/utils_infer.py:251).

    y_{i+1} = y_i + dt * fn(t_i + dt / 2, y_i + dt / 2 * fn(t_i, y_i)),  dt = t_{i+1} - t_i."""
    ys = [y0]
    y = y0
    for i in range(t.numel() - 1):
        dt = t[i + 1] - t[i]
        half = 0.5 * dt
        y = y + dt * fn(t[i] + half, y + half * fn(t[i], y))
        if keep_trajectory:
            ys.append(y)
    if keep_trajectory:
        return torch.stack(ys)
    return y


# ----------------------------------------------------------------------------
# reference-owned math (pinned by tests/golden fixtures)
# ----------------------------------------------------------------------------

def lens_to_mask(lens: torch.Tensor, length: int | None = None) -> torch.Tensor:
    """F/model/utils.py:42-47."""
    if length is None:
        length = int(lens.amax())
    return torch.arange(length)[None, :] < lens[:, None]


def sinus_time_embed(t: torch.Tensor, dim: int = 256, scale: float = 1000.0) -> torch.Tensor:
    """F/model/modules.py:154-161: [sin(1000 t f_k) || cos(..)], f_k = exp(-k ln(1e4)/(half-1))."""
    half = dim // 2
    f = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    e = scale * t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_embed(sd, t: torch.Tensor, p="transformer.time_embed.") -> torch.Tensor:
    """TimestepEmbedding, F/model/modules.py:648-658."""
    h = sinus_time_embed(t).to(t.dtype)
    h = F.linear(h, sd[p + "time_mlp.0.weight"], sd[p + "time_mlp.0.bias"])
    h = F.silu(h)
    return F.linear(h, sd[p + "time_mlp.2.weight"], sd[p + "time_mlp.2.bias"])


def text_pos_table(dim: int, end: int = 4096, theta: float = 10000.0) -> torch.Tensor:
    """precompute_freqs_cis, F/model/modules.py:196-207: [cos(pos w_j) || sin(pos w_j)]."""
    w = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.outer(torch.arange(end), w).float()
    return torch.cat([ang.cos(), ang.sin()], dim=-1)


def grn(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """GRN, F/model/modules.py:231-234 (L2 norm over the sequence axis, unmasked)."""
    g = torch.norm(x, p=2, dim=1, keepdim=True)
    nx = g / (g.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def convnext_v2_block(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """ConvNeXtV2Block.forward, F/model/modules.py:259-269 (dilation 1, k=7, pad 3)."""
    dim = x.shape[-1]
    r = x
    y = F.conv1d(x.transpose(1, 2), sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=dim)
    y = y.transpose(1, 2)
    y = F.layer_norm(y, (dim,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
    y = F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
    y = F.gelu(y)  # erf form (nn.GELU() default, modules.py:255)
    y = grn(y, sd[p + "grn.gamma"], sd[p + "grn.beta"])
    y = F.linear(y, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    return r + y


def text_embed(sd, cfg: DiTConfig, text: torch.Tensor, seq_len: int, drop_text: bool,
               p="transformer.text_embed.") -> torch.Tensor:
    """TextEmbedding.forward, F/model/backbones/dit.py:47-69."""
    ids = (text + 1)[:, :seq_len]
    ids = F.pad(ids, (0, seq_len - ids.shape[1]), value=0)
    if drop_text:
        ids = torch.zeros_like(ids)
    e = F.embedding(ids, sd[p + "text_embed.weight"])
    if cfg.conv_layers > 0:
        pos = torch.arange(seq_len).clamp(max=4095)  # get_pos_embed_indices, modules.py:210-219
        e = e + text_pos_table(cfg.text_dim)[pos][None]
        for i in range(cfg.conv_layers):
            e = convnext_v2_block(sd, f"{p}text_blocks.{i}.", e)
    return e


def conv_pos_embed(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """ConvPositionEmbedding.forward without mask, F/model/modules.py:171-190 (k=31, groups=16)."""
    y = x.permute(0, 2, 1)
    y = F.mish(F.conv1d(y, sd[p + "conv1d.0.weight"], sd[p + "conv1d.0.bias"], padding=15, groups=16))
    y = F.mish(F.conv1d(y, sd[p + "conv1d.2.weight"], sd[p + "conv1d.2.bias"], padding=15, groups=16))
    return y.permute(0, 2, 1)


def input_embed(sd, x, cond, temb, drop_audio_cond: bool, p="transformer.input_embed.") -> torch.Tensor:
    """InputEmbedding.forward, F/model/backbones/dit.py:81-87."""
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    h = F.linear(torch.cat((x, cond, temb), dim=-1), sd[p + "proj.weight"], sd[p + "proj.bias"])
    return conv_pos_embed(sd, p + "conv_pos_embed.", h) + h


def attention(sd, p: str, cfg: DiTConfig, x, mask, rope) -> torch.Tensor:
    """AttnProcessor.__call__, F/model/modules.py:399-449."""
    b, n, _ = x.shape
    q = F.linear(x, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(x, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(x, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    if rope is not None:
        q = apply_rotary(q, rope)
        k = apply_rotary(k, rope)
    h, dh = cfg.heads, cfg.dim_head
    q = q.view(b, n, h, dh).transpose(1, 2)
    k = k.view(b, n, h, dh).transpose(1, 2)
    v = v.view(b, n, h, dh).transpose(1, 2)
    am = None
    if mask is not None:
        am = mask[:, None, None, :].expand(b, h, n, n)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=am, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(b, n, h * dh)
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    if mask is not None:
        o = o.masked_fill(~mask[..., None], 0.0)
    return o


def dit_block(sd, p: str, cfg: DiTConfig, x, t, mask, rope) -> torch.Tensor:
    """DiTBlock.forward + AdaLayerNormZero.forward, F/model/modules.py:558-572, 285-290."""
    d = cfg.dim
    m = F.linear(F.silu(t), sd[p + "attn_norm.linear.weight"], sd[p + "attn_norm.linear.bias"])
    shift_a, scale_a, gate_a, shift_m, scale_m, gate_m = m.chunk(6, dim=1)
    h = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_a[:, None]) + shift_a[:, None]
    a = attention(sd, p + "attn.", cfg, h, mask, rope)
    x = x + gate_a[:, None] * a
    h = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_m[:, None]) + shift_m[:, None]
    f = F.linear(h, sd[p + "ff.ff.0.0.weight"], sd[p + "ff.ff.0.0.bias"])
    f = F.gelu(f, approximate="tanh")  # modules.py:556
    f = F.linear(f, sd[p + "ff.ff.2.weight"], sd[p + "ff.ff.2.bias"])
    return x + gate_m[:, None] * f


def dit_forward(sd, cfg: DiTConfig, x, cond, text, time, drop_audio_cond: bool, drop_text: bool,
                mask=None) -> torch.Tensor:
    """DiT.forward, F/model/backbones/dit.py:130-163 (long_skip_connection=False)."""
    b, n = x.shape[:2]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embed(sd, time)
    te = text_embed(sd, cfg, text, n, drop_text)
    h = input_embed(sd, x, cond, te, drop_audio_cond)
    rope = rotary_freqs(n, cfg.dim_head)
    for i in range(cfg.depth):
        h = dit_block(sd, f"transformer.transformer_blocks.{i}.", cfg, h, t, mask, rope)
    s = F.linear(F.silu(t), sd["transformer.norm_out.linear.weight"], sd["transformer.norm_out.linear.bias"])
    scale, shift = s.chunk(2, dim=1)  # (scale, shift) order: modules.py:308
    h = F.layer_norm(h, (cfg.dim,), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(h, sd["transformer.proj_out.weight"], sd["transformer.proj_out.bias"])


def sway_time_grid(steps: int, sway_sampling_coef: float | None, dtype=torch.float32) -> torch.Tensor:
    """F/model/cfm.py:196-198."""
    t = torch.linspace(0, 1, steps + 1, dtype=dtype)
    if sway_sampling_coef is not None:
        t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
    return t


def make_noise(durations, mel_dim: int, seed: int | None, y0=None) -> torch.Tensor:
    """F/model/cfm.py:181-186: per item randn(dur, mel_dim) from the global CPU generator, zero padded."""
    if y0 is not None:
        return y0
    ys = []
    for dur in durations:
        if seed is not None:
            torch.manual_seed(seed)
        ys.append(torch.randn(int(dur), mel_dim))
    return torch.nn.utils.rnn.pad_sequence(ys, padding_value=0, batch_first=True)


@torch.no_grad()
def cfm_sample(sd, cfg: DiTConfig, cond: torch.Tensor, text: torch.Tensor, duration, *, lens=None, steps=32,
               cfg_strength=1.0, sway_sampling_coef=None, seed=None, max_duration=4096, y0=None,
               edit_mask=None, no_ref_audio=False, forward_fn=None, keep_trajectory=True, method="euler"):
    """CFM.sample, F/model/cfm.py:82-210, for mel `cond` [b, n, 100] and int `text` [b, nt] (-1 padded).

    `forward_fn` lets a test substitute another backbone (UNetT oracle, or a
    precision-emulating variant); default is dit_forward.  Returns (out, trajectory)."""
    fwd = forward_fn or (lambda **kw: dit_forward(sd, cfg, **kw))
    cond = cond.float()
    b, cond_len = cond.shape[:2]
    if lens is None:
        lens = torch.full((b,), cond_len, dtype=torch.long)
    text_lens = (text != -1).sum(dim=-1)
    lens = torch.maximum(text_lens, lens)                                   # cfm.py:123-125
    cond_mask = lens_to_mask(lens)
    if edit_mask is not None:
        cond_mask = cond_mask & edit_mask
    if isinstance(duration, int):
        duration = torch.full((b,), duration, dtype=torch.long)
    duration = torch.maximum(lens + 1, duration).clamp(max=max_duration)    # cfm.py:136-137
    nmax = int(duration.amax())
    cond = F.pad(cond, (0, 0, 0, nmax - cond_len), value=0.0)
    cond_mask = F.pad(cond_mask, (0, nmax - cond_mask.shape[-1]), value=False)[..., None]
    step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
    mask = lens_to_mask(duration) if b > 1 else None                        # cfm.py:151-154
    if no_ref_audio:
        cond = torch.zeros_like(cond)

    def fn(t, x):
        pred = fwd(x=x, cond=step_cond, text=text, time=t, mask=mask, drop_audio_cond=False, drop_text=False)
        if cfg_strength < 1e-5:
            return pred
        null = fwd(x=x, cond=step_cond, text=text, time=t, mask=mask, drop_audio_cond=True, drop_text=True)
        return pred + (pred - null) * cfg_strength

    y0 = make_noise(duration, cfg.mel_dim, seed, y0)
    t = sway_time_grid(steps, sway_sampling_coef)
    if method not in ("euler", "midpoint"):
        raise ValueError(f"unknown ODE method {method!r}")
    traj = (euler_odeint if method == "euler" else midpoint_odeint)(fn, y0, t, keep_trajectory=keep_trajectory)
    last = traj[-1] if keep_trajectory else traj
    out = torch.where(cond_mask, cond, last)
    return out, (traj if keep_trajectory else None)


# ----------------------------------------------------------------------------
# UNetT (E2-TTS) backbone, F/model/backbones/unett.py:96-219
# ----------------------------------------------------------------------------

@dataclass(frozen=True)
class UNetTConfig:
    """model.arch of F/configs/E2TTS_{Base,Small}_train.yaml:24-28 (text_dim defaults to mel_dim, no text conv)."""
    dim: int = 1024
    depth: int = 24
    heads: int = 16
    ff_mult: int = 4
    mel_dim: int = 100
    text_num_embeds: int = 2545
    dim_head: int = 64
    # DiT-compatible fields used by the shared helpers
    text_dim: int = 100
    conv_layers: int = 0


E2_BASE = UNetTConfig()
E2_SMALL = UNetTConfig(dim=768, depth=20, heads=12)


def rms_norm(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """x-transformers 2.2.8 RMSNorm (call sites unett.py:17,135,144,161): F.normalize(x, dim=-1) * sqrt(dim) * g."""
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * g


def unett_forward(sd, cfg: UNetTConfig, x, cond, text, time, drop_audio_cond: bool, drop_text: bool, mask=None) -> torch.Tensor:
    """UNetT.forward, F/model/backbones/unett.py:164-219 (skip_connect_type="concat")."""
    b, n = x.shape[:2]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embed(sd, time)
    te = text_embed(sd, cfg, text, n, drop_text)
    h = input_embed(sd, x, cond, te, drop_audio_cond)
    h = torch.cat([t.unsqueeze(1), h], dim=1)                       # time token first (unett.py:184)
    if mask is not None:
        mask = F.pad(mask, (1, 0), value=True)
    rope = rotary_freqs(n + 1, cfg.dim_head)
    skips = []
    for i in range(cfg.depth):
        p = f"transformer.layers.{i}."
        if i < cfg.depth // 2:
            skips.append(h)
        else:
            h = F.linear(torch.cat((h, skips.pop()), dim=-1), sd[p + "0.weight"])
        h = attention(sd, p + "2.", cfg, rms_norm(h, sd[p + "1.g"]), mask, rope) + h
        f = F.linear(rms_norm(h, sd[p + "3.g"]), sd[p + "4.ff.0.0.weight"], sd[p + "4.ff.0.0.bias"])
        f = F.gelu(f, approximate="tanh")
        h = F.linear(f, sd[p + "4.ff.2.weight"], sd[p + "4.ff.2.bias"]) + h
    assert not skips
    h = rms_norm(h, sd["transformer.norm_out.g"])[:, 1:, :]
    return F.linear(h, sd["transformer.proj_out.weight"], sd["transformer.proj_out.bias"])


# ----------------------------------------------------------------------------
# MMDiT backbone, F/model/backbones/mmdit.py:83-146 + MMDiTBlock / JointAttnProcessor (F/model/modules.py:578-642, 456-536)
# ----------------------------------------------------------------------------

@dataclass(frozen=True)
class MMDiTConfig:
    """MMDiT.__init__ arguments (mmdit.py:84-95): text is embedded at `dim`, no text ConvNeXt, the last block is context-pre-only."""
    dim: int = 512
    depth: int = 16
    heads: int = 16
    ff_mult: int = 2
    text_num_embeds: int = 256
    mel_dim: int = 100
    dim_head: int = 64


def mmdit_text_embed(sd, cfg: MMDiTConfig, text: torch.Tensor, drop_text: bool, p="transformer.text_embed.") -> torch.Tensor:
    """TextEmbedding.forward, mmdit.py:37-52: ids + 1 (-1 padding -> filler 0), all ids 0 when dropped, + the absolute position table
    precompute_freqs_cis(dim, 1024) at positions 0 .. nt - 1 (get_pos_embed_indices: clamped below max_pos).  The whole nt is kept."""
    ids = text + 1
    if drop_text:
        ids = torch.zeros_like(ids)
    e = F.embedding(ids, sd[p + "text_embed.weight"])
    pos = torch.arange(ids.shape[1]).clamp(max=1023)
    return e + text_pos_table(cfg.dim, end=1024)[pos][None]


def mmdit_audio_embed(sd, x, cond, drop_audio_cond: bool, p="transformer.audio_embed.") -> torch.Tensor:
    """AudioEmbedding.forward, mmdit.py:64-70: Linear(cat(x, cond)) + ConvPositionEmbedding (no mask) + residual."""
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    h = F.linear(torch.cat((x, cond), dim=-1), sd[p + "linear.weight"], sd[p + "linear.bias"])
    return conv_pos_embed(sd, p + "conv_pos_embed.", h) + h


def joint_attention(sd, p: str, cfg: MMDiTConfig, x, c, mask, rope, c_rope, context_pre_only: bool):
    """JointAttnProcessor.__call__, modules.py:460-536: separate projections for the audio stream x and the text stream c, rotary on each
    with its OWN positions, softmax over the concatenated keys [x ; c] (padding mask on the x keys only), outputs split back."""
    b, n, _ = x.shape
    nt = c.shape[1]
    q = apply_rotary(F.linear(x, sd[p + "to_q.weight"], sd[p + "to_q.bias"]), rope)
    k = apply_rotary(F.linear(x, sd[p + "to_k.weight"], sd[p + "to_k.bias"]), rope)
    v = F.linear(x, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    qc = apply_rotary(F.linear(c, sd[p + "to_q_c.weight"], sd[p + "to_q_c.bias"]), c_rope)
    kc = apply_rotary(F.linear(c, sd[p + "to_k_c.weight"], sd[p + "to_k_c.bias"]), c_rope)
    vc = F.linear(c, sd[p + "to_v_c.weight"], sd[p + "to_v_c.bias"])
    h, dh = cfg.heads, cfg.dim_head
    Q = torch.cat([q, qc], dim=1).view(b, n + nt, h, dh).transpose(1, 2)
    K = torch.cat([k, kc], dim=1).view(b, n + nt, h, dh).transpose(1, 2)
    V = torch.cat([v, vc], dim=1).view(b, n + nt, h, dh).transpose(1, 2)
    am = None
    if mask is not None:
        am = F.pad(mask, (0, nt), value=True)[:, None, None, :].expand(b, h, n + nt, n + nt)   # no mask for the text keys
    o = F.scaled_dot_product_attention(Q, K, V, attn_mask=am, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(b, n + nt, h * dh)
    ox, oc = o[:, :n], o[:, n:]
    ox = F.linear(ox, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    if not context_pre_only:
        oc = F.linear(oc, sd[p + "to_out_c.weight"], sd[p + "to_out_c.bias"])
    if mask is not None:
        ox = ox.masked_fill(~mask[..., None], 0.0)
    return ox, oc


def _ff(sd, p: str, h: torch.Tensor) -> torch.Tensor:
    f = F.linear(h, sd[p + "ff.0.0.weight"], sd[p + "ff.0.0.bias"])
    f = F.gelu(f, approximate="tanh")   # MMDiTBlock: FeedForward(approximate="tanh"), modules.py:607,612
    return F.linear(f, sd[p + "ff.2.weight"], sd[p + "ff.2.bias"])


def mmdit_block(sd, p: str, cfg: MMDiTConfig, x, c, t, mask, rope, c_rope, context_pre_only: bool):
    """MMDiTBlock.forward, modules.py:614-642.  Returns (c, x); c is None behind the last (context-pre-only) block."""
    d = cfg.dim
    st = F.silu(t)
    mc = F.linear(st, sd[p + "attn_norm_c.linear.weight"], sd[p + "attn_norm_c.linear.bias"])
    if context_pre_only:
        c_scale, c_shift = mc.chunk(2, dim=1)                     # AdaLayerNormZero_Final: (scale, shift), modules.py:308
        norm_c = F.layer_norm(c, (d,), eps=1e-6) * (1 + c_scale[:, None]) + c_shift[:, None]
    else:
        c_shift_a, c_scale_a, c_gate_a, c_shift_m, c_scale_m, c_gate_m = mc.chunk(6, dim=1)
        norm_c = F.layer_norm(c, (d,), eps=1e-6) * (1 + c_scale_a[:, None]) + c_shift_a[:, None]
    mx = F.linear(st, sd[p + "attn_norm_x.linear.weight"], sd[p + "attn_norm_x.linear.bias"])
    x_shift_a, x_scale_a, x_gate_a, x_shift_m, x_scale_m, x_gate_m = mx.chunk(6, dim=1)
    norm_x = F.layer_norm(x, (d,), eps=1e-6) * (1 + x_scale_a[:, None]) + x_shift_a[:, None]
    ax, ac = joint_attention(sd, p + "attn.", cfg, norm_x, norm_c, mask, rope, c_rope, context_pre_only)
    if context_pre_only:
        c = None
    else:
        c = c + c_gate_a[:, None] * ac
        hc = F.layer_norm(c, (d,), eps=1e-6) * (1 + c_scale_m[:, None]) + c_shift_m[:, None]
        c = c + c_gate_m[:, None] * _ff(sd, p + "ff_c.", hc)
    x = x + x_gate_a[:, None] * ax
    hx = F.layer_norm(x, (d,), eps=1e-6) * (1 + x_scale_m[:, None]) + x_shift_m[:, None]
    x = x + x_gate_m[:, None] * _ff(sd, p + "ff_x.", hx)
    return c, x


def mmdit_forward(sd, cfg: MMDiTConfig, x, cond, text, time, drop_audio_cond: bool, drop_text: bool, mask=None) -> torch.Tensor:
    """MMDiT.forward, mmdit.py:115-146."""
    b, n = x.shape[:2]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embed(sd, time)
    c = mmdit_text_embed(sd, cfg, text, drop_text)
    h = mmdit_audio_embed(sd, x, cond, drop_audio_cond)
    rope, c_rope = rotary_freqs(n, cfg.dim_head), rotary_freqs(text.shape[1], cfg.dim_head)
    for i in range(cfg.depth):
        c, h = mmdit_block(sd, f"transformer.transformer_blocks.{i}.", cfg, h, c, t, mask, rope, c_rope, i == cfg.depth - 1)
    s = F.linear(F.silu(t), sd["transformer.norm_out.linear.weight"], sd["transformer.norm_out.linear.bias"])
    scale, shift = s.chunk(2, dim=1)
    h = F.layer_norm(h, (cfg.dim,), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(h, sd["transformer.proj_out.weight"], sd["transformer.proj_out.bias"])

