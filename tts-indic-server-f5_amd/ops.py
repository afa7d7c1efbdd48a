"""Python face of the per-kernel unit ops of the C ABI (include/f5hip.h): one production HIP kernel each, fp32 torch
tensors on the HIP device in and out.  Used by the per-kernel parity tests and the timing tools; not on the hot path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

ACT = {"none": 0, "gelu_tanh": 1, "gelu_erf": 2, "mish": 3, "silu": 4}


def _p(t):
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(t.data_ptr())


def _f32(t, dev):
    return None if t is None else t.to(dev, torch.float32).contiguous()


def gemm(a, w, bias=None, *, prec=3, act="none", mul=None, res=None, row_keep=None, out16=False, w_copies=1, iters=0):
    """out = (act(a @ w.T + bias), masked rows zeroed) * mul + res.  Returns (out, avg_us); out is fp32 [M, N], or the fp16 plane."""
    dev = a.device
    M, K = a.shape
    N = w.shape[0]
    a, w, bias, mul, res = (_f32(t, dev) for t in (a, w, bias, mul, res))
    out = torch.empty(M, N, device=dev, dtype=torch.float16 if out16 else torch.float32)
    keep = None if row_keep is None else np.ascontiguousarray(row_keep.cpu().numpy().astype(np.uint8))
    us = C.c_double(0.0)
    _lib.check(_lib.lib().f5hip_op_gemm(M, N, K, _p(a), _p(w), _p(bias), prec, ACT[act], _p(mul), _p(res), _p(keep),
                                        None if out16 else _p(out), _p(out) if out16 else None, w_copies, iters, C.byref(us),
                                        _lib.current_stream_ptr()), "f5hip_op_gemm")
    return out, us.value


def qkv(a, w, bias, row_pos, *, prec=3, iters=0):
    """Fused QKV projection + epilogue.  Returns (q [M, D] (already scaled by log2(e) / 8), k [M, D], v [M, D]) as fp32 views of the fp16 outputs, avg_us."""
    dev = a.device
    M, D = a.shape
    M_pad = (M + 127) // 128 * 128
    a, w, bias = (_f32(t, dev) for t in (a, w, bias))
    qk = torch.zeros(M_pad, 2 * D, device=dev, dtype=torch.float16)
    vt = torch.zeros(D, M_pad, device=dev, dtype=torch.float16)
    pos = np.ascontiguousarray(np.asarray(row_pos, dtype=np.int32))
    us = C.c_double(0.0)
    _lib.check(_lib.lib().f5hip_op_qkv(M, D, _p(a), _p(w), _p(bias), _p(pos), prec, _p(qk), _p(vt), iters, C.byref(us),
                                       _lib.current_stream_ptr()), "f5hip_op_qkv")
    # V^T keeps the tokens of every aligned group of 16 in the order 0-3, 8-11, 4-7, 12-15 (csrc/common.h vt_col): undo it for the caller
    t = torch.arange(M_pad, device=dev)
    col = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    return qk[:M, :D].float(), qk[:M, D:].float(), vt[:, col[:M]].t().float(), us.value


def layernorm(x, scale, shift, *, gain_off=1.0, eps=1e-6, rms=False):
    dev = x.device
    M, D = x.shape
    x, scale, shift = (_f32(t, dev) for t in (x, scale, shift))
    out = torch.empty_like(x)
    _lib.check(_lib.lib().f5hip_op_layernorm(M, D, _p(x), _p(scale), _p(shift), float(gain_off), float(eps), int(rms), _p(out),
                                             _lib.current_stream_ptr()), "f5hip_op_layernorm")
    return out


def attention(q, k, v, seq_len, kv_len=None, *, heads, impl=3, iters=0):
    """softmax(q k^T / 8 + key mask) v per (sequence, head); q / k / v fp32 [sum(seq_len), 64 * heads] packed.  Returns (out, avg_us)."""
    dev = q.device
    q, k, v = (_f32(t, dev) for t in (q, k, v))
    out = torch.empty_like(q)
    sl = np.ascontiguousarray(np.asarray(seq_len, dtype=np.int32))
    kl = None if kv_len is None else np.ascontiguousarray(np.asarray(kv_len, dtype=np.int32))
    us = C.c_double(0.0)
    _lib.check(_lib.lib().f5hip_op_attention(len(sl), _p(sl), _p(kl), heads, _p(q), _p(k), _p(v), _p(out), impl, iters, C.byref(us),
                                             _lib.current_stream_ptr()), "f5hip_op_attention")
    return out, us.value


def conv1d(x, weight, bias=None, res=None, *, batch, valid, dilation=1, prec=2, impl=5, iters=0, stamps=False):
    """One BigVGAN-style Conv1d over channel-last rows.  x fp32 [batch * P, c_in] (P = rows per sequence, a multiple of 128; `valid`
    rows of each are real, the rest is padding), weight [c_out, c_in, k] (nn.Conv1d layout), `same` zero padding at the sequence bounds.
    Returns (out [batch * P, c_out], avg_us, stamps | None); impl 0 = implicit GEMM, 5 = sliding-window kernel."""
    dev = x.device
    x = _f32(x, dev)
    c_out, c_in, k = weight.shape
    M = x.shape[0]
    P = M // batch
    w = np.ascontiguousarray(weight.detach().to(torch.float32).cpu().numpy())
    b = None if bias is None else np.ascontiguousarray(bias.detach().to(torch.float32).cpu().numpy())
    r = None if res is None else _f32(res, dev)
    out = torch.empty(M, c_out, dtype=torch.float32, device=dev)
    us = C.c_double(0.0)
    nblk = (M // 256) * ((c_out + 127) // 128)
    st = np.zeros((nblk, 16), dtype=np.uint64) if stamps else None
    _lib.check(_lib.lib().f5hip_op_conv1d(batch, P, valid, c_in, c_out, k, dilation, _p(x), _p(w), _p(b), _p(r), _p(out), prec, impl, iters,
                                          C.byref(us), _p(st), nblk, _lib.current_stream_ptr()), "f5hip_op_conv1d")
    return out, us.value, st


def joint_attention(q, k, v, x_len, c_len, x_kvlen=None, *, heads):
    """MMDiT joint attention: per sequence softmax(q [x ; c] k^T / 8 + mask on the padded audio keys) v over the concatenation of its audio
    rows and its text rows.  q / k / v fp32 [sum(x_len) + sum(c_len), 64 * heads]: all audio frames first, then all text tokens."""
    dev = q.device
    q, k, v = (_f32(t, dev) for t in (q, k, v))
    out = torch.empty_like(q)
    xl = np.ascontiguousarray(np.asarray(x_len, dtype=np.int32))
    cl = np.ascontiguousarray(np.asarray(c_len, dtype=np.int32))
    kl = None if x_kvlen is None else np.ascontiguousarray(np.asarray(x_kvlen, dtype=np.int32))
    _lib.check(_lib.lib().f5hip_op_joint_attention(len(xl), _p(xl), _p(kl), _p(cl), heads, _p(q), _p(k), _p(v), _p(out), _lib.current_stream_ptr()),
               "f5hip_op_joint_attention")
    return out
