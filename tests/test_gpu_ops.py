"""GPU parity of the per-kernel unit ops (include/f5hip.h "unit ops"): every hot kernel alone, through the C ABI, against a
plain fp64 torch reference of the same op evaluated on the operand values the kernel actually multiplies (fp16 / bf16 /
split-bf16 rounding of the inputs), so the tolerance only has to cover the fp32 accumulation order: 2e-5 relative rms.
Shapes are the transformer-block GEMMs of F5-TTS-Base at the BASELINE configs (M = 2816 rows = one 10 s utterance with both CFG
branches; F/model/modules.py:324-328,409-447) plus ragged / partial-tile / short-K edge cases."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
Q_SCALE = 0.125 * math.log2(math.e)   # csrc/common.h F5_Q_SCALE: softmax scale 1/8 and log2(e), folded into q by the QKV epilogue


def _counter(name):
    from tts_indic_server_f5_amd import _lib
    v = C.c_int64(0)
    _lib.check(_lib.lib().f5hip_get_counter(name.encode(), C.byref(v)), "get_counter")
    return v.value


def _reset_counters():
    from tts_indic_server_f5_amd import _lib
    _lib.check(_lib.lib().f5hip_get_counter(b"reset", None), "reset counters")


def _operand_values(x, prec):
    """fp64 tensors whose products sum to what the kernel multiplies: [(a_part, w_part_selector)]."""
    if prec == 3:
        return x.half().double(), None
    hi = x.bfloat16()
    if prec == 1:
        return hi.double(), None
    lo = (x - hi.float()).bfloat16()
    return hi.double(), lo.double()


def _ref_matmul(a, w, prec):
    ah, al = _operand_values(a, prec)
    wh, wl = _operand_values(w, prec)
    y = ah @ wh.T
    if prec == 2:   # bf16x3: hi*hi + hi*lo + lo*hi (the lo*lo term is dropped by design)
        y = y + ah @ wl.T + al @ wh.T
    return y


def _act(y, act):
    if act == "gelu_tanh":
        return torch.nn.functional.gelu(y, approximate="tanh")
    if act == "gelu_erf":
        return torch.nn.functional.gelu(y)
    if act == "silu":
        return torch.nn.functional.silu(y)
    if act == "mish":
        return torch.nn.functional.mish(y)
    return y


def _rel(got, ref):
    return ((got.double().cpu() - ref.cpu()).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30)).item()


CASES = [
    # (M, N, K, prec, act, bias, mul, res, row_keep, out16, expected counter)
    (2816, 1024, 1024, 3, "none", True, True, True, False, False, "gemm5_rb11"),      # attention out-projection, C2 (176 x 64 tiles)
    (2816, 1024, 2048, 3, "none", True, True, True, False, False, "gemm5_rb11"),      # FF2, C2
    (2816, 2048, 1024, 3, "gelu_tanh", True, False, False, False, True, "gemm5_wide"),  # FF1, C2: fp16 plane out (176 x 128 tiles)
    (2816, 1024, 1024, 3, "none", True, False, True, True, False, "gemm5_rb11"),      # masked rows (padded-batch semantics)
    (1404, 1024, 1024, 3, "none", True, True, True, False, False, None),               # M not a multiple of any tile height: partial row slab
    (1536, 768, 768, 3, "none", True, True, True, False, False, "gemm5_rb8"),         # F5-Small widths (C1): 128-row tiles
    (2816, 100, 1024, 3, "none", True, False, False, False, False, None),              # N = mel_dim: partial column panel
    (2816, 1024, 128, 3, "silu", True, False, False, False, False, None),              # K shorter than the ring depth
    (22528, 1024, 1024, 3, "none", True, True, True, False, False, "gemm6"),           # C3 share: 8 utterances x 2 branches (batch mode: 256 x 256 ping-pong tiles)
    (22528, 2048, 1024, 3, "gelu_tanh", True, False, False, False, True, "gemm6"),      # FF1 at the C3 share: fp16 plane out
    (22528, 1024, 2048, 3, "none", True, True, True, False, False, "gemm6"),           # FF2 at the C3 share (32 K-tiles)
    (22400, 1024, 1024, 3, "none", True, True, True, True, False, "gemm6"),            # ragged batch: the last 256-row tile has 128 valid rows; masked rows
    (16384, 1024, 64, 3, "silu", True, False, False, False, False, "gemm6"),           # one K-tile only (prologue without a second tile)
    (16384, 1024, 128, 3, "none", True, False, True, False, False, "gemm6"),           # two K-tiles
    (2816, 1024, 1024, 2, "none", True, True, True, False, False, None),               # bf16x3 (strict mode)
    (2816, 2048, 1024, 1, "gelu_tanh", True, False, False, False, False, None),        # plain bf16
    (200, 512, 1024, 2, "gelu_erf", True, False, False, False, False, None),           # Vocos-sized, erf GELU
]


@pytest.mark.parametrize("case", CASES, ids=[f"M{c[0]}_N{c[1]}_K{c[2]}_p{c[3]}_{c[4]}{'_keep' if c[8] else ''}{'_f16out' if c[9] else ''}" for c in CASES])
def test_gemm_unit_op(case):
    from tts_indic_server_f5_amd import ops
    M, N, K, prec, act, use_bias, use_mul, use_res, use_keep, out16, counter = case
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + prec)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1 if use_bias else None
    mul = torch.randn(N, generator=g) if use_mul else None
    res = torch.randn(M, N, generator=g) if use_res else None
    keep = (torch.rand(M, generator=g) > 0.3) if use_keep else None
    ref = _ref_matmul(a, w, prec)
    if bias is not None:
        ref = ref + bias.double()
    ref = _act(ref, act)
    if keep is not None:
        ref = ref * keep.double()[:, None]
    if mul is not None:
        ref = ref * mul.double()
    if res is not None:
        ref = ref + res.double()
    _reset_counters()
    out, _ = ops.gemm(a.to(DEV), w.to(DEV), bias, prec=prec, act=act, mul=mul, res=res, row_keep=keep, out16=out16)
    if counter:
        assert _counter(counter) == 1, f"{counter} path was not taken"
    assert torch.isfinite(out).all()
    if out16:
        # one fp16 rounding of the output on top of the accumulation error
        assert _rel(out.float(), ref) < 6e-4
        assert (out.float().cpu() - ref.half().float()).abs().max() <= 2 * torch.finfo(torch.float16).eps * ref.abs().max()
    else:
        assert _rel(out, ref) < (3e-4 if act in ("gelu_tanh", "silu", "mish") else 2e-5)   # hardware exp2 / rcp in the activations: ~1 ulp each


def test_gemm_f16_output_saturates():
    """fp16 safety: a pre-activation beyond the fp16 range must come out as +-65504, never inf (a trained checkpoint with FF1 / GELU
    outliers would otherwise poison the next GEMM's whole row with NaN)."""
    from tts_indic_server_f5_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(256, 1024, generator=g)
    w = torch.randn(1024, 1024, generator=g) / 32.0
    a[3] *= 40000.0       # rows 3 and 77 produce |y| >> 65504
    a[77] *= -90000.0
    out, _ = ops.gemm(a.to(DEV), w.to(DEV), None, prec=2, act="none", out16=True)
    assert torch.isfinite(out).all()
    ref = (a.double() @ w.double().T).clamp(-65504.0, 65504.0)
    assert out[3].float().abs().max().item() == 65504.0 and out[77].float().abs().max().item() == 65504.0
    ok = torch.ones(256, dtype=torch.bool); ok[3] = ok[77] = False
    assert _rel(out[ok].float(), ref[ok]) < 1e-3
    big = ref[~ok].abs() >= 65504.0
    assert (out[~ok].float().cpu().abs()[big] == 65504.0).all()


@pytest.mark.parametrize("M,D,prec", [(2816, 1024, 3), (1404, 1024, 3), (1536, 768, 3), (2816, 1024, 2), (300, 256, 2), (22528, 1024, 3), (8320, 768, 3)])
def test_qkv_unit_op(M, D, prec):
    """Fused QKV projection + epilogue against a reference that applies x-transformers' interleaved rotary embedding to channels
    0..63 of q and k (head 0 only: F/model/modules.py:414-426), scales q by log2(e) / 8 (the attention kernel works in base-2 exponents) and rounds to fp16 like the kernel's outputs."""
    from oracle import dit_oracle as O
    from tts_indic_server_f5_amd import ops
    g = torch.Generator().manual_seed(M + D + prec)
    a = torch.randn(M, D, generator=g)
    w = torch.randn(3 * D, D, generator=g) / D ** 0.5
    bias = torch.randn(3 * D, generator=g) * 0.1
    pos = torch.arange(M) % 1405          # two sequences' worth of positions
    y = (_ref_matmul(a, w, prec) + bias.double()).float()
    q, k, v = y[:, :D].clone(), y[:, D:2 * D].clone(), y[:, 2 * D:]
    freqs = O.rotary_freqs(1405, 64)[0][pos]           # [M, 64]
    q[:, :64] = O.apply_rotary(q[None, :, :64], freqs[None])[0]
    k[:, :64] = O.apply_rotary(k[None, :, :64], freqs[None])[0]
    q = q * Q_SCALE
    _reset_counters()
    gq, gk, gv, _ = ops.qkv(a.to(DEV), w.to(DEV), bias, pos.numpy(), prec=prec)
    if prec == 3 and M == 2816:
        assert _counter("gemm5_wide") == 1 and _counter("gemm5_rb11") == 1
    if M >= 8320:
        assert _counter("gemm6") == 1          # batch-mode shapes: the 256 x 256 ping-pong kernel (all-Q, all-K and all-V tiles; 8320 = 32.5 row tiles)
    for name, got, ref in (("q", gq, q), ("k", gk, k), ("v", gv, v)):
        err = (got.cpu() - ref.half().float()).abs()
        # fp16 outputs: identical up to accumulation-order flips of the last fp16 bit on some elements
        assert err.max() <= 2.0 ** -10 * ref.abs().max(), name
        assert (err > 0).float().mean() < 0.10, name
        assert _rel(got, ref.double()) < 4e-4, name


def test_qkv_and_attention_operands_saturate():
    """The attention operands are fp16 since round 3 (csrc/attn3.h): a q / k / v value beyond the fp16 range must leave the QKV epilogue as +-65504,
    never as inf (an inf score would make exp2 produce NaN through inf - inf in the next block), and the attention kernel must stay finite on
    operands at that limit (the fixed-offset fast loop overflows its fp16 probabilities there and the workgroup redoes its tile)."""
    from tts_indic_server_f5_amd import ops
    g = torch.Generator().manual_seed(77)
    M, D = 300, 256
    a = torch.randn(M, D, generator=g)
    w = torch.randn(3 * D, D, generator=g) / D ** 0.5
    a[7] *= 3.0e5                                   # row 7: |q|, |k|, |v| far beyond 65504
    gq, gk, gv, _ = ops.qkv(a.to(DEV), w.to(DEV), torch.zeros(3 * D), torch.arange(M).numpy(), prec=2)
    for t in (gq, gk, gv):
        assert torch.isfinite(t).all() and t[7].abs().max().item() == 65504.0
    n, heads = 200, 2
    q = torch.randn(n, 64 * heads, generator=g)
    k = torch.randn(n, 64 * heads, generator=g)
    v = torch.randn(n, 64 * heads, generator=g)
    k[150] = 60000.0                                # one key at the fp16 limit: scores of +-1e5 for every query
    out, _ = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), (n,), None, heads=heads)
    assert torch.isfinite(out).all()
    qs = (q * Q_SCALE).half().double().view(n, heads, 64).transpose(0, 1) * math.log(2.0)
    ks, vs = (t.half().double().view(n, heads, 64).transpose(0, 1) for t in (k, v))
    ref = (torch.softmax(qs @ ks.transpose(1, 2), dim=-1) @ vs).transpose(0, 1).reshape(n, 64 * heads)
    assert (out.double().cpu() - ref).abs().max().item() < 2.5e-3


@pytest.mark.parametrize("rms", [False, True])
def test_layernorm_unit_op(rms):
    from tts_indic_server_f5_amd import ops
    g = torch.Generator().manual_seed(9)
    M, D = 2816, 1024
    x = torch.randn(M, D, generator=g) * 3 + 0.5
    scale = torch.randn(D, generator=g) * 0.3
    shift = torch.randn(D, generator=g) * 0.3
    xd = x.double()
    if rms:
        ref = xd / xd.norm(dim=-1, keepdim=True).clamp_min(1e-12) * D ** 0.5 * scale.double()
        out = ops.layernorm(x.to(DEV), scale, torch.zeros(D), gain_off=0.0, eps=0.0, rms=True)
    else:
        mu, var = xd.mean(-1, keepdim=True), xd.var(-1, unbiased=False, keepdim=True)
        ref = (xd - mu) / (var + 1e-6).sqrt() * (1 + scale.double()) + shift.double()
        out = ops.layernorm(x.to(DEV), scale, shift, gain_off=1.0, eps=1e-6)
    assert _rel(out, ref) < 2e-6


@pytest.mark.parametrize("impl", [3])
@pytest.mark.parametrize("lens,kv,heads,k_gain", [((1404, 1404), None, 16, 1.0), ((300, 50, 257), (300, 41, 200), 4, 1.0), ((748,), None, 12, 1.0),
                                                  ((64,), (1,), 2, 1.0), ((2341, 2341), None, 16, 1.0), ((33,), (33,), 2, 1.0), ((97, 160), (40, 129), 2, 1.0),
                                                  ((1404, 300), (1404, 290), 4, 12.0), ((200,), None, 2, 40.0),
                                                  # 6-wave workgroups with the 9-stage ring (one barrier per two tiles) next to sequences of 2 / 3 / 3 / 5 tiles
                                                  ((1404, 70, 130, 200), (1404, 65, 130, 129), 8, 1.0), ((1404, 320, 130, 200), (1404, 300, 130, 129), 8, 12.0),
                                                  # the balanced 8-wave kernel with key counts inside the first half tile (its key-half waves 6 / 7 then own nothing
                                                  # valid), just past it, and one key; the last one also through the running-maximum redo
                                                  ((1404, 1404), (1404, 20), 16, 1.0), ((1404, 1404), (33, 1), 16, 1.0), ((1500, 1310), (47, 1310), 16, 12.0)])
def test_attention_unit_op(impl, lens, kv, heads, k_gain):
    """Attention kernel alone vs fp64 softmax attention on the fp16-rounded operands (q after the log2(e) / 8 scale, undone in fp64), incl. the key-padding mask
    (F/model/modules.py:429-434), ragged sequences, tiles that overhang a sequence, key counts that end inside either half of a 64-key tile,
    and the C2 / C1 / C5 shapes.  k_gain > 1 multiplies every key at a position = 7 mod 16 from position 40 on (never one of a query block's
    sample keys, csrc/attn3.h): logits tens to hundreds of nats above each query's maximum over its sample, which is what sends a workgroup
    of attn3 from its fixed-offset fast loop to the running-maximum redo."""
    from tts_indic_server_f5_amd import ops
    g = torch.Generator().manual_seed(sum(lens) + heads)
    n, D = sum(lens), 64 * heads
    q = torch.randn(n, D, generator=g) * 1.5
    k = torch.randn(n, D, generator=g) * 1.5
    v = torch.randn(n, D, generator=g)
    if k_gain != 1.0:
        o = 0
        for L in lens:
            k[o + 47:o + L:16] *= k_gain
            o += L
    out, _ = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), lens, kv, heads=heads, impl=impl)
    qb, kb, vb = (q * Q_SCALE).half().double() * math.log(2.0), k.half().double(), v.half().double()
    o, refs = 0, []
    for i, L in enumerate(lens):
        kl = L if kv is None else kv[i]
        qs, ks, vs = (t[o:o + L].view(L, heads, 64).transpose(0, 1) for t in (qb, kb, vb))
        s = qs @ ks.transpose(1, 2)
        s[:, :, kl:] = float("-inf")
        refs.append((torch.softmax(s, dim=-1) @ vs).transpose(0, 1).reshape(L, D))
        o += L
    ref = torch.cat(refs)
    # P is rounded to fp16 before the P V product (11 significand bits); the output planes carry 16 bits
    assert torch.isfinite(out).all()
    err, rel = (out.double().cpu() - ref).abs().max().item(), _rel(out, ref)
    print(f"[parity] attention lens {lens} kv {kv} heads {heads} k_gain {k_gain}: max err {err:.3e} rel rms {rel:.3e}")
    assert err < 2.5e-3
    assert rel < 5e-4


# ---------------------------------------------------------------------------------------------------------------- conv1d (BigVGAN convolutions)
def _conv_ref(x, w, bias, res, batch, P, T, dil):
    """nn.Conv1d on the valid rows of every sequence (zero padding at the sequence bounds), channel-last in / out."""
    M, ci = x.shape
    k = w.shape[-1]
    xv = x.view(batch, P, ci)[:, :T].transpose(1, 2).double()
    y = torch.nn.functional.conv1d(xv, w.double(), None if bias is None else bias.double(), dilation=dil, padding=dil * (k - 1) // 2)
    y = y.transpose(1, 2)
    if res is not None:
        y = y + res.view(batch, P, -1)[:, :T].double()
    return y


@pytest.mark.parametrize("impl", [5, 0])
@pytest.mark.parametrize("prec,tol", [(2, 4e-5), (3, 4e-3)])
@pytest.mark.parametrize("ci,co,k,dil,batch,P,T", [
    (768, 768, 11, 5, 2, 512, 470),     # stage-1 shape, the largest halo (25 rows), 12 channel chunks of 64 / 24 of 32, 6 column tiles
    (192, 192, 3, 1, 3, 256, 256),      # no padding rows at all: the window must stop at the sequence bounds, not read the neighbour
    (96, 96, 7, 3, 2, 512, 300),        # 96 channels: 64-byte rows in fp16 mode (not a multiple of 64), one column tile of 128 with 96 real
    (24, 24, 11, 1, 2, 1024, 1000),     # last stage: one chunk (no window double buffer), 24 of 64 columns real
    (48, 48, 3, 5, 1, 512, 512),        # one chunk of 64 with 48 real channels
    (384, 768, 3, 1, 2, 256, 200),      # c_out != c_in (the 3-tap form of an up-sampler)
])
def test_conv1d_vs_torch(impl, prec, tol, ci, co, k, dil, batch, P, T):
    """f5hip_op_conv1d through both kernels (conv5.h sliding window, gemm.h implicit GEMM) vs torch conv1d in float64.
    Tolerance (absolute, outputs of rms ~1.7): split bf16 carries ~16 mantissa bits per operand -- 2.4e-5 measured at K = 8448,
    i.e. 1.4e-5 relative; fp16 (11 bits) is the documented fast mode."""
    from tts_indic_server_f5_amd import ops
    g = torch.Generator().manual_seed(1000 + ci + k + dil)
    x = torch.randn(batch * P, ci, generator=g)
    w = torch.randn(co, ci, k, generator=g) / (ci * k) ** 0.5
    bias = torch.randn(co, generator=g)
    res = torch.randn(batch * P, co, generator=g)
    out, _, _ = ops.conv1d(x.to(DEV), w, bias, res.to(DEV), batch=batch, valid=T, dilation=dil, prec=prec, impl=impl)
    ref = _conv_ref(x, w, bias, res, batch, P, T, dil)
    got = out.cpu().view(batch, P, co)[:, :T].double()
    err = (got - ref).abs().max().item()
    print(f"[parity] conv1d impl {impl} prec {prec} ci {ci} co {co} k {k} dil {dil}: max err {err:.3e} (ref rms {ref.pow(2).mean().sqrt():.3f})")
    assert err < tol


# ---------------------------------------------------------------------------------------------------------------- MMDiT joint attention
@pytest.mark.parametrize("x_len,c_len,x_kv", [([300], [21], None), ([1404, 257, 64], [240, 65, 7], [1404, 200, 33]), ([130, 129], [128, 1], [100, 129])])
def test_joint_attention_vs_torch(x_len, c_len, x_kv):
    """attn3's two-range kernels (keys = audio rows then text rows of the same sequence; queries from either) vs fp64 torch SDPA over
    the concatenation, padding masked on the audio keys only (F/model/modules.py:506-514).  Same operand rounding and tolerance as the
    single-range attention test."""
    from tts_indic_server_f5_amd import ops
    heads, D = 4, 256
    g = torch.Generator().manual_seed(500 + sum(x_len))
    Fx, Fc = sum(x_len), sum(c_len)
    q, k, v = (torch.randn(Fx + Fc, D, generator=g) for _ in range(3))
    out = ops.joint_attention(q.to(DEV), k.to(DEV), v.to(DEV), x_len, c_len, x_kv, heads=heads).cpu()
    bf = lambda t: t.to(torch.float16).double()
    ox, oc = 0, Fx
    worst = 0.0
    for i, (n, nt) in enumerate(zip(x_len, c_len)):
        sel = torch.cat([torch.arange(ox, ox + n), torch.arange(oc, oc + nt)])
        qq = (bf(q[sel] * Q_SCALE) * math.log(2.0)).view(n + nt, heads, 64).transpose(0, 1)
        kk = bf(k[sel]).view(n + nt, heads, 64).transpose(0, 1)
        vv = bf(v[sel]).view(n + nt, heads, 64).transpose(0, 1)
        s = qq @ kk.transpose(1, 2)
        kv = n if x_kv is None else x_kv[i]
        key_ok = torch.cat([torch.arange(n) < kv, torch.ones(nt, dtype=torch.bool)])
        s = s.masked_fill(~key_ok[None, None, :], float("-inf"))
        ref = (torch.softmax(s, dim=-1) @ vv).transpose(0, 1).reshape(n + nt, D)
        err = (out[sel].double() - ref).abs().max().item()
        worst = max(worst, err)
        ox += n; oc += nt
    print(f"[parity] joint attention x {x_len} c {c_len} kv {x_kv}: max err {worst:.3e}")
    assert worst < 2.5e-3   # fp16 P and V operands (11 significand bits) over up to ~1.6 k keys; the single-range test uses the same bound


def test_attention_random_shapes(monkeypatch):
    """tools/attn_fuzz.py with a fixed seed: 24 random ragged batches (1-16 heads, key counts ending anywhere in a tile, a third of them with
    logits far above the first key block) + 6 two-range cases, each against fp64 attention on the rounded operands, bound 2.5e-3 as above."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "attn_fuzz.py")
    spec = importlib.util.spec_from_file_location("attn_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["attn_fuzz.py", "24", "3"])
    mod.main()   # exits non-zero on the first case out of bounds
