#!/usr/bin/env python3
"""attn5 (ping-pong attention kernel) against fp64 attention on the fp16-rounded operands at the shapes of tests/test_gpu_ops.py::test_attention_unit_op,
and timed next to attn3 at the config shapes (f5hip_op_attention impl 5 / 3, HIP events, 100 launches).  Needs a library built with
`python tts-indic-server-f5_amd/build.py --experiments` (without it impl 5 runs attn3)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

QS = 0.125 * math.log2(math.e)
CASES = [((1404, 1404), None, 16, 1.0), ((300, 50, 257), (300, 41, 200), 4, 1.0), ((748,), None, 12, 1.0), ((64,), (1,), 2, 1.0), ((2341, 2341), None, 16, 1.0),
         ((33,), (33,), 2, 1.0), ((97, 160), (40, 129), 2, 1.0), ((1404, 300), (1404, 290), 4, 12.0), ((200,), None, 2, 40.0),
         ((1404, 70, 130, 200), (1404, 65, 130, 129), 8, 1.0), ((1404, 320, 130, 200), (1404, 300, 130, 129), 8, 12.0),
         ((1404, 1404), (1404, 20), 16, 1.0), ((1404, 1404), (33, 1), 16, 1.0), ((1500, 1310), (47, 1310), 16, 12.0), ((1404,) * 16, None, 16, 1.0)]
bad = 0
for lens, kv, heads, gain in CASES:
    g = torch.Generator().manual_seed(sum(lens) + heads)
    n, D = sum(lens), 64 * heads
    q = torch.randn(n, D, generator=g) * 1.5
    k = torch.randn(n, D, generator=g) * 1.5
    v = torch.randn(n, D, generator=g)
    if gain != 1.0:
        o = 0
        for L in lens:
            k[o + 47:o + L:16] *= gain
            o += L
    out, _ = ops.attention(q.cuda(), k.cuda(), v.cuda(), lens, kv, heads=heads, impl=5)
    out3, _ = ops.attention(q.cuda(), k.cuda(), v.cuda(), lens, kv, heads=heads, impl=3)
    qb, kb, vb = (q * QS).half().double() * math.log(2.0), k.half().double(), v.half().double()
    o, refs = 0, []
    for i, L in enumerate(lens):
        kl = L if kv is None else kv[i]
        qs, ks, vs = (t[o:o + L].view(L, heads, 64).transpose(0, 1) for t in (qb, kb, vb))
        s = qs @ ks.transpose(1, 2)
        s[:, :, kl:] = float("-inf")
        refs.append((torch.softmax(s, dim=-1) @ vs).transpose(0, 1).reshape(L, D))
        o += L
    ref = torch.cat(refs)
    err = (out.double().cpu() - ref).abs().max().item()
    err3 = (out3.double().cpu() - ref).abs().max().item()
    ok = torch.isfinite(out).all().item() and err < 2.5e-3
    bad += not ok
    print(f"lens {str(lens)[:40]:40s} kv {str(kv)[:24]:24s} heads {heads:2d} gain {gain:4.1f}: attn5 max err {err:.3e} (attn3 {err3:.3e}) {'ok' if ok else '<-- FAIL'}", flush=True)
print("FAILURES:", bad)
for tag, lens, heads in (("C2  2 x 1404, 16 heads", (1404, 1404), 16), ("C3 share 16 x 1404", (1404,) * 16, 16), ("C4 32 x 1404", (1404,) * 32, 16), ("C5  2 x 2341", (2341, 2341), 16),
                         ("C5 16 x 2341", (2341,) * 16, 16), ("C1  2 x 748, 12 heads", (748, 748), 12)):
    n, D = sum(lens), 64 * heads
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(n, D, generator=g).cuda() for _ in range(3))
    fl = sum(4.0 * L * L * 64 * heads for L in lens)
    line = f"{tag:26s}"
    for impl in (3, 5, 3, 5):
        _, us = ops.attention(q, k, v, lens, heads=heads, impl=impl, iters=100)
        line += f"  attn{impl}: {us:8.2f} us ({fl / us / 1e6 / 2500:.3f})"
    print(line, flush=True)
