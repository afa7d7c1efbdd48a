#!/usr/bin/env python3
"""Randomised shapes for the attention kernel against fp64 softmax attention on the fp16-rounded operands (same reference and bounds as
tests/test_gpu_ops.py::test_attention_unit_op): ragged batches, key counts ending anywhere in a tile, 1-16 heads, occasional large logits
(the running-maximum redo), and the two-range (joint) kernels.  Usage: python tools/attn_fuzz.py [n_cases] [seed]"""
import math
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

QS = 0.125 * math.log2(math.e)
LN2 = math.log(2.0)


def ref_attn(q, k, v, heads, key_ok):
    L = q.shape[0]
    qs, ks, vs = (t.view(t.shape[0], heads, 64).transpose(0, 1) for t in (q, k, v))
    s = qs @ ks.transpose(1, 2)
    s = s.masked_fill(~key_ok[None, None, :], float("-inf"))
    return (torch.softmax(s, dim=-1) @ vs).transpose(0, 1).reshape(L, heads * 64)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    worst = 0.0
    for case in range(n_cases):
        heads = rng.choice([1, 2, 3, 4, 8, 12, 16])
        n_seq = rng.choice([1, 1, 2, 2, 3, 5, 9])
        big = rng.random() < 0.35
        lens = [rng.randint(1, 2900 if big else 400) for _ in range(n_seq)]
        if big and rng.random() < 0.5:
            lens[0] = rng.choice([1404, 1405, 2340, 2341, 1023, 1024, 1025])
        kv = [rng.randint(1, L) if rng.random() < 0.5 else L for L in lens]
        gain = rng.choice([1.0, 1.0, 1.0, 8.0, 30.0])
        g = torch.Generator().manual_seed(case)
        n, D = sum(lens), 64 * heads
        q = torch.randn(n, D, generator=g) * 1.5
        k = torch.randn(n, D, generator=g) * 1.5
        v = torch.randn(n, D, generator=g)
        if gain != 1.0:
            o = 0
            for L in lens:
                k[o + min(40, L - 1):o + L] *= gain
                o += L
        out, _ = ops.attention(q.cuda(), k.cuda(), v.cuda(), lens, kv, heads=heads, impl=3)
        out = out.double().cpu()
        qb, kb, vb = (q * QS).half().double() * LN2, k.half().double(), v.half().double()
        o, err = 0, 0.0
        for L, kl in zip(lens, kv):
            ok = torch.arange(L) < kl
            r = ref_attn(qb[o:o + L], kb[o:o + L], vb[o:o + L], heads, ok)
            err = max(err, (out[o:o + L] - r).abs().max().item())
            o += L
        worst = max(worst, err)
        flag = "" if (err < 2.5e-3 and torch.isfinite(out).all()) else "   <-- FAIL"
        print(f"case {case:3d}: heads {heads:2d} lens {lens} kv {kv} gain {gain:4.1f}: max err {err:.3e}{flag}", flush=True)
        if flag:
            sys.exit(1)
    # two-range (joint) attention
    for case in range(max(4, n_cases // 4)):
        heads = rng.choice([2, 4, 8])
        n_seq = rng.choice([1, 2, 3])
        x_len = [rng.randint(1, 700) for _ in range(n_seq)]
        c_len = [rng.randint(1, 200) for _ in range(n_seq)]
        x_kv = [rng.randint(1, L) if rng.random() < 0.5 else L for L in x_len]
        g = torch.Generator().manual_seed(1000 + case)
        Fx, Fc, D = sum(x_len), sum(c_len), 64 * heads
        q, k, v = (torch.randn(Fx + Fc, D, generator=g) for _ in range(3))
        out = ops.joint_attention(q.cuda(), k.cuda(), v.cuda(), x_len, c_len, x_kv, heads=heads).double().cpu()
        bf = lambda t: t.to(torch.float16).double()
        ox, oc, err = 0, Fx, 0.0
        for n, nt, kvn in zip(x_len, c_len, x_kv):
            sel = torch.cat([torch.arange(ox, ox + n), torch.arange(oc, oc + nt)])
            ok = torch.cat([torch.arange(n) < kvn, torch.ones(nt, dtype=torch.bool)])
            r = ref_attn(bf(q[sel] * QS) * LN2, bf(k[sel]), bf(v[sel]), heads, ok)
            err = max(err, (out[sel] - r).abs().max().item())
            ox += n; oc += nt
        worst = max(worst, err)
        flag = "" if (err < 2.5e-3 and torch.isfinite(out).all()) else "   <-- FAIL"
        print(f"joint {case:3d}: heads {heads} x {x_len} c {c_len} kv {x_kv}: max err {err:.3e}{flag}", flush=True)
        if flag:
            sys.exit(1)
    print(f"all cases within 2.5e-3 (worst {worst:.3e})")


if __name__ == "__main__":
    main()
