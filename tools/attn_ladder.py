#!/usr/bin/env python3
"""Attention-operand precision ladder: which of Q.K^T and P.V must leave bf16 for the reference's tiny
UNetT / MMDiT CFM.sample fixtures to pass the UNSCALED 1e-3, and what it buys on F5-Base at 32 NFE.  CPU only.

Emulates inside the fp32 oracle (oracle/dit_oracle.py, pinned by the reference fixtures) what the library rounds:
  GEMMs      "x3"    = split bf16 everywhere (gemm_planes 2);  "mixed" = fp16 x fp16 for the block GEMMs (to_q/k/v/out, ff), split bf16 elsewhere
  attention  qk / pv in {bf16, fp16}: q (pre-scaled by 1/8) and k rounded to qk; exp(s - max) and v rounded to pv; fp32 scores, sums, accumulation

usage: python tools/attn_ladder.py tiny            # the two tiny fixtures (seconds)
       python tools/attn_ladder.py base [steps]    # F5-Base N = 1404 (minutes per row)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as TF

from oracle import dit_oracle as O
from tts_indic_server_f5_amd import synth


def rnd(x, kind):
    return x if kind == "fp32" else x.to(torch.bfloat16 if kind == "bf16" else torch.float16).float()


class Emu:
    def __init__(self, gemm, qk, pv, names):
        self.gemm, self.qk, self.pv, self.names = gemm, qk, pv, names

    def __getattr__(self, k):
        return getattr(TF, k)

    def linear(self, x, w, b=None):
        if self.gemm == "fp32" or x.dim() != 3:
            return TF.linear(x, w, b)
        nm = self.names.get(w.data_ptr(), "")
        block = any(s in nm for s in (".to_q", ".to_k", ".to_v", ".to_out", ".ff.", "ff_x.", "ff_c."))
        if self.gemm == "mixed" and block:
            return TF.linear(rnd(x, "fp16"), rnd(w, "fp16"), b)
        xh, wh = rnd(x, "bf16"), rnd(w, "bf16")
        return TF.linear(xh, wh, b) + TF.linear(xh, rnd(w - wh, "bf16")) + TF.linear(rnd(x - xh, "bf16"), wh)

    def scaled_dot_product_attention(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        if self.qk == "fp32" and self.pv == "fp32":
            return TF.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
        s = rnd(q * 0.125, self.qk) @ rnd(k, self.qk).transpose(-1, -2)
        if attn_mask is not None:
            s = s.masked_fill(~attn_mask, float("-inf"))
        p = torch.exp(s - s.amax(-1, keepdim=True))
        return (rnd(p, self.pv) @ rnd(v, self.pv)) / p.sum(-1, keepdim=True)


ROWS = [("x3", "bf16", "bf16"), ("x3", "fp16", "bf16"), ("x3", "bf16", "fp16"), ("x3", "fp16", "fp16"), ("x3", "fp32", "fp32"),
        ("mixed", "bf16", "bf16"), ("mixed", "fp16", "fp16"), ("mixed", "fp32", "fp32"), ("fp32", "bf16", "bf16"), ("fp32", "fp16", "fp16")]


def ladder(label, sd, sample, refs):
    names = {v.data_ptr(): k for k, v in sd.items()}
    print(f"== {label}")
    for gemm, qk, pv in ROWS:
        O.F = Emu(gemm, qk, pv, names)
        t0 = time.time()
        got = sample()
        O.F = TF
        msg = f"  gemm {gemm:5s} qk {qk:4s} pv {pv:4s}"
        for tag, ref in refs.items():
            d = got - ref
            msg += f"  vs {tag}: rms {d.pow(2).mean().sqrt():.3e}"
        print(msg + f"   ({time.time() - t0:.1f} s)", flush=True)


def tiny():
    gd = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    load = lambda n: {k: torch.from_numpy(v) for k, v in np.load(os.path.join(gd, n + ".npz")).items()}
    ut = dict(dim=128, depth=4, heads=2, ff_mult=4, text_num_embeds=40)
    g, sd, cfg = load("unett_tiny"), synth.unett_state_dict(**ut), O.UNetTConfig(**ut)

    def s_unett():
        torch.manual_seed(0)
        out, _ = O.cfm_sample(sd, cfg, g["cond"][:, :15], g["text"], 45, steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9,
                              forward_fn=lambda **kw: O.unett_forward(sd, cfg, **kw), keep_trajectory=False)
        return out[:, 15:]
    ladder("tiny UNetT CFM.sample, 8 NFE (reference fixture unett_tiny.npz; tests/test_gpu_dit.py:147)", sd, s_unett,
           {"reference": g["sample_out"][:, 15:], "oracle fp32": s_unett()})
    mt = dict(dim=128, depth=3, heads=2, ff_mult=2, text_num_embeds=40)
    g2, sd2, cfg2 = load("mmdit_tiny"), synth.mmdit_state_dict(**mt), O.MMDiTConfig(**mt)

    def s_mmdit():
        out, _ = O.cfm_sample(sd2, cfg2, g2["sample_cond"], g2["sample_text"], 48, steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7,
                              forward_fn=lambda **kw: O.mmdit_forward(sd2, cfg2, **kw), keep_trajectory=False)
        return out[:, 20:]
    ladder("tiny MMDiT CFM.sample, 8 NFE (reference fixture mmdit_tiny.npz; tests/test_gpu_dit.py:185)", sd2, s_mmdit,
           {"reference": g2["sample_out"][:, 20:], "oracle fp32": s_mmdit()})


def base(steps):
    sd, cfg = synth.dit_state_dict(), O.F5_BASE
    gc = torch.Generator().manual_seed(14)
    cond = torch.randn(1, 469, 100, generator=gc)
    text, y0 = synth.text_ids(), synth.noise(1404, 0)[None]

    def s():
        out, _ = O.cfm_sample(sd, cfg, cond, text, 1404, steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, keep_trajectory=False)
        return out[:, 469:]
    global ROWS
    ROWS = [("mixed", "bf16", "bf16"), ("mixed", "fp16", "bf16"), ("mixed", "bf16", "fp16"), ("mixed", "fp16", "fp16"), ("x3", "fp16", "fp16")]
    ladder(f"F5-Base CFM.sample, N = 1404, {steps} NFE, CFG 2, sway -1", sd, s, {"oracle fp32": s()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "base":
        base(int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    else:
        tiny()
