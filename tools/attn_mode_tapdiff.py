#!/usr/bin/env python3
"""Per-layer account of "the same utterance, another attention kernel variant" (VERDICT r2 item 1c).  GPU.

One F5-Base forward over TWO copies of the C2 utterance (N = 1404 each: the 2 x 16 x 8 = 256-workgroup launch of a CFG pair, which is what
selects the balanced kernel), hidden state behind n blocks, three runs per GEMM mode:
  A  default attention (balanced 8-wave kernel: the key halves of a third of the query blocks are summed separately)
  B  f5hip_set_attention_shape_invariant(1) (6-wave kernel, one association for every query block)
  C  like B, but the INPUT x perturbed by 1 ulp-sized relative noise (2^-24): no attention difference at all
and prints rms(A - B) and rms(C - B) behind 0, 1, 2, 4, 8, 16, 22 blocks.  If the two columns grow alike, the divergence of two attention
variants is the mode's own sensitivity to ANY last-bit change (fp16 operand roundings flipping), not attention arithmetic.
usage: python tools/attn_mode_tapdiff.py > profiles/r03_attn_mode_tapdiff.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import _lib, synth  # noqa: E402
from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel  # noqa: E402


def rms(a, b):
    return (a.float() - b.float()).pow(2).mean().sqrt().item()


def main():
    sd = synth.dit_state_dict()
    g = torch.Generator().manual_seed(14)
    cond = (torch.randn(1, 1404, 100, generator=g) * (torch.arange(1404)[None, :, None] < 469)).expand(2, -1, -1).contiguous()
    text = synth.text_ids().expand(2, -1).contiguous()
    x = synth.noise(1404, 0)[None].expand(2, -1, -1).contiguous()
    x_eps = x * (1.0 + 2.0 ** -24 * torch.randn(x.shape, generator=g))
    inv = lambda on: _lib.check(_lib.lib().f5hip_set_attention_shape_invariant(int(on)), "set_attention_shape_invariant")
    for planes, name in ((2, "bf16x3 (split bf16 everywhere)"), (3, "mixed (fp16 block GEMMs)")):
        m = F5HipModel(F5TTS_BASE, sd, gemm_planes=planes)
        print(f"== gemm mode {planes}: {name}")
        print("   blocks   rms(default attn - invariant attn)   rms(x (1 + 2^-24 noise) - x), both invariant      hidden rms")
        for nb in (0, 1, 2, 4, 8, 16, 22):
            inv(0)
            a = m.transformer_forward(x, cond, text, 0.25, False, False, n_blocks=nb)
            inv(1)
            b = m.transformer_forward(x, cond, text, 0.25, False, False, n_blocks=nb)
            c = m.transformer_forward(x_eps, cond, text, 0.25, False, False, n_blocks=nb)
            print(f"   {nb:4d}     {rms(a, b):.3e}                             {rms(c, b):.3e}                                   {b.float().pow(2).mean().sqrt().item():.3f}", flush=True)
        inv(0)
        a = m.transformer_forward(x, cond, text, 0.25, False, False)
        inv(1)
        b = m.transformer_forward(x, cond, text, 0.25, False, False)
        c = m.transformer_forward(x_eps, cond, text, 0.25, False, False)
        print(f"   output   {rms(a, b):.3e}                             {rms(c, b):.3e}")
        inv(0)
        del m


if __name__ == "__main__":
    main()
