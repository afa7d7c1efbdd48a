#!/usr/bin/env python3
"""Precision-ladder experiment: emulate bf16 MFMA operand rounding inside the fp32 oracle
and measure the mel RMS error of the full CFG Euler loop against plain fp32.  CPU only; decides which GEMMs can
run plain bf16 and which need split-bf16 (hi+lo, 3 MFMAs).

usage: python tools/precision_ladder.py small|base [steps]
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF
from oracle import dit_oracle as O
from tts_indic_server_f5_amd import synth


def bf(x):
    return x.to(torch.bfloat16).float()


class Emu:
    """F-proxy: linear() with emulated operand precision, per weight-shape policy."""
    def __init__(self, policy, attn_bf16):
        self.policy, self.attn_bf16 = policy, attn_bf16
        self.wcache = {}

    def __getattr__(self, k):
        return getattr(TF, k)

    def _w(self, w, mode):
        key = (w.data_ptr(), mode)
        if key not in self.wcache:
            hi = bf(w)
            self.wcache[key] = (hi, bf(w - hi))
        return self.wcache[key]

    def linear(self, x, w, b=None):
        mode = self.policy(w.shape, x.shape)
        if callable(mode):
            mode = mode(self.names.get(w.data_ptr(), ""))
        if mode == "fp32":
            return TF.linear(x, w, b)
        wh, wl = self._w(w, mode)
        xh = bf(x)
        if mode == "bf16":
            return TF.linear(xh, wh, b)
        xl = bf(x - xh)
        if mode == "bf16x3":
            return TF.linear(xh, wh, b) + TF.linear(xh, wl) + TF.linear(xl, wh)
        if mode == "bf16x2w":   # activations split, weights bf16 only
            return TF.linear(xh, wh, b) + TF.linear(xl, wh)
        raise ValueError(mode)

    def scaled_dot_product_attention(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        if not self.attn_bf16:
            return TF.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
        q, k, v = bf(q * 0.125), bf(k), bf(v)
        s = q @ k.transpose(-1, -2)
        if attn_mask is not None:
            s = s.masked_fill(~attn_mask, float("-inf"))
        m = s.amax(-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(-1, keepdim=True)
        return (bf(p) @ v) / l


def run(model, steps, policy, attn_bf16, label, ref=None):
    arch = dict(dim=768, depth=18, heads=12) if model == "small" else {}
    cfg = O.F5_SMALL if model == "small" else O.F5_BASE
    n_ref, n = (468, 748) if model == "small" else (468, 1404)
    sd = run.sd.setdefault(model, synth.dit_state_dict(**arch))
    g = torch.Generator().manual_seed(14)
    cond = torch.randn(1, 469, 100, generator=g)
    text = synth.text_ids(60, 36 if model == "small" else 120)
    y0 = synth.noise(n, 0)[None]
    O.F = Emu(policy, attn_bf16) if policy else TF
    if policy:
        O.F.names = {v.data_ptr(): k for k, v in sd.items()}
    t0 = time.time()
    out, _ = O.cfm_sample(sd, cfg, cond, text, n, steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0,
                          keep_trajectory=False)
    O.F = TF
    gen = out[:, n_ref:]
    msg = f"{label:34s} {time.time()-t0:6.1f}s  gen std {gen.std():.3f}"
    if ref is not None:
        d = gen - ref
        msg += f"  RMS err {d.pow(2).mean().sqrt():.3e}  max {d.abs().max():.3e}"
    print(msg, flush=True)
    return gen
run.sd = {}


if __name__ == "__main__":
    model = sys.argv[1] if len(sys.argv) > 1 else "small"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if model == "small" else 32)
    torch.set_num_threads(8)
    big = lambda ws: ws[0] >= 512 and ws[1] >= 512 and ws[0] % 64 == 0   # block GEMMs (q,k,v,out,ff) + text pwconv
    ref = run(model, steps, None, False, "fp32")
    allbf = lambda ws, xs: "bf16" if len(xs) == 3 else "fp32"
    x3 = lambda ws, xs: "bf16x3" if len(xs) == 3 else "fp32"
    if "--base-decide" in sys.argv:
        run(model, steps, allbf, True, "all token GEMMs bf16 + attn bf16", ref)
        run(model, steps, x3, True, "all bf16x3 + attn bf16", ref)
    else:
        run(model, steps, allbf, False, "all token GEMMs bf16", ref)
        run(model, steps, allbf, True, "all token GEMMs bf16 + attn bf16", ref)
        blk = lambda ws, xs: ("bf16" if big(ws) else "bf16x3") if len(xs) == 3 else "fp32"
        run(model, steps, blk, False, "block GEMMs bf16, in/out x3", ref)
        run(model, steps, x3, False, "all token GEMMs bf16x3", ref)
        run(model, steps, x3, True, "all bf16x3 + attn bf16", ref)
        x2 = lambda ws, xs: "bf16x2w" if len(xs) == 3 else "fp32"
        run(model, steps, x2, False, "act split, weights bf16", ref)
    if "--base-decide" in sys.argv:
        qkvbf = lambda ws, xs: (lambda nm: "bf16" if (".to_q." in nm or ".to_k." in nm or ".to_v." in nm) else "bf16x3") if len(xs) == 3 else "fp32"
        run(model, steps, qkvbf, True, "x3, QKV bf16, attn bf16", ref)
        sys.exit(0)
    # fp16 variants (same MFMA rate as bf16, 3 more mantissa bits)
    def h(x): return x.to(torch.float16).float()
    class EmuH(Emu):
        def linear(self, x, w, b=None):
            mode = self.policy(w.shape, x.shape)
            if mode == "fp32": return TF.linear(x, w, b)
            wh = h(w); xh = h(x)
            if mode == "fp16": return TF.linear(xh, wh, b)
            if mode == "fp16x2a": return TF.linear(xh, wh, b) + TF.linear(h(x - xh), wh)
            if mode == "fp16x3": return TF.linear(xh, wh, b) + TF.linear(h(x - xh), wh) + TF.linear(xh, h(w - wh))
    globals()["Emu"] = EmuH
    for md in ("fp16", "fp16x2a", "fp16x3"):
        run(model, steps, (lambda ws, xs, md=md: md if len(xs) == 3 else "fp32"), True, f"all {md} + attn bf16", ref)
