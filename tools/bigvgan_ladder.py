#!/usr/bin/env python3
"""Per-stage precision ladder of the BigVGAN convolutions: which stages tolerate ONE fp16 plane per operand (one MFMA per
product instead of split bf16's three) while the waveform stays inside north_star's 1e-4?  CPU only: emulates the operand rounding inside
the fp32 oracle (oracle/bigvgan_oracle.py; parity unpinned leaf) on the full 112 M-parameter geometry.

  stage -1 = conv_pre, stage i = ups[i] + its three AMP blocks (36 convolutions + the up-sampler at 1536 >> (i + 1) channels), stage 6 = conv_post
  "fp16": activations and weights rounded to fp16, exact products, fp32 accumulation;  "x3": split bf16 (hi hi + hi lo + lo hi)
usage: python tools/bigvgan_ladder.py [frames=96]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF

from oracle import bigvgan_oracle as B
from tts_indic_server_f5_amd import synth


class Emu:
    def __init__(self):
        self.stage, self.fp16_stages, self.cache = -1, set(), {}

    def __getattr__(self, k):
        return getattr(TF, k)

    def _round(self, x, w):
        if self.stage in self.fp16_stages:
            return [(x.half().float(), w.half().float())]
        xh, wh = x.bfloat16().float(), w.bfloat16().float()
        return [(xh, wh), (xh, (w - wh).bfloat16().float()), ((x - xh).bfloat16().float(), wh)]

    def conv1d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1:   # the anti-aliasing FIRs run on the vector units in fp32
            return TF.conv1d(x, w, b, stride=stride, padding=padding, dilation=dilation, groups=groups)
        parts = self._round(x, w)
        y = TF.conv1d(parts[0][0], parts[0][1], b, padding=padding, dilation=dilation)
        for xp, wp in parts[1:]:
            y = y + TF.conv1d(xp, wp, None, padding=padding, dilation=dilation)
        return y

    def conv_transpose1d(self, x, w, b=None, stride=1, padding=0, groups=1):
        if groups != 1:
            return TF.conv_transpose1d(x, w, b, stride=stride, padding=padding, groups=groups)
        parts = self._round(x, w)
        y = TF.conv_transpose1d(parts[0][0], parts[0][1], b, stride=stride, padding=padding)
        for xp, wp in parts[1:]:
            y = y + TF.conv_transpose1d(xp, wp, None, stride=stride, padding=padding)
        return y


@torch.no_grad()
def forward(sd, cfg, mel, emu):
    """bigvgan_forward with the stage index announced to the emulator (same operations, same order)."""
    F = emu if emu is not None else TF
    B.F = F
    if emu:
        emu.stage = -1
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg.resblock_kernel_sizes)
    for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        if emu:
            emu.stage = i
        x = F.conv_transpose1d(x, sd[f"ups.{i}.0.weight"], sd[f"ups.{i}.0.bias"], stride=r, padding=(k - r) // 2)
        xs = None
        for j in range(nk):
            y = B.amp_block1(sd, f"resblocks.{i * nk + j}.", x, cfg.resblock_kernel_sizes[j], cfg.resblock_dilation_sizes[j])
            xs = y if xs is None else xs + y
        x = xs / nk
    if emu:
        emu.stage = 6
    x = B.activation1d(x, sd["activation_post.act.alpha"], sd["activation_post.act.beta"])
    x = F.conv1d(x, sd["conv_post.weight"], None, padding=3)
    B.F = TF
    return torch.clamp(x, min=-1.0, max=1.0)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    torch.set_num_threads(8)
    sd, cfg = synth.bigvgan_state_dict(), B.BIGVGAN_V2_24K_100B_256X
    g = torch.Generator().manual_seed(78)
    mel = torch.randn(1, 100, T, generator=g) * 1.5 - 1.0
    t0 = time.time()
    ref = forward(sd, cfg, mel, None)
    print(f"BigVGAN v2 24 kHz 100-band 256x, T = {T} frames ({ref.shape[-1]} samples), fp32 reference {time.time() - t0:.1f} s; waveform rms {ref.pow(2).mean().sqrt():.3f}, "
          f"clipped {(ref.abs() >= 1).float().mean():.4f}; bound: 1e-4 max")
    # share of the convolution FLOPs per stage (per output sample)
    ch = [cfg.upsample_initial_channel >> (i + 1) for i in range(6)]
    up = [4, 16, 32, 64, 128, 256]
    fl = [u * c * c * (3 + 7 + 11) * 6 for u, c in zip(up, ch)]
    print("conv FLOP share per stage:", ", ".join(f"{i}: {100 * f / sum(fl):.1f} %" for i, f in enumerate(fl)))
    rows = [("all split bf16 (parity mode)", set()), ("all fp16 (fast mode)", {-1, 0, 1, 2, 3, 4, 5, 6})]
    rows += [(f"fp16 in stage {i} only ({ch[i]} channels)", {i}) for i in range(6)]
    rows += [("fp16 in conv_pre only", {-1}), ("fp16 in conv_post only", {6})]
    rows += [(f"fp16 in stages 0..{k}", set(range(k + 1))) for k in (1, 2, 3)]
    rows += [(f"fp16 in stages {k}..5", set(range(k, 6))) for k in (2, 3, 4)]
    for label, st in rows:
        emu = Emu()
        emu.fp16_stages = st
        t0 = time.time()
        got = forward(sd, cfg, mel, emu)
        d = (got - ref).abs()
        share = sum(fl[i] for i in st if 0 <= i < 6) / sum(fl)
        print(f"  {label:38s} max err {d.max():.3e}  rms {d.pow(2).mean().sqrt():.3e}  {'inside' if d.max() < 1e-4 else 'OUTSIDE'} 1e-4   (fp16 share of conv FLOPs {100 * share:5.1f} %)  {time.time() - t0:5.1f} s", flush=True)


if __name__ == "__main__":
    main()
