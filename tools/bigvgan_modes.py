#!/usr/bin/env python3
"""BigVGAN operand-precision modes vs the CPU oracle and their timing: python tools/bigvgan_modes.py PLANES [--oracle936] [--bench]
PLANES = 2 (split bf16) or 3 (fp16).  F5HIP_BV_SNAKE=1 in the environment selects the round-1 activation kernel (A/B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
from oracle import bigvgan_oracle as B   # checker only

planes = int(sys.argv[1])
tag = f"planes={planes} snake={'old' if os.environ.get('F5HIP_BV_SNAKE') == '1' else 'new'}"


def report(name, got, ref):
    d = (got.float().cpu() - ref.float()).abs()
    print(f"[{tag}] {name}: max err {d.max():.3e} rms {d.pow(2).mean().sqrt():.3e} ref rms {ref.pow(2).mean().sqrt():.3e} clipped {(ref.abs() >= 1).float().mean():.4f}", flush=True)


sd_s = synth.bigvgan_state_dict(upsample_initial_channel=256)
voc_s = F5HipBigVGAN(sd_s, upsample_initial_channel=256, gemm_planes=planes)
for b, t in ((1, 40), (2, 13), (1, 130)):
    mel = torch.randn(b, 100, t, generator=torch.Generator().manual_seed(200 + t)) * 1.5 - 1.0
    report(f"c0=256 b{b} t{t}", voc_s(mel), B.bigvgan_forward(sd_s, B.BigVGANConfig(upsample_initial_channel=256), mel))
sd = synth.bigvgan_state_dict()
voc = F5HipBigVGAN(sd, gemm_planes=planes)
mel = torch.randn(1, 100, 48, generator=torch.Generator().manual_seed(77)) * 1.5 - 1.0
report("full t48", voc(mel), B.bigvgan_forward(sd, B.BIGVGAN_V2_24K_100B_256X, mel))
if "--oracle936" in sys.argv:
    mel = torch.randn(2, 100, 936, generator=torch.Generator().manual_seed(78)) * 1.5 - 1.0
    t0 = time.time()
    ref = B.bigvgan_forward(sd, B.BIGVGAN_V2_24K_100B_256X, mel)
    print(f"oracle {time.time() - t0:.1f} s", flush=True)
    got = voc(mel)
    report("full t936 b2", got, ref)
    many = voc(mel[:1].expand(16, -1, -1))
    print(f"[{tag}] 16 copies vs single: {(many - got[:1]).abs().max().item():.3e}", flush=True)
if "--bench" in sys.argv:
    for b in (1, 4, 16):
        mel = (torch.randn(b, 100, 936) * 1.5 - 1.0).cuda()
        for _ in range(2):
            voc(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            voc(mel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        gbs = 9.1e9 * b / dt / 1e9
        print(f"[{tag}] bigvgan batch {b:2d}: {dt*1e3:8.2f} ms / call  {b*936/dt:9.0f} mel-frames/s  {gbs:6.0f} GB/s algorithmic = {gbs / 80:4.1f} % of 8 TB/s", flush=True)
