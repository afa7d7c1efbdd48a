#!/usr/bin/env python3
"""Vocoder timing (Vocos vs BigVGAN) at the C2 / C4 geometry (936 generated frames): python tools/vocoder_bench.py
The GB/s column is the algorithmic activation traffic of BASELINE.md section 2 (Vocos 55 MB, BigVGAN v2 9.1 GB ideal fused fp32 per
936-frame decode) over the measured time, next to the 8 TB/s HBM3E peak."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN, F5HipVocos
voc = F5HipVocos(synth.vocos_state_dict())
bv = F5HipBigVGAN(synth.bigvgan_state_dict())
ALGO_BYTES = {"vocos": 55e6, "bigvgan": 9.1e9}
for b in (1, 4, 16):
    mel = (torch.randn(b, 100, 936) * 1.5 - 1.0).cuda()
    for name, fn in (("vocos", voc.decode), ("bigvgan", bv)):
        for _ in range(2):
            fn(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            w = fn(mel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        gbs = ALGO_BYTES[name] * b / dt / 1e9
        print(f"{name:8s} batch {b:2d}: {dt*1e3:8.2f} ms / call   {b*936/dt:10.0f} mel-frames/s   {gbs:6.0f} GB/s algorithmic = {gbs / 80:4.1f} % of 8 TB/s   out {tuple(w.shape)}", flush=True)
