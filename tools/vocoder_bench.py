#!/usr/bin/env python3
"""Vocoder timing (Vocos vs BigVGAN) at the C2 / C4 geometry (936 generated frames): python tools/vocoder_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN, F5HipVocos
voc = F5HipVocos(synth.vocos_state_dict())
bv = F5HipBigVGAN(synth.bigvgan_state_dict())
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
        print(f"{name:8s} batch {b:2d}: {dt*1e3:8.2f} ms / call   {b*936/dt:10.0f} mel-frames/s   out {tuple(w.shape)}", flush=True)
