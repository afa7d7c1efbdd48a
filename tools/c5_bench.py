#!/usr/bin/env python3
"""BASELINE configs[4] (C5): E2-TTS Base (UNetT), 64 NFE, CFG 2, sway -1, 60 s long-form text = chunks of N = 2340 frames (469 reference +
1871 generated), B chunks in one sampler call.  python tools/c5_bench.py [B ...]  -> generated mel-frames/s in mixed and bf16x3 mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.model import E2TTS_BASE, F5HipModel
sd = synth.unett_state_dict()
N, REF = 2340, 469
for planes, tag in ((3, "mixed (fp16 block GEMMs)"), (2, "bf16x3 everywhere")):
    m = F5HipModel(E2TTS_BASE, sd, gemm_planes=planes)
    for b in [int(a) for a in sys.argv[1:]] or [1, 8]:
        cond = torch.randn(1, REF, 100).expand(b, -1, -1)
        text = synth.text_ids(60, 240).expand(b, -1)
        y0 = [synth.noise(N, i) for i in range(b)]
        kw = dict(steps=64, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
        m.sample(cond, text, N, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            m.sample(cond, text, N, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"E2-Base 64 NFE N={N} x {b} chunk(s), {tag}: {dt*1e3:8.1f} ms  {b * (N - REF) / dt:9.0f} generated mel-frames/s  RTF {dt / (b * (N - REF) * 256 / 24000):.4f}", flush=True)
    del m
