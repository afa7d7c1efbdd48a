#!/usr/bin/env python3
"""gemm5 ablation (needs a library built with F5HIP_BUILD_ABL=1): re-runs a K sweep in child processes with
F5HIP_GEMM5_ABL = 0..4 (0 full kernel, 1 no MFMAs, 2 no fragment reads + no MFMAs, 3 no LDS-DMA in the loop, 4 MFMAs only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(%r))
from tts_indic_server_f5_amd import ops
M = 2816
for N, act, out16, res in ((1024, "none", False, True), (2048, "gelu_tanh", True, False)):
    for K in (64, 1024, 4096):
        g = torch.Generator().manual_seed(1)
        a = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        r = torch.randn(M, N, generator=g).cuda() if res else None
        _, us = ops.gemm(a, w, torch.zeros(N), prec=3, act=act, res=r, out16=out16, w_copies=max(1, min(64, int(600e6 // (N * K * 2)))), iters=200)
        print(f"ABL {os.environ.get('F5HIP_GEMM5_ABL', '0')}  N {N:5d} K {K:5d}  {us:8.2f} us", flush=True)
''' % HERE
for abl in range(5):
    env = dict(os.environ, F5HIP_GEMM5_ABL=str(abl))
    subprocess.run([sys.executable, "-c", CODE], env=env, check=False)
