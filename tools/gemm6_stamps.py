#!/usr/bin/env python3
"""In-kernel time line of gemm6 at the C3-share shapes (run-time stamps, no special build): where a 256 x 256 tile spends its time --
k-loop, the four epilogue quarters, the store drain -- and how the rounds lay out.  Also times each shape (HIP events, 20 launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

M = int(os.environ.get("M", 22528))
print("F5HIP_GEMM6 =", os.environ.get("F5HIP_GEMM6", "(auto)"), flush=True)
for name, N, K, act, out16, res in (("out", 1024, 1024, "none", False, True), ("FF1", 2048, 1024, "gelu_tanh", True, False), ("FF2", 1024, 2048, "none", False, True),
                                    ("out, K = 64 (epilogue only)", 1024, 64, "none", False, True), ("FF1, K = 64 (epilogue only)", 2048, 64, "gelu_tanh", True, False)):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    os.environ.pop("F5HIP_GEMM6_STAMPS", None)
    ops.gemm(a, w, torch.zeros(N), prec=3, act=act, res=r, mul=torch.ones(N) if res else None, out16=out16, w_copies=4, iters=max(20, int(os.environ.get("WARM", 3000))))   # seconds of load first: the clock settles
    os.environ["F5HIP_GEMM6_STAMPS"] = "1"
    _, us = ops.gemm(a, w, torch.zeros(N), prec=3, act=act, res=r, mul=torch.ones(N) if res else None, out16=out16, w_copies=4, iters=20)
    fl = 2.0 * M * N * K
    print(f"{name:30s} M {M} N {N} K {K}: {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s ({fl / us / 1e6 / 2500:.3f} of the MFMA roof)", flush=True)
