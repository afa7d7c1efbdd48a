#!/usr/bin/env python3
"""In-kernel time line of gemm5: s_memrealtime stamps of every workgroup, printed by f5hip_op_gemm / f5hip_op_qkv.  The W-direct kernels
(FF1, QKV) stamp at run time; the others need a library built with F5HIP_BUILD_ABL=1."""
import os
import sys

os.environ["F5HIP_GEMM5_ABL"] = "5"
os.environ["F5HIP_GEMM5_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

M = 2816
for N, K, act, out16, res in ((1024, 1024, "none", False, True), (1024, 64, "none", False, True), (2048, 1024, "gelu_tanh", True, False), (2048, 64, "gelu_tanh", True, False)):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    ops.gemm(a, w, torch.zeros(N), prec=3, act=act, res=r, mul=torch.ones(N) if res else None, out16=out16, w_copies=32)

D = 1024
g = torch.Generator().manual_seed(2)
a = torch.randn(M, D, generator=g).cuda()
w = (torch.randn(3 * D, D, generator=g) / 32).cuda()
ops.qkv(a, w, torch.zeros(3 * D), [i % 1404 for i in range(M)], prec=3)
