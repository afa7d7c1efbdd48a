#!/usr/bin/env python3
"""Times the attention kernels in isolation (f5hip_op_attention, HIP events): attn4 (production) vs attn3 (round 1) at the config shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

for tag, lens, heads in (("C2  2 x 1404, 16 heads", (1404, 1404), 16), ("C3 share 16 x 1404", (1404,) * 16, 16), ("C5  2 x 2341", (2341, 2341), 16),
                         ("C1  2 x 748, 12 heads", (748, 748), 12)):
    n, D = sum(lens), 64 * heads
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(n, D, generator=g).cuda() for _ in range(3))
    fl = sum(4.0 * L * L * 64 * heads for L in lens)
    for impl in (3,):
        _, us = ops.attention(q, k, v, lens, heads=heads, impl=impl, iters=100)
        print(f"{tag:26s} attn{impl}: {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s ({fl / us / 1e6 / 2500:.3f} of the bf16 MFMA roof)", flush=True)
