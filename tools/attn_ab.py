#!/usr/bin/env python3
"""attn3 at the config shapes, HIP events, alternating repeats (for A/B across builds on one box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import ops
SHAPES = (("C2  2 x 1404, 16 heads", (1404, 1404), 16), ("C3 share 16 x 1404", (1404,) * 16, 16), ("C4  32 x 1404", (1404,) * 32, 16), ("C5  2 x 2341", (2341, 2341), 16),
                         ("C5  16 x 2341 (batch 8)", (2341,) * 16, 16), ("ragged 16 x 900..1900", tuple(900 + 66 * i for i in range(16)), 16),
          ("ragged 16 x U(1030, 1780)", tuple(1030 + (i * 7919 + 13) % 751 for i in range(16)), 16), ("C1  2 x 748, 12 heads", (748, 748), 12))
if os.environ.get("ATTN_AB_SHAPES"):   # comma-separated indices into SHAPES
    SHAPES = tuple(SHAPES[int(i)] for i in os.environ["ATTN_AB_SHAPES"].split(","))
for tag, lens, heads in SHAPES:
    n, D = sum(lens), 64 * heads
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(n, D, generator=g).cuda() for _ in range(3))
    fl = sum(4.0 * L * L * 64 * heads for L in lens)
    us = [ops.attention(q, k, v, lens, heads=heads, impl=3, iters=200)[1] for _ in range(3)]
    print(f"{tag:26s} " + "  ".join(f"{u:8.2f} us ({fl / u / 1e6 / 2500:.3f})" for u in us), flush=True)
