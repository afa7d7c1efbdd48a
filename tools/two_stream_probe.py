#!/usr/bin/env python3
"""Upper bound of "the two CFG branches as two half-height chains on two HIP streams"  without new
library code: a CFG pair is two independent 1404-row sequences until the Euler update, so TWO sampler handles, each running ONE branch
(cfg_strength = 0: M = 1408 rows, 128 exact-fit tiles per GEMM launch) from its own thread on its own stream, are exactly the two chains
-- minus the per-step join, which could only cost more.  Compared with the ONE 2 x 1404-row chain the library runs (256 tiles per launch).
If the concurrent pair is not faster than the single chain, the idea is dead.  usage: python tools/two_stream_probe.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import synth  # noqa: E402
from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel  # noqa: E402

sd = synth.dit_state_dict()
g = torch.Generator().manual_seed(14)
cond = torch.randn(1, 469, 100, generator=g).cuda()
text = synth.text_ids()
y0 = synth.noise(1404, 0)[None].cuda()
kw = dict(steps=32, sway_sampling_coef=-1.0, y0=y0)
pair = F5HipModel(F5TTS_BASE, sd)
halves = [F5HipModel(F5TTS_BASE, sd), F5HipModel(F5TTS_BASE, sd)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run_pair(n):
    for _ in range(n):
        pair.sample(cond, text, 1404, cfg_strength=2.0, **kw)
    torch.cuda.synchronize()


def run_half(i, n):
    with torch.cuda.stream(streams[i]):
        for _ in range(n):
            halves[i].sample(cond, text, 1404, cfg_strength=0.0, **kw)
    streams[i].synchronize()


def timed(fn, *a):
    t0 = time.perf_counter()
    fn(*a)
    return time.perf_counter() - t0


run_pair(2); run_half(0, 2); run_half(1, 2)
N = 8
for rep in range(3):
    t_pair = timed(run_pair, N) / N
    t_one = timed(run_half, 0, N) / N
    th = [threading.Thread(target=run_half, args=(i, N)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / N
    print(f"rep {rep}: one chain of 2 x 1404 rows (CFG pair, what ships) {t_pair * 1e3:7.2f} ms | one branch alone (1404 rows, half the chip's tiles) {t_one * 1e3:7.2f} ms | "
          f"two branches concurrently on two streams {t_two * 1e3:7.2f} ms  -> two-stream / shipped = {t_two / t_pair:.3f}", flush=True)
