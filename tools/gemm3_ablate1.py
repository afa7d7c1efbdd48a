#!/usr/bin/env python3
"""One-plane gemm3 ablation (diagnostics): full | no DMA | no MFMA | no MFMA + cache-hot DMA addresses; and the split-bf16 cache-hot arm."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_double)]
torch.cuda.init()
shapes = [("out", 2816, 1024, 1024), ("ff1", 2816, 2048, 1024), ("qkv", 2816, 3072, 1024), ("ff1 B=8", 22528, 2048, 1024), ("4096^3", 4096, 4096, 4096)]
arms = [(1, 30, "one plane: full"), (1, 33, "one plane: no DMA"), (1, 34, "one plane: no MFMA"), (1, 35, "one plane: no MFMA, hot addresses"),
        (2, 30, "split bf16: full"), (2, 32, "split bf16: no MFMA"), (2, 36, "split bf16: no MFMA, hot addresses")]
for nm, M, N, K in shapes:
    for planes, var, label in arms:
        us = C.c_double(0)
        rc = fn(M, N, K, planes, 128, var, 20, C.byref(us))
        if rc: print("ERR", L.f5hip_last_error()); continue
        fill = (M // 128) * (N // 128) * (K // 32) * (16384 * planes) / 1e6
        print(f"{nm:8s} {label:36s} {us.value:8.1f} us   LDS fill {fill:7.1f} MB -> {fill / us.value:6.2f} TB/s chip, {fill * 1e6 / us.value / 1e-6 / 256 / 2.1e9:5.1f} B/clk/CU", flush=True)
