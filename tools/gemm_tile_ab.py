#!/usr/bin/env python3
"""One-plane GEMM tile-shape A/B (diagnostics): gemm3 128x128 (2 workgroups / CU) vs gemm2 256x128 vs gemm2 128x128."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_double)]
torch.cuda.init()
shapes = [("qkv", 2816, 3072, 1024), ("ff1", 2816, 2048, 1024), ("ff2", 2816, 1024, 2048), ("out", 2816, 1024, 1024),
          ("ff1 B=8", 22528, 2048, 1024), ("qkv B=8", 22528, 3072, 1024), ("ff2 B=8", 22528, 1024, 2048)]
names = {30: "gemm3 128x128", 20: "gemm2 128x128", 21: "gemm2 256x128"}
for nm, M, N, K in shapes:
    for var in (30, 20, 21):
        us = C.c_double(0)
        rc = fn(M, N, K, 1, 128, var, 20, C.byref(us))
        if rc: print("ERR", L.f5hip_last_error()); continue
        print(f"{nm:9s} M{M} N{N} K{K} one plane {names[var]:15s} {us.value:8.1f} us  {2.0 * M * N * K / us.value / 1e6:7.1f} TF", flush=True)
