#!/usr/bin/env python3
"""gemm3 ablation (diagnostics): full kernel vs no-DMA vs no-MFMA, on the batch-1 shapes and 4096^3."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_double)]
torch.cuda.init()
shapes = [("out", 2816, 1024, 1024), ("ff2", 2816, 1024, 2048), ("ff1", 2816, 2048, 1024), ("4096^3", 4096, 4096, 4096)]
names = {30: "gemm3", 31: "gemm3 no-DMA", 32: "gemm3 no-MFMA", 0: "v1", 1: "v1 no-gload", 2: "v1 no-mfma"}
for nm, M, N, K in shapes:
    for var in (30, 31, 32, 0, 1, 2):
        us = C.c_double(0)
        rc = fn(M, N, K, 2, 128, var, 20, C.byref(us))
        if rc:
            print("ERR", L.f5hip_last_error()); continue
        tf = 2.0 * M * N * K / us.value / 1e6
        print(f"{nm:9s} M{M} N{N} K{K} {names[var]:16s} {us.value:9.1f} us  {tf*3:7.1f} TF mfma", flush=True)
