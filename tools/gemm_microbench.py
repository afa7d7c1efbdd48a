#!/usr/bin/env python3
"""GEMM kernel micro-benchmark / ablation on the GPU (diagnostics): python tools/gemm_microbench.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_double)]
torch.cuda.init()
shapes = [("qkv-like", 2816, 3072, 1024), ("ff1", 2816, 2048, 1024), ("ff2", 2816, 1024, 2048), ("out", 2816, 1024, 1024),
          ("big", 22528, 2048, 1024), ("4096^3", 4096, 4096, 4096)]
names = {30: "gemm3 warp-spec", 20: "gemm2 128x128", 21: "gemm2 256x128", 100: "normal +pad", 101: "no-gload +pad", 102: "no-mfma +pad", 0: "normal", 1: "no-gload", 2: "no-mfma", 10: "normal f32-rmw epi", 11: "no-gload f32 epi", 12: "no-mfma f32 epi"}
for nm, M, N, K in shapes:
    for planes in (2, 1):
        for bn in (128, 64):
            for var in ((0, 20, 30) if bn == 128 else (0,)):
                us = C.c_double(0)
                rc = fn(M, N, K, planes, bn, var, 20, C.byref(us))
                if rc:
                    print("ERR", L.f5hip_last_error()); continue
                tf = 2.0 * M * N * K / us.value / 1e6
                print(f"{nm:9s} M{M} N{N} K{K} planes{planes} bn{bn:3d} {names[var]:20s} {us.value:9.1f} us  {tf:7.1f} TF algo  {tf*(3 if planes==2 else 1):7.1f} TF mfma", flush=True)
