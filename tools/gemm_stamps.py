#!/usr/bin/env python3
"""Per-phase s_memtime breakdown of one workgroup of gemm_kernel (diagnostic build, ABL = 3)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_uint64)]
torch.cuda.init()
names = ["prologue", "gload issue", "lds rd+mfma", "gwait+lds wr", "barrier", "epilogue", "total"]
for nm, M, N, K, bn in [("out(176 tiles)", 2816, 1024, 1024, 128), ("ff1(352)", 2816, 2048, 1024, 128), ("qkv(528)", 2816, 3072, 1024, 128), ("4096^3", 4096, 4096, 4096, 128)]:
    for (bx, by) in [(3, 7), (-4, 7)]:   # negative bx: fp32 residual epilogue instead of gelu + split bf16
        out = (C.c_uint64 * 12)()
        rc = fn(M, N, K, bn, bx, by, out)
        if rc: print("ERR", L.f5hip_last_error()); continue
        tot = out[6]
        print(f"{nm:15s} wg({bx:2d},{by:2d}) total {tot:7d} cyc | " + "  ".join(f"{names[i]} {out[i]:6d}" for i in range(6)) + f" || epi: barrier {out[7]} slab {out[8]} rows {out[9]} drain {out[10]}", flush=True)
