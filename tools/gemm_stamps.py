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

# gemm3 (warp-specialised): consumer wave 0: wait for tile 0, k-loop, epilogue; several workgroups to see the spread
for nm, M, N, K in [("out", 2816, 1024, 1024), ("ff2", 2816, 1024, 2048), ("4096^3", 4096, 4096, 4096)]:
    for (bx, by) in [(0, 0), (3, 7), (7, 21), (-4, 7), (-8, 21)]:
        out = (C.c_uint64 * 12)()
        rc = fn(M, N, K, 3, bx, by, out)
        if rc: print("ERR", L.f5hip_last_error()); continue
        print(f"gemm3 {nm:7s} wg({bx:2d},{by:2d}) first-tile wait {out[0]:6d}  k-loop {out[1]:7d} ({out[1] // (K // 32)} / k-step)  epilogue {out[2]:6d}  || epi: barrier {out[7]} slab {out[8]} rows {out[9]} drain {out[10]}", flush=True)
