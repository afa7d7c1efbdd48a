#!/usr/bin/env python3
"""Per-phase s_memtime totals of one wave of attn3_fwd_kernel (diagnostics): ring wait + barrier | QK^T issue | softmax + PV."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_attn_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
torch.cuda.init()
for n, heads in ((748, 12), (1404, 16), (2816, 16)):
    out = (C.c_uint64 * 6)(); us = C.c_double(0)
    rc = fn(n, heads, 20, out, C.byref(us))
    if rc: print("ERR", L.f5hip_last_error()); continue
    nkt = max(1, out[3])
    print(f"N={n} heads={heads}: kernel {us.value:7.1f} us | per KV tile (wave 0): landed-wait {out[5] // nkt:5d}  barrier {out[4] // nkt:5d}  DMA issue {out[0] // nkt:5d}  half 0 {out[1] // nkt:5d}  half 1 {out[2] // nkt:5d} (s_memtime ticks)  ({nkt} tiles)", flush=True)
