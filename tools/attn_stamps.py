#!/usr/bin/env python3
"""Per-phase s_memtime totals of wave 0 of workgroup 0 of attn3_fwd_kernel (diagnostics; needs a library built with `build.py --experiments`):
landed-wait | barrier | DMA issue | half tile 0 | half tile 1 per KV tile, the prologue (entry -> first loop step) and the wave's whole life.
s_memtime counts shader clocks; the last column converts with s_memrealtime (100 MHz) over the same span."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_attn_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
torch.cuda.init()
for n, heads, n_seq in ((748, 12, 2), (1404, 16, 2), (1404, 16, 16), (2341, 16, 16), (2816, 16, 16)):
    out = (C.c_uint64 * 9)(); us = C.c_double(0)
    rc = fn(n, heads, n_seq, 20, out, C.byref(us))
    if rc: print("ERR", L.f5hip_last_error()); continue
    nkt = max(1, out[3])
    loop = out[0] + out[1] + out[2] + out[4] + out[5]
    print(f"{n_seq:2d} x {n} heads={heads}: kernel {us.value:7.1f} us | per KV tile (wave 0): landed-wait {out[5] / nkt:6.1f}  barrier {out[4] / nkt:6.1f}  DMA issue {out[0] / nkt:6.1f}  "
          f"half 0 {out[1] / nkt:6.1f}  half 1 {out[2] / nkt:6.1f}  = {loop / nkt:6.1f} clocks/tile x {nkt} tiles = {loop} | prologue {out[6]}  whole wave {out[7]} clocks = {out[8] / 100:.2f} us ({out[7] / max(1, out[8]) / 10:.2f} GHz)", flush=True)
