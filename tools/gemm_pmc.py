#!/usr/bin/env python3
"""Runs a few GEMM launches for rocprofv3 --pmc collection (diagnostics)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_double)]
torch.cuda.init()
us = C.c_double(0)
for var in (0, 20):
    fn(4096, 4096, 4096, 2, 128, var, 2, C.byref(us))
    fn(2816, 2048, 1024, 2, 128, var, 2, C.byref(us))
print("done")
