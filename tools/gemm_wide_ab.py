import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_double)]
torch.cuda.init()
for nm, M, N, K in [("ff1", 2816, 2048, 1024), ("qkv", 2816, 3072, 1024), ("ff2", 2816, 1024, 2048), ("ff1 B=4", 11264, 2048, 1024), ("ff1 B=8", 22528, 2048, 1024), ("qkv B=8", 22528, 3072, 1024), ("ff2 B=8", 22528, 1024, 2048)]:
    for var, label in ((30, "128x128"), (37, "128x256")):
        us = C.c_double(0)
        rc = fn(M, N, K, 1, 128, var, 20, C.byref(us))
        if rc: print("ERR", L.f5hip_last_error()); continue
        print(f"{nm:8s} one plane gemm3 {label}: {us.value:8.1f} us  {2.0*M*N*K/us.value/1e6:7.1f} TF", flush=True)
