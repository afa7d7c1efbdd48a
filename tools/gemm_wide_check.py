import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch
from tts_indic_server_f5_amd import _lib
L = _lib.lib()
fn = L.f5hip_debug_gemm_wide_check
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_double)] * 2
torch.cuda.init()
for mode, (M, N, K) in [(0, (1408, 2048, 1024)), (1, (1408, 2048, 1024)), (2, (1408, 3072, 1024)), (10, (1408, 2048, 1024)), (11, (1408, 2048, 1024)), (12, (2816, 3072, 1024)), (12, (5632, 3072, 1024))]:
    a, b = C.c_double(0), C.c_double(0)
    rc = fn(M, N, K, mode, C.byref(a), C.byref(b))
    print("mode", mode, M, N, K, "rc", rc, "max diff / differing elements", a.value, "max abs", b.value, flush=True)
