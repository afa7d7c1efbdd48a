#!/usr/bin/env python3
"""A few launches of the four transformer-block GEMMs at the C2 shape (M = 2816) through the unit ops, for rocprofv3 --pmc collection:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
            SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_gemm -o g -- python3 tools/gemm_pmc.py
then `python tools/attn_pmc.py --summarise gpurun_out/pmc_gemm gemm5` prints per-kernel means."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

DEV = "cuda:0"
M = int(os.environ.get("M", 2816))
g = torch.Generator().manual_seed(1)


def run(N, K, **kw):
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV) if kw.pop("res", False) else None
    mul = torch.randn(N, generator=g).to(DEV) if res is not None else None
    ops.gemm(a, w, torch.zeros(N), res=res, mul=mul, w_copies=8, iters=10, **kw)


run(1024, 1024, prec=3, res=True)                       # out projection
run(2048, 1024, prec=3, act="gelu_tanh", out16=True)    # FF1
run(1024, 2048, prec=3, res=True)                       # FF2
a = torch.randn(M, 1024, generator=g).to(DEV)
w = (torch.randn(3072, 1024, generator=g) / 32).to(DEV)
ops.qkv(a, w, torch.zeros(3072), [i % 1404 for i in range(M)], prec=3, iters=10)
torch.cuda.synchronize()
print("done")
