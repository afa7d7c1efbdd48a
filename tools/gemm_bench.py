#!/usr/bin/env python3
"""Times the production GEMM kernels in isolation through the f5hip_op_gemm / f5hip_op_qkv unit ops (HIP events, HBM-cold weights:
the packed weight cycles through a pool larger than the Infinity Cache like consecutive layers do in the real forward).

  python tools/gemm_bench.py            C2 block-GEMM shapes + a K sweep (fixed cost vs per-k-step cost)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

DEV = "cuda:0"


def run(tag, M, N, K, **kw):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    copies = int(os.environ["COPIES"]) if os.environ.get("COPIES") else max(1, min(64, int(600e6 // (N * K * 2))))   # COPIES=1: weights stay warm
    res = torch.randn(M, N, generator=g).to(DEV) if kw.pop("res", False) else None
    mul = torch.randn(N, generator=g).to(DEV) if res is not None else None
    _, us = ops.gemm(a, w, torch.zeros(N), res=res, mul=mul, w_copies=copies, iters=200, **kw)
    fl = 2.0 * M * N * K
    print(f"{tag:28s} M{M:6d} N{N:5d} K{K:5d}  {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s  ({fl / us / 1e6 / 2500:.3f} of the bf16 MFMA roof)", flush=True)


def main():
    M = int(os.environ.get("M", 2816))
    run("out  (res, gate)", M, 1024, 1024, prec=3, res=True)
    run("FF1  (gelu, fp16 out)", M, 2048, 1024, prec=3, act="gelu_tanh", out16=True)
    run("FF2  (res, gate)", M, 1024, 2048, prec=3, res=True)
    g = torch.Generator().manual_seed(2)
    D = 1024
    a = torch.randn(M, D, generator=g).to(DEV)
    w = (torch.randn(3 * D, D, generator=g) / 32).to(DEV)
    *_, us = ops.qkv(a, w, torch.zeros(3 * D), [i % 1404 for i in range(M)], prec=3, iters=200)
    fl = 2.0 * M * 3 * D * D
    print(f"{'QKV  (rotary, V^T) hot W':28s} M{M:6d} N{3 * D:5d} K{D:5d}  {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s  ({fl / us / 1e6 / 2500:.3f} of the bf16 MFMA roof)", flush=True)
    print("# K sweep, out-projection epilogue: the intercept is the fixed cost (launch, first tile, epilogue)")
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        run("out K sweep", M, 1024, K, prec=3, res=True)
    print("# K sweep, FF1 epilogue")
    for K in (64, 256, 1024, 4096):
        run("FF1 K sweep", M, 2048, K, prec=3, act="gelu_tanh", out16=True)
    if os.environ.get("F5HIP_GEMM_IMPL") is None:
        print("# plain fp32-out, no residual")
        run("plain N1024", M, 1024, 1024, prec=3)


if __name__ == "__main__":
    main()
