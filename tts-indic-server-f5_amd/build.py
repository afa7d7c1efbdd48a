"""Builds csrc/libf5hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree on purpose: the .so travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libf5hip.so")
SOURCES = ["f5hip.hip"]
HEADERS = ["common.h", "gemm.h", "gemm2.h", "gemm3.h", "gemm_epilogue.h", "attn.h", "attn2.h", "debug_bench.h", "elementwise.h", "host_util.h", "vocos.h", "bigvgan.h",
           os.path.join("..", "..", "include", "f5hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB,
           os.path.join(CSRC, "f5hip.hip")]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
