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
HEADERS = ["common.h", "gemm.h", "gemm2.h", "gemm3.h", "gemm4.h", "gemm_epilogue.h", "attn.h", "attn2.h", "attn3.h", "debug_bench.h", "elementwise.h", "host_util.h", "vocos.h", "bigvgan.h",
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
    # -Rpass-analysis=kernel-resource-usage: per-kernel VGPR / scratch report.  A hot kernel that touches scratch pays a
    # scratch set-up per wave plus the spills (a run-time index into the by-value argument struct once cost every GEMM 7 us).
    cmd.insert(-1, "-Rpass-analysis=kernel-resource-usage")
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    report, name = [], None
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[-R")[0].strip()
        elif "ScratchSize [bytes/lane]:" in line and name:
            report.append((name, int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])))
            name = None
    if r.returncode != 0:
        sys.stderr.write("\n".join(l for l in r.stderr.splitlines() if "remark:" not in l) + "\n")
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(os.path.join(CSRC, "kernel_resources.txt"), "w") as f:
        for n, sc in report:
            f.write(f"{sc:6d} B scratch  {n}\n")
    hot = [n for n, sc in report if sc and any(k in n for k in ("gemm", "attn", "ln_kernel"))]
    if hot:
        raise RuntimeError("hot kernels use scratch memory: " + ", ".join(hot))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
