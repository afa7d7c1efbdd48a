"""Builds csrc/libf5hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree on purpose: the .so travels to the GPU box with the repo snapshot.  Every kernel family is its own translation
unit (csrc/tu_*.hip + f5hip.hip), compiled in parallel into csrc/_obj/ and re-compiled only when one of the files it
includes changed; `--experiments` adds -DF5HIP_EXPERIMENTS (the measured-and-rejected kernels under csrc/experiments/ and
the f5hip_debug_* entry points the tools/ scripts of round 1 use)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libf5hip.so")
UNITS = ["f5hip.hip", "tu_gemm_reg.hip", "tu_gemm3.hip", "tu_gemm5_generic.hip", "tu_gemm5_qkv.hip", "tu_gemm6.hip", "tu_conv5.hip", "tu_attn.hip"]
HOT = ("gemm", "conv5", "attn", "ln_kernel")   # kernels that must not touch scratch memory
# Per-unit flags.  tu_attn: hipcc's SLP vectorizer turns the softmax row sums into v_pk_add_f32, which beside MFMAs costs more issue time than
# the two v_add_f32 it replaces (MI355X_MICROARCH "packed f32 VALU ... an anti-lever beside MFMAs"); measured in profiles/r02_attn_bench.txt.
UNIT_FLAGS = {"tu_attn.hip": ["-fno-slp-vectorize"]}

_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path: str, seen: set[str] | None = None) -> set[str]:
    """The file and everything it includes with "...", recursively (conditional includes count as dependencies too)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen:
        return seen
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: included by a translation unit of libf5hip but missing")
    seen.add(path)
    with open(path, encoding="utf-8") as f:
        for inc in _INC.findall(f.read()):
            _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _compile(unit: str, flags: list[str], verbose: bool):
    src, obj = os.path.join(CSRC, unit), os.path.join(OBJ, unit.replace(".hip", ".o"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -Rpass-analysis=kernel-resource-usage: per-kernel VGPR / scratch report.  A hot kernel that touches scratch pays a
    # scratch set-up per wave plus the spills (a run-time index into the by-value argument struct once cost every GEMM 7 us).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Rpass-analysis=kernel-resource-usage", *flags, *UNIT_FLAGS.get(unit, []), "-o", obj, src]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    report, name, vg = [], None, None
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[-R")[0].strip()
        elif "VGPRs:" in line and "AGPRs" not in line and name:
            vg = int(line.split("VGPRs:")[1].split()[0])
        elif "ScratchSize [bytes/lane]:" in line and name:
            report.append((name, int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]), vg))
            name = None
    if r.returncode != 0:
        sys.stderr.write("\n".join(l for l in r.stderr.splitlines() if "remark:" not in l) + "\n")
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(obj + ".resources", "w") as f:
        for n, sc, v in report:
            f.write(f"{sc:6d} B scratch  {v if v is not None else -1:4d} VGPR  {n}\n")
    return unit


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    flags = ["-DF5HIP_EXPERIMENTS"] if experiments else []
    if os.environ.get("F5HIP_BUILD_ABL"):   # ablation variants of gemm5 (diagnostics): F5HIP_GEMM5_ABL=<n> then selects one at run time
        flags.append("-DF5HIP_GEMM5_ABL")
    stamp = os.path.join(OBJ, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    stale = []
    for u in UNITS:
        obj = os.path.join(OBJ, u.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(os.path.join(CSRC, u))):
            stale.append(u)
    if stale:
        with cf.ThreadPoolExecutor(max_workers=min(len(stale), max(1, (os.cpu_count() or 2) - 1))) as ex:
            for u in ex.map(lambda u: _compile(u, flags, verbose), stale):
                pass
        with open(stamp, "w") as f:
            f.write(" ".join(flags))
    objs = [os.path.join(OBJ, u.replace(".hip", ".o")) for u in UNITS]
    if stale or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    lines = []
    for o in objs:
        if os.path.exists(o + ".resources"):
            lines += open(o + ".resources").read().splitlines()
    with open(os.path.join(CSRC, "kernel_resources.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    # (attn3's STAMPS = true instantiations exist only in --experiments builds, for tools/attn_stamps.py: not production kernels)
    hot = [l.split("VGPR", 1)[1].strip() for l in lines if not l.lstrip().startswith("0 B") and any(k in l for k in HOT)
           and not re.search(r"attn3_fwd_kernelILi\dELb\dELb1EE", l)]
    if hot:
        raise RuntimeError("hot kernels use scratch memory: " + ", ".join(hot))
    build_torch_ops(verbose)
    return LIB


TORCH_LIB = os.path.join(CSRC, "libf5hip_torch.so")


def build_torch_ops(verbose: bool = True) -> str:
    """csrc/torch_ops.cpp -> csrc/libf5hip_torch.so: the TORCH_LIBRARY operators (torch.ops.f5hip.*) over the C ABI.  Host C++ only, linked against
    libf5hip.so (rpath $ORIGIN) and the torch libraries of this interpreter; rebuilt when the source, the header or libf5hip.so changed."""
    import torch
    src, hdr = os.path.join(CSRC, "torch_ops.cpp"), os.path.join(os.path.dirname(HERE), "include", "f5hip.h")
    if os.path.exists(TORCH_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_LIB) for d in (src, hdr, LIB)):
        return TORCH_LIB
    tdir = os.path.dirname(torch.__file__)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include",
           "-I/opt/rocm/include", "-o", TORCH_LIB, src, f"-L{CSRC}", "-lf5hip", f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir}/lib"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return TORCH_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, experiments="--experiments" in sys.argv)
