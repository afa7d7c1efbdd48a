"""torch.ops.f5hip.*: the hot path as PyTorch custom operators (TORCH_LIBRARY registration in csrc/torch_ops.cpp over the C ABI of libf5hip;
north_star: "host Python calling HIP through PyTorch-ROCm custom ops").  `load()` registers them (once); `F5HipModel.sample`, `F5HipVocos.decode`
and `F5HipBigVGAN.__call__` go through them when the extension is present and through ctypes otherwise -- the same C entry points either way."""
from __future__ import annotations

import os

from . import _lib

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libf5hip_torch.so")
_loaded = None


def load() -> bool:
    """True when torch.ops.f5hip is usable.  Loads libf5hip.so first (the operators are linked against it)."""
    global _loaded
    if _loaded is None:
        _loaded = False
        if os.path.exists(TORCH_LIB_PATH) and os.environ.get("F5HIP_TORCH_OPS", "1") != "0":
            import torch
            _lib.lib()
            torch.ops.load_library(TORCH_LIB_PATH)
            _loaded = True
    return _loaded


def ops():
    import torch
    if not load():
        raise _lib.F5HipError(f"{TORCH_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return torch.ops.f5hip
