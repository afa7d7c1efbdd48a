"""ctypes binding of libf5hip (include/f5hip.h).  Fails loudly when the HIP library is missing: there is no
CPU path in this package."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("F5HIP_LIB") or os.path.join(HERE, "csrc", "libf5hip.so")   # F5HIP_LIB: A/B builds of the same ABI (diagnostics)

_lib = None


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "depth", "heads", "ff_mult", "text_dim", "conv_layers", "mel_dim",
                                          "text_num_embeds", "gemm_planes", "arch")]


class VocosConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "dim", "intermediate_dim", "num_layers", "n_fft", "hop_length",
                                          "gemm_planes")]


class BigVGANConfig(C.Structure):
    _fields_ = [("num_mels", C.c_int32), ("num_upsamples", C.c_int32), ("upsample_rates", C.c_int32 * 8),
                ("upsample_kernel_sizes", C.c_int32 * 8), ("upsample_initial_channel", C.c_int32),
                ("resblock_kernel_sizes", C.c_int32 * 3), ("resblock_dilations", C.c_int32 * 9), ("gemm_planes", C.c_int32)]


# every symbol include/f5hip.h declares: (restype, argtypes)
SYMBOLS = {
    "f5hip_abi_version": (C.c_int, []),
    "f5hip_last_error": (C.c_char_p, []),
    "f5hip_dit_create": (C.c_void_p, [C.POINTER(DitConfig)]),
    "f5hip_dit_destroy": (None, [C.c_void_p]),
    "f5hip_dit_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "f5hip_dit_finalize": (C.c_int, [C.c_void_p]),
    "f5hip_dit_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "f5hip_dit_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "f5hip_cfm_sample": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "f5hip_cfm_sample_masked": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "f5hip_dit_set_ode_method": (C.c_int, [C.c_void_p, C.c_int32]),
    "f5hip_dit_set_attention_shape_invariant": (C.c_int, [C.c_void_p, C.c_int32]),
    "f5hip_dit_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "f5hip_dit_get_profile": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "f5hip_set_profiling": (C.c_int, [C.c_int32]),
    "f5hip_set_attention_shape_invariant": (C.c_int, [C.c_int32]),
    "f5hip_get_profile": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "f5hip_get_counter": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "f5hip_op_gemm": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_void_p]),
    "f5hip_op_qkv": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                               C.POINTER(C.c_double), C.c_void_p]),
    "f5hip_op_attention": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_double), C.c_void_p]),
    "f5hip_op_layernorm": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "f5hip_op_joint_attention": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "f5hip_op_conv1d": (C.c_int, [C.c_int32] * 7 + [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "f5hip_vocos_create": (C.c_void_p, [C.POINTER(VocosConfig)]),
    "f5hip_vocos_destroy": (None, [C.c_void_p]),
    "f5hip_vocos_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "f5hip_vocos_finalize": (C.c_int, [C.c_void_p]),
    "f5hip_vocos_decode": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "f5hip_bigvgan_create": (C.c_void_p, [C.POINTER(BigVGANConfig)]),
    "f5hip_bigvgan_destroy": (None, [C.c_void_p]),
    "f5hip_bigvgan_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "f5hip_bigvgan_finalize": (C.c_int, [C.c_void_p]),
    "f5hip_bigvgan_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "f5hip_mel_spectrogram_bigvgan": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                C.c_int32, C.c_void_p]),
    "f5hip_mel_spectrogram": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_void_p]),
}


class F5HipError(RuntimeError):
    pass


def lib():
    """Loads libf5hip.so (once).  Raises F5HipError if it has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise F5HipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(this package has no CPU fallback)")
        # torch first: its wheel bundles the HIP runtime (same soname as /opt/rocm's).  If libf5hip.so is the first to pull in a
        # libamdhip64, the process ends up with the other copy and hipGetDeviceCount() fails inside the library.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.f5hip_abi_version() != 1:
            raise F5HipError("libf5hip ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise F5HipError(f"{what} failed ({rc}): {lib().f5hip_last_error().decode(errors='replace')}")


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device, as an int for the `stream` arguments."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
