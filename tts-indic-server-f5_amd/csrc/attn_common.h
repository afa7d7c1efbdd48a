// Shared declarations of the attention kernels (argument block, LDS swizzle of 128-byte rows, counted vmcnt wait).
#pragma once
#include "common.h"

struct AttnArgs {
    const __bf16* qk;   // [M_pad][2 D]   q | k
    const __bf16* vt;   // [D][ldvt]
    int D, ldvt;
    const int* seq_row0;
    const int* seq_len;
    const int* seq_kvlen;
    __bf16* out_hi;     // [M_pad][D]
    __bf16* out_lo;     // may be null
    int f16_out;        // 1: out_hi receives one fp16 plane (A operand of the fp16 out-projection GEMM)
    unsigned long long* dbg;   // diagnostics (attn3): per-phase s_memtime totals of wave 0 of workgroup (0,0,0), or null
};

F5_DEVICE int lds_off128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N>
F5_DEVICE void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
