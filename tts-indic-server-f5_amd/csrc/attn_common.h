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
    // two-segment keys (attn3 SEG2 kernels; MMDiT joint attention, F/model/modules.py:496-514): the keys of (pseudo-)sequence s are rows
    // seq_kv_row0[s] .. + seq_kvlen[s] followed by rows seq_kv2_row0[s] .. + seq_kv2_len[s]; its queries are rows seq_row0[s] .. + seq_len[s],
    // which may be either segment.  Segment starts are multiples of 16 rows (V^T keeps tokens in vt_col order within aligned groups of 16).
    const int* seq_kv_row0;
    const int* seq_kv2_row0;
    const int* seq_kv2_len;
    __bf16* out_hi;     // [M_pad][D]
    __bf16* out_lo;     // may be null
    int f16_out;        // 1: out_hi receives one fp16 plane (A operand of the fp16 out-projection GEMM)
    unsigned long long* dbg;   // diagnostics (attn3): per-phase s_memtime totals of wave 0 of workgroup (0,0,0), or null
    int shape_invariant;       // host side (kernel choice, tu_attn.hip): 1 = one association for every launch shape, 0 = fastest kernel per shape, -1 = the process default
};

F5_DEVICE int lds_off128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N>
F5_DEVICE void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One 1 KiB LDS-DMA piece (64 lanes x 16 B) through inline asm.  With the builtin (__builtin_amdgcn_global_load_lds) in the same loop as
// the fragment reads, hipcc's wait-count model sees a FLAT-class instruction with an LDS operand ("pending flat": may return out of
// order) and degrades every s_waitcnt lgkmcnt(N) of the loop to lgkmcnt(0) -- it then waits for the fragment reads it has just issued.
// An asm statement is invisible to that model; it has no VGPR destination, and its completion is counted by hand (vmcnt).
// M0 (the LDS destination) is compiler-reserved: saved and restored inside the statement.
F5_DEVICE void attn_lds_dma16(const char* gsrc, char* lds_dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(d) : "memory");
}
