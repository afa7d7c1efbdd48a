// Shared epilogue of the MFMA GEMM kernels (gemm.h, gemm2.h).
//
// A wave holds TM x TN accumulator tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16 C/D layout: column on the lane,
// rows (g&3) + 8 (g>>2) + 4 (lane>>5) in register g).  Each 32 x (32 TN) slice is transposed through a wave-private
// LDS slab so that every lane owns 4 consecutive columns of a row: residual / output traffic becomes 16-byte (fp32)
// and 8-byte (bf16) row-contiguous accesses, all residual loads are issued before any arithmetic, and the rotary
// pairs of the QKV epilogue are lane-local.
#pragma once
#include "common.h"
#include "ln_row.h"

enum { EPI_GENERIC = 0, EPI_QKV = 1 };

// 64-byte LDS rows (32-deep k-steps of gemm.h / gemm3.h): 16-byte chunk XOR-swizzled with (row >> 2) & 3
F5_DEVICE int lds_off2(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// Opt-in to more than 64 KiB of dynamic LDS.  The attribute is per device, so one bit per device ordinal is kept per kernel
// (a process-wide flag left a second device's launches without the opt-in).
static inline hipError_t f5_set_lds_attr(const void* fn, int bytes, unsigned& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (done_mask >> (dev & 31) & 1u) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done_mask |= 1u << (dev & 31);
    return e;
}

template <int N>
F5_DEVICE void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct GemmArgs {
    const __bf16* A[2];
    int lda;
    const __bf16* W[2];
    int ldw;              // row stride of W in elements (>= K)
    const __bf16* Wf;     // W[0] once more in MFMA-fragment order (gemm5 W-direct kernels), or null: block (16 rows, 32 k) = 64 lanes x 8 elements,
                          // lane = (row & 15) + 16 (k chunk of 8), offset ((row / 16) * (K / 32) + k / 32) * 512 elements
    int M, N, K;
    // implicit-GEMM conv (gemm.h only)
    int conv_kpt;         // k-tiles (of 32 channels) per tap
    int conv_center;      // (kernel_size - 1) / 2
    int conv_group_cols;  // A column base = blockIdx.x * conv_group_cols (grouped conv with BN == group width)
    int conv_dil;         // tap spacing in rows (0 is treated as 1): dilated Conv1d
    const int* row_seq_start;   // per-row sequence bounds, or null: uniform sequences of seq_pitch rows with seq_valid valid ones
    const int* row_seq_end;
    int seq_pitch, seq_valid;
    int group_w;          // > 0: N is laid out as groups padded to 64 columns; real column = (n/64)*group_w + n%64
    // generic epilogue: v = act(acc + bias); rows with row_keep == 0 -> 0; v = v * mul + res; store fp32 and/or split bf16
    const float* bias;
    const int* row_keep;
    int act;
    const float* mul;
    const float* res;
    int ldres;
    float* out_f32;
    int ldo;
    __bf16* out_hi;
    __bf16* out_lo;
    int ldob;
    int f16_out;          // 1: out_hi receives ONE fp16 plane (A operand of a PREC_F16 GEMM); only for the (no residual, no fp32 output) epilogues
    // QKV epilogue
    int D;                   // model dim (N == 3 D)
    const int* row_pos;      // [M_pad] frame index inside the row's sequence, or null: rope_cos / rope_sin are per ROW ([M_pad][32], gathered once per call)
    const float* rope_cos;   // [max_pos][32]
    const float* rope_sin;
    __bf16* qk;              // [M_pad][2 D]  fp16 bits (attention operands are fp16: attn3.h)
    __bf16* vt;              // [D][ldvt]     fp16 bits
    int ldvt;
    // fused LayerNorm behind the epilogue (gemm5 LNE kernels): ln.x == out_f32 (the residual stream this GEMM updates); ln_sync[slab] counts
    // the workgroups of a row slab that have stored their tile (monotonic: this launch waits for ln_target), *ln_err is set on a time-out
    LnArgs ln;
    unsigned* ln_sync;
    unsigned ln_target;
    int* ln_err;
    // diagnostics (ABL == 3 builds only)
    unsigned long long* stamps;
    int stamp_bx, stamp_by;
};

// one 32 x WN slice staged in `stg` (fp32, row stride WN): generic epilogue, lane owns 4 columns of 32 / RPP rows.
// RES / OUTF / OUTS (residual present, fp32 output, split-bf16 output) are compile-time so the hot variants carry no
// per-element pointer tests; row pointers advance incrementally (one 64-bit add per row group instead of a 64-bit multiply).
// GUARD = false is the interior fast path (whole wave tile inside M x N, no column groups, no row_keep): straight-line code, so
// the compiler counts vmcnt exactly -- all residual loads in flight, stores never waited on.  With per-row-group exec
// branches (GUARD = true) it falls back to s_waitcnt vmcnt(0) per group, which serialises every store's latency.
template <int ACT, int WN, int ROWS, bool RES, bool OUTF, int OUTS, bool GUARD, int SLD = WN>   // OUTS: 0 none, 1 split bf16, 2 one fp16 plane; SLD: row stride of `stg` in floats
F5_DEVICE void epi_generic_rows_t(const GemmArgs& p, const float* stg, int m_base, int n_base, int lane) {
    constexpr int LPR = WN / 4, RPP = 64 / LPR, NQ = ROWS / RPP;
    const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
    const int n = n_base + c4;   // column in the (possibly group-padded) weight layout
    int no = n;                  // column in the output / residual / multiplier
    bool nok = GUARD ? n < ((p.N + 3) & ~3) : true;
    if (GUARD && p.group_w) {
        nok = nok && (n & 63) < p.group_w;
        no = (n >> 6) * p.group_w + (n & 63);
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, mv = {1.f, 1.f, 1.f, 1.f};
    if (p.bias && nok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.mul && nok) mv = *reinterpret_cast<const f32x4*>(p.mul + no);
    const int mrow = m_base + r0;
    const float* rp = RES ? p.res + (size_t)mrow * p.ldres + no : nullptr;
    f32x4 rs[NQ];
    int keep[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        rs[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        keep[q] = 1;
        if (RES) {
            if (!GUARD || (nok && mrow + q * RPP < p.M)) rs[q] = *reinterpret_cast<const f32x4*>(rp + (size_t)q * RPP * p.ldres);
        }
        if (GUARD && p.row_keep) keep[q] = p.row_keep[mrow + q * RPP];
    }
    float* of = OUTF ? p.out_f32 + (size_t)mrow * p.ldo + no : nullptr;
    __bf16* oh = OUTS ? p.out_hi + (size_t)mrow * p.ldob + no : nullptr;
    __bf16* ol = OUTS == 1 && (!GUARD || p.out_lo) ? p.out_lo + (size_t)mrow * p.ldob + no : nullptr;
    const size_t sf = (size_t)RPP * p.ldo, sb = (size_t)RPP * p.ldob;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + (q * RPP + r0) * SLD + c4) + bv;
        if (ACT != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = apply_act(v[e], ACT);
        }
        if (GUARD && !keep[q]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        v = v * mv + rs[q];
        if (!GUARD || (nok && mrow + q * RPP < p.M)) {
            if (OUTF) *reinterpret_cast<f32x4*>(of + q * sf) = v;
            if (OUTS == 2) {
                const float vv[4] = {v[0], v[1], v[2], v[3]};
                store_f16x4(oh + q * sb, vv);
            } else if (OUTS == 1) {
                bf16x4 hi, lo;
                const float vv[4] = {v[0], v[1], v[2], v[3]};
                split_bf16x4(vv, hi, lo);
                *reinterpret_cast<bf16x4*>(oh + q * sb) = hi;
                if (!GUARD || ol) *reinterpret_cast<bf16x4*>(ol + q * sb) = lo;
            }
        }
    }
}

template <int ACT, int WN, int ROWS, bool GUARD, int SLD = WN>
F5_DEVICE void epi_generic_rows_g(const GemmArgs& p, const float* stg, int m_base, int n_base, int lane) {
    const bool res = p.res != nullptr, outf = p.out_f32 != nullptr, outs = p.out_hi != nullptr;
    if (ACT == ACT_NONE) {   // residual / plain projections: every output combination occurs
        if (res) {
            if (outf && outs) epi_generic_rows_t<ACT, WN, ROWS, true, true, 1, GUARD, SLD>(p, stg, m_base, n_base, lane);
            else if (outf) epi_generic_rows_t<ACT, WN, ROWS, true, true, 0, GUARD, SLD>(p, stg, m_base, n_base, lane);
            else epi_generic_rows_t<ACT, WN, ROWS, true, false, 1, GUARD, SLD>(p, stg, m_base, n_base, lane);
        } else {
            if (outf && outs) epi_generic_rows_t<ACT, WN, ROWS, false, true, 1, GUARD, SLD>(p, stg, m_base, n_base, lane);
            else if (outf) epi_generic_rows_t<ACT, WN, ROWS, false, true, 0, GUARD, SLD>(p, stg, m_base, n_base, lane);
            else if (p.f16_out) epi_generic_rows_t<ACT, WN, ROWS, false, false, 2, GUARD, SLD>(p, stg, m_base, n_base, lane);
            else epi_generic_rows_t<ACT, WN, ROWS, false, false, 1, GUARD, SLD>(p, stg, m_base, n_base, lane);
        }
    } else {                 // activations: (no residual -> split or fp32) and (residual -> fp32) are the combinations in use
        if (res) epi_generic_rows_t<ACT, WN, ROWS, true, true, 0, GUARD, SLD>(p, stg, m_base, n_base, lane);
        else if (outs && !outf && p.f16_out) epi_generic_rows_t<ACT, WN, ROWS, false, false, 2, GUARD, SLD>(p, stg, m_base, n_base, lane);
        else if (outs && !outf) epi_generic_rows_t<ACT, WN, ROWS, false, false, 1, GUARD, SLD>(p, stg, m_base, n_base, lane);
        else if (outf && !outs) epi_generic_rows_t<ACT, WN, ROWS, false, true, 0, GUARD, SLD>(p, stg, m_base, n_base, lane);
        else epi_generic_rows_t<ACT, WN, ROWS, false, true, 1, GUARD, SLD>(p, stg, m_base, n_base, lane);
    }
}

template <int ACT, int WN, int ROWS, int SLD = WN>
F5_DEVICE void epi_generic_rows(const GemmArgs& p, const float* stg, int m_base, int n_base, int lane) {
    // wave-uniform: interior tile with both split planes (or none) and no per-row / per-group special cases
    const bool interior = m_base + ROWS <= p.M && n_base + WN <= p.N && !p.group_w && !p.row_keep && (!p.out_hi || p.out_lo || p.f16_out);
    if (interior) epi_generic_rows_g<ACT, WN, ROWS, false, SLD>(p, stg, m_base, n_base, lane);
    else epi_generic_rows_g<ACT, WN, ROWS, true, SLD>(p, stg, m_base, n_base, lane);
}

// Q / K blocks of the fused QKV projection: bias, rotary embedding on head 0 (x-transformers interleaved pairs, applied
// before the head split: F/model/modules.py:414-419), q * log2(e) / 8 (softmax scale, base-2 exponents), fp16 row-major [M][2 D]
template <int WN, int ROWS, bool ROT, bool GUARD, int SLD = WN>
F5_DEVICE void epi_qk_rows_t(const GemmArgs& p, const float* stg, int m_base, int n_base, int lane) {
    constexpr int LPR = WN / 4, RPP = 64 / LPR, NQ = ROWS / RPP;
    const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
    const int n = n_base + c4;
    const int D = p.D;
    const int which = n_base / D;    // 0 q, 1 k (uniform per wave: D % 64 == 0)
    const int nd = n - which * D;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    float2 cs[NQ], sn[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        cs[q] = make_float2(1.f, 1.f);
        sn[q] = make_float2(0.f, 0.f);
        if (ROT && nd < 64) {   // (a wave tile wider than one head -- WN = 128 -- rotates only its lanes inside head 0)
            const int pos = p.row_pos ? p.row_pos[m_base + q * RPP + r0] : m_base + q * RPP + r0;   // (row_pos == null: per-row tables)
            cs[q] = *reinterpret_cast<const float2*>(p.rope_cos + pos * 32 + (nd >> 1));
            sn[q] = *reinterpret_cast<const float2*>(p.rope_sin + pos * 32 + (nd >> 1));
        }
    }
    const float qs = which == 0 ? F5_Q_SCALE : 1.0f;
    __bf16* op = p.qk + (size_t)(m_base + r0) * (2 * D) + which * D + nd;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (q * RPP + r0) * SLD + c4) + bv;
        float o[4];
        if (ROT) {
            // explicit product + fma: left to the compiler, the contraction of a*c - b*s differs between template instantiations (64- vs
            // 128-wide wave tiles), a 1-ulp flip in a few q values that the 22-layer sampler amplifies to 5e-4 (tools/wide_pipeline_check.py)
            o[0] = __builtin_fmaf(v[0], cs[q].x, -__fmul_rn(v[1], sn[q].x)) * qs;
            o[1] = __builtin_fmaf(v[1], cs[q].x, __fmul_rn(v[0], sn[q].x)) * qs;
            o[2] = __builtin_fmaf(v[2], cs[q].y, -__fmul_rn(v[3], sn[q].y)) * qs;
            o[3] = __builtin_fmaf(v[3], cs[q].y, __fmul_rn(v[2], sn[q].y)) * qs;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = v[e] * qs;
        }
        if (!GUARD || m_base + q * RPP + r0 < p.M) store_f16x4(op + (size_t)q * RPP * (2 * D), o);   // fp16, saturating
    }
}

template <int WN, int ROWS, int SLD = WN>
F5_DEVICE void epi_qk_rows(const GemmArgs& p, const float* stg, int m_base, int n_base, int lane) {
    const int which = n_base / p.D;
    const bool rot = n_base - which * p.D < 64;   // head 0 only (wave-uniform)
    if (m_base + ROWS <= p.M) {
        if (rot) epi_qk_rows_t<WN, ROWS, true, false, SLD>(p, stg, m_base, n_base, lane);
        else epi_qk_rows_t<WN, ROWS, false, false, SLD>(p, stg, m_base, n_base, lane);
    } else {
        if (rot) epi_qk_rows_t<WN, ROWS, true, true, SLD>(p, stg, m_base, n_base, lane);
        else epi_qk_rows_t<WN, ROWS, false, true, SLD>(p, stg, m_base, n_base, lane);
    }
}

// Whole-wave epilogue.  `slab` = wave-private LDS (32 TM x 32 TN floats); m_wave / n_wave = first row / (padded) column of
// the wave's sub-tile; n_blk = first column of the workgroup tile (uniform per workgroup, selects q/k vs v).
// Every wave of the workgroup must call this: one workgroup barrier protects the k-loop stages the slabs alias; after it
// the slab is wave-private, DS operations of one wave execute in order, so a wavefront-scope fence (compiler ordering
// only, no instruction) is all that separates the transposing writes from the row reads.  The whole sub-tile is staged
// at once so the accumulators are dead before the row phase (its residual prefetch needs their registers).
template <int EPI, int TM, int TN, bool BAR = true>   // BAR = false: the slab does not alias the k-loop stages (gemm4.h), no workgroup barrier
F5_DEVICE void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], float* slab, int m_wave, int n_wave, int n_blk, int lane,
                              unsigned long long* dbg = nullptr) {
    constexpr int WN = TN * 32, ROWS = TM * 32;
    unsigned long long dprev = 0;
#define EPI_STAMP(IDX)                                                                               \
    if (dbg) {                                                                                       \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if ((IDX) >= 0) dbg[(IDX)] += t_ - dprev;                                                    \
        dprev = t_;                                                                                  \
    }
    EPI_STAMP(-1);
    const int fr = lane & 31, fh = lane >> 5;
    if (BAR) __syncthreads();
    EPI_STAMP(0);
    if (EPI == EPI_QKV && n_blk >= 2 * p.D) {
        // V block: written transposed ([feature][token]) straight from the accumulators, 4 tokens = 8 bytes per store
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int nd = n_wave - 2 * p.D + j * 32 + fr;
                const float bv = p.bias[n_wave + j * 32 + fr];
#pragma unroll
                for (int a4 = 0; a4 < 4; a4++) {
                    float pk[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) pk[e] = acc[i][j][a4 * 4 + e] + bv;
                    store_f16x4(p.vt + (size_t)nd * p.ldvt + vt_col(m_wave + i * 32 + 8 * a4 + 4 * fh), pk);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) slab[(i * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh) * WN + j * 32 + fr] = acc[i][j][g];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    EPI_STAMP(1);
    if (EPI == EPI_GENERIC) {
        switch (p.act) {
            case ACT_GELU_TANH: epi_generic_rows<ACT_GELU_TANH, WN, ROWS>(p, slab, m_wave, n_wave, lane); break;
            case ACT_GELU_ERF: epi_generic_rows<ACT_GELU_ERF, WN, ROWS>(p, slab, m_wave, n_wave, lane); break;
            case ACT_MISH: epi_generic_rows<ACT_MISH, WN, ROWS>(p, slab, m_wave, n_wave, lane); break;
            case ACT_SILU: epi_generic_rows<ACT_SILU, WN, ROWS>(p, slab, m_wave, n_wave, lane); break;
            default: epi_generic_rows<ACT_NONE, WN, ROWS>(p, slab, m_wave, n_wave, lane); break;
        }
    } else {
        epi_qk_rows<WN, ROWS>(p, slab, m_wave, n_wave, lane);
    }
    EPI_STAMP(2);
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); EPI_STAMP(3); }   // drain of the outstanding stores
#undef EPI_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------
// Eight-wave epilogue of the warp-specialised kernel (gemm3.h): the four consumer waves stage their 64 x 64 accumulators in
// LDS as before, then the four producer waves -- idle once the k-loop is over -- take the lower 32 rows of each slab, so the
// row phase (bias / activation / split / residual, the VALU-bound part) runs on two waves per SIMD instead of one.
// Barrier protocol (every wave of the workgroup, in this order): A = k-loop stages dead, B = slabs complete.  The V block of the
// QKV projection is stored transposed straight from the accumulators by the consumers alone (n_blk is workgroup-uniform).
template <int EPI, int WN>
F5_DEVICE void gemm_epilogue_rows32(const GemmArgs& p, const float* slab_half, int m_base, int n_wave, int lane) {
    if (EPI == EPI_GENERIC) {
        switch (p.act) {
            case ACT_GELU_TANH: epi_generic_rows<ACT_GELU_TANH, WN, 32>(p, slab_half, m_base, n_wave, lane); break;
            case ACT_GELU_ERF: epi_generic_rows<ACT_GELU_ERF, WN, 32>(p, slab_half, m_base, n_wave, lane); break;
            case ACT_MISH: epi_generic_rows<ACT_MISH, WN, 32>(p, slab_half, m_base, n_wave, lane); break;
            case ACT_SILU: epi_generic_rows<ACT_SILU, WN, 32>(p, slab_half, m_base, n_wave, lane); break;
            default: epi_generic_rows<ACT_NONE, WN, 32>(p, slab_half, m_base, n_wave, lane); break;
        }
    } else {
        epi_qk_rows<WN, 32>(p, slab_half, m_base, n_wave, lane);
    }
}

template <int EPI, int TN = 2>
F5_DEVICE void gemm_epilogue8_consumer(const GemmArgs& p, f32x16 (&acc)[2][TN], float* slab, int m_wave, int n_wave, int n_blk, int lane) {
    const int fr = lane & 31, fh = lane >> 5;
    __syncthreads();                                   // A
    if (EPI == EPI_QKV && n_blk >= 2 * p.D) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int nd = n_wave - 2 * p.D + j * 32 + fr;
                const float bv = p.bias[n_wave + j * 32 + fr];
#pragma unroll
                for (int a4 = 0; a4 < 4; a4++) {
                    float pk[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) pk[e] = acc[i][j][a4 * 4 + e] + bv;
                    store_f16x4(p.vt + (size_t)nd * p.ldvt + vt_col(m_wave + i * 32 + 8 * a4 + 4 * fh), pk);
                }
            }
        return;                                        // (no barrier B for V blocks: the producers skip it too)
    }
    constexpr int WN = 32 * TN;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) slab[(i * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh) * WN + j * 32 + fr] = acc[i][j][g];
    __syncthreads();                                   // B
    gemm_epilogue_rows32<EPI, WN>(p, slab, m_wave, n_wave, lane);
}

template <int EPI, int TN = 2>
F5_DEVICE void gemm_epilogue8_producer(const GemmArgs& p, const float* slab_of_consumer, int m_wave, int n_wave, int n_blk, int lane) {
    __syncthreads();                                   // A
    if (EPI == EPI_QKV && n_blk >= 2 * p.D) return;
    __syncthreads();                                   // B
    gemm_epilogue_rows32<EPI, 32 * TN>(p, slab_of_consumer + 32 * (32 * TN), m_wave + 32, n_wave, lane);
}
