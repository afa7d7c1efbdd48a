// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)
//
//  * A (activations) and W (nn.Linear weight layout, K contiguous) are bf16.  With NSPLIT = 2 every
//    operand is a (hi, lo) pair of bf16 planes (x ~= hi + lo, ~16 mantissa bits) and each k-step issues
//    hi*hi + hi*lo + lo*hi (3 MFMAs) -- "bf16x3".  tools/precision_ladder.py shows plain bf16 misses the
//    1e-3 mel RMS bound of the 32-step CFG loop by ~8x while bf16x3 meets it with >5x margin.
//  * Tile: BM = 128 x BN in {128, 64} x BK = 32, 256 threads = 4 waves (2x2 or 4x1), v_mfma_f32_32x32x16_bf16,
//    fp32 accumulators.  LDS rows are 64 B; the 16-byte chunk index is XOR-swizzled with (row >> 2) & 3 so the
//    ds_read_b128 fragment reads and the ds_write_b128 staging writes are bank-conflict free.
//  * Pipeline: global -> registers (issued before the MFMAs of the current tile) -> LDS (after them), two LDS
//    stages, one barrier per k-tile, two workgroups per CU.
//  * CONV = true turns the A operand into a shifted window (implicit GEMM for Conv1d over the frame axis,
//    grouped or dense): k-tile kt reads rows (m + tap - center) with zero fill outside the row's sequence.
//  * Epilogues: GENERIC (bias, activation, per-column multiplier, fp32 residual, fp32 and/or split-bf16
//    outputs) and QKV (bias, rotary embedding on head 0, q pre-scaled by 1/8, V written transposed).
#pragma once
#include "common.h"

#include "gemm_epilogue.h"

F5_DEVICE int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }


// ABL (diagnostics only, f5hip_debug_gemm_bench): 0 = normal, 1 = no global loads inside the k-loop, 2 = no LDS reads / MFMAs
template <int NSPLIT, int BN, bool CONV, int EPI, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_kernel(const GemmArgs p) {
    constexpr int NPL = NSPLIT == 2 ? 2 : 1;   // NSPLIT = operand precision: 1 bf16, 2 split bf16 (3 MFMAs), 3 fp16 (PREC_F16)
    constexpr bool F16 = NSPLIT == 3;
    constexpr int BM = 128;
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
    constexpr int TM = BM / WAVES_M / 32, TN = 2;
    constexpr int A_RPT = BM / 64, B_RPT = BN / 64;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lrow = tid >> 2, lchunk = tid & 3;
    const int nk = p.K >> 5;

    int sstart[A_RPT] = {0}, send[A_RPT] = {0};
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < A_RPT; i++) {
            const int mr = m0 + lrow + 64 * i;
            if (p.row_seq_start) {
                sstart[i] = p.row_seq_start[mr];
                send[i] = p.row_seq_end[mr];
            } else {
                sstart[i] = (mr / p.seq_pitch) * p.seq_pitch;
                send[i] = sstart[i] + p.seq_valid;
            }
        }
    }
    const int a_col0 = CONV ? (int)blockIdx.x * p.conv_group_cols : 0;

    // register staging of the next k-tile (kept in named registers: plain arrays + fully unrolled static indexing)
    u32x4 ra[NPL * A_RPT], rb[NPL * B_RPT];
    const __bf16* a_ptr[NPL];
    const __bf16* w_ptr[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) {
        a_ptr[pl] = p.A[pl] + (size_t)(m0 + lrow) * p.lda + lchunk * 8 + a_col0;
        w_ptr[pl] = p.W[pl] + (size_t)(n0 + lrow) * p.ldw + lchunk * 8;
    }
    const size_t a_row64 = (size_t)64 * p.lda, w_row64 = (size_t)64 * p.ldw;

#define LOAD_TILES(KT)                                                                                         \
    {                                                                                                          \
        const int kt_ = (KT);                                                                                  \
        int a_off_, shift_ = 0;                                                                                \
        if constexpr (CONV) {                                                                                  \
            const int tap_ = kt_ / p.conv_kpt;                                                                 \
            a_off_ = (kt_ - tap_ * p.conv_kpt) * 32;                                                           \
            shift_ = (tap_ - p.conv_center) * (p.conv_dil > 1 ? p.conv_dil : 1);                                                                     \
        } else {                                                                                               \
            a_off_ = kt_ * 32;                                                                                 \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < A_RPT; i++) {                                                    \
            _Pragma("unroll") for (int pl = 0; pl < NPL; pl++) {                                            \
                if constexpr (CONV) {                                                                          \
                    const int src_ = m0 + lrow + 64 * i + shift_;                                              \
                    const bool ok_ = src_ >= sstart[i] && src_ < send[i];                                      \
                    u32x4 v_ = {0u, 0u, 0u, 0u};                                                         \
                    if (ok_) v_ = *reinterpret_cast<const u32x4*>(a_ptr[pl] + (ptrdiff_t)shift_ * p.lda + i * a_row64 + a_off_); \
                    ra[pl * A_RPT + i] = v_;                                                                   \
                } else {                                                                                       \
                    ra[pl * A_RPT + i] = *reinterpret_cast<const u32x4*>(a_ptr[pl] + i * a_row64 + a_off_);    \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < B_RPT; i++) {                                                    \
            _Pragma("unroll") for (int pl = 0; pl < NPL; pl++)                                              \
                rb[pl * B_RPT + i] = *reinterpret_cast<const u32x4*>(w_ptr[pl] + i * w_row64 + kt_ * 32);      \
        }                                                                                                      \
    }

#define STORE_TILES(STG)                                                                                       \
    {                                                                                                          \
        char* base_ = smem + (STG) * STAGE;                                                                    \
        _Pragma("unroll") for (int pl = 0; pl < NPL; pl++) {                                                \
            _Pragma("unroll") for (int i = 0; i < A_RPT; i++)                                                  \
                *reinterpret_cast<u32x4*>(base_ + pl * A_PLANE + lds_off(lrow + 64 * i, lchunk)) = ra[pl * A_RPT + i]; \
            _Pragma("unroll") for (int i = 0; i < B_RPT; i++)                                                  \
                *reinterpret_cast<u32x4*>(base_ + NPL * A_PLANE + pl * B_PLANE + lds_off(lrow + 64 * i, lchunk)) = rb[pl * B_RPT + i]; \
        }                                                                                                      \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

    // ABL == 3 (diagnostics): s_memtime stamps of workgroup (0,0) wave 0 accumulated per phase into p.stamps[0..7]
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0, st_begin = 0;
#define STAMP(IDX)                                                                                   \
    if constexpr (ABL == 3) {                                                                        \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if ((IDX) >= 0) st_acc[(IDX) < 0 ? 0 : (IDX)] += t_ - st_prev; else st_begin = t_;           \
        st_prev = t_;                                                                                \
    }
    STAMP(-1);
    LOAD_TILES(0);
    STORE_TILES(0);
    __syncthreads();
    STAMP(0);   // prologue

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const bool more = kt + 1 < nk;
        if (more && ABL != 1) LOAD_TILES(kt + 1);
        STAMP(1);   // global load issue
        const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < (ABL == 2 ? 0 : 2); s++) {
            bf16x8 af[NPL][TM], bf[NPL][TN];
            const int chunk = s * 2 + fh;
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
                for (int i = 0; i < TM; i++)
                    af[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * A_PLANE + lds_off(wm * (TM * 32) + i * 32 + fr, chunk));
#pragma unroll
                for (int j = 0; j < TN; j++)
                    bf[pl][j] = *reinterpret_cast<const bf16x8*>(base + NPL * A_PLANE + pl * B_PLANE + lds_off(wn * 64 + j * 32 + fr, chunk));
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if (NSPLIT == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = mfma_32x32x16<F16>(af[0][i], bf[0][j], acc[i][j]);
                }
        }
        STAMP(2);   // LDS reads + MFMAs
        if (more && ABL != 1) STORE_TILES((kt + 1) & 1);
        STAMP(3);   // wait for the global loads + LDS writes
        __syncthreads();
        STAMP(4);   // barrier
    }
#undef LOAD_TILES
#undef STORE_TILES

    // ---------------------------------------------------------------- epilogue (gemm_epilogue.h)
    unsigned long long epi_dbg[4] = {0, 0, 0, 0};
    gemm_epilogue<EPI, TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * (TM * TN * 1024), m0 + wm * (TM * 32), n0 + wn * 64, n0, lane,
                               ABL == 3 ? epi_dbg : nullptr);
    STAMP(5);   // epilogue
    if constexpr (ABL == 3) {
        if (p.stamps && blockIdx.x == p.stamp_bx && blockIdx.y == p.stamp_by && tid == 0) {
            for (int i = 0; i < 6; i++) p.stamps[i] = st_acc[i];
            p.stamps[6] = st_prev - st_begin;
            for (int i = 0; i < 4; i++) p.stamps[7 + i] = epi_dbg[i];
        }
    }
#undef STAMP
}

template <int NSPLIT, int BN, bool CONV, int EPI, int ABL = 0>
static hipError_t launch_gemm_t(const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    constexpr int EPI_LDS = (128 / (4 / (BN / 64)) / 32) * 2 * 4096 * 4;   // 4 waves x (TM x TN) x 4 KiB epilogue slabs
    constexpr int NPL = NSPLIT == 2 ? 2 : 1;
    constexpr int LDS = 2 * NPL * (128 + BN) * 64 < EPI_LDS ? EPI_LDS : 2 * NPL * (128 + BN) * 64;
    static unsigned attr_mask = 0;
    if (hipError_t e = f5_set_lds_attr(reinterpret_cast<const void*>(&gemm_kernel<NSPLIT, BN, CONV, EPI, ABL>), LDS, attr_mask); e != hipSuccess) return e;
    dim3 grid(n_pad / BN, m_pad / 128);
    hipLaunchKernelGGL((gemm_kernel<NSPLIT, BN, CONV, EPI, ABL>), grid, dim3(256), LDS, st, a);
    return hipGetLastError();
}
