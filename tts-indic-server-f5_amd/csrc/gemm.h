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

enum { EPI_GENERIC = 0, EPI_QKV = 1 };

struct GemmArgs {
    const __bf16* A[2];
    int lda;
    const __bf16* W[2];
    int M, N, K;
    // implicit-GEMM conv
    int conv_kpt;         // k-tiles (of 32 channels) per tap
    int conv_center;      // (kernel_size - 1) / 2
    int conv_group_cols;  // A column base = blockIdx.x * conv_group_cols (grouped conv with BN == group width)
    const int* row_seq_start;
    const int* row_seq_end;
    int group_w;          // > 0: N is laid out as groups padded to 64 columns; real column = (n/64)*group_w + n%64
    // generic epilogue
    const float* bias;
    const int* row_keep;  // optional [M]: rows with 0 contribute nothing (reference zeroes masked attention rows)
    int act;
    const float* mul;
    const float* res;
    int ldres;
    float* out_f32;
    int ldo;
    __bf16* out_hi;
    __bf16* out_lo;
    int ldob;
    // QKV epilogue
    int D;                   // model dim (N == 3 D)
    const int* row_pos;      // [M_pad] frame index inside the row's sequence
    const float* rope_cos;   // [max_pos][32]
    const float* rope_sin;
    __bf16* qk;              // [M_pad][2 D]
    __bf16* vt;              // [D][ldvt]
    int ldvt;
};

F5_DEVICE int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int NSPLIT, int BN, bool CONV, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_kernel(const GemmArgs p) {
    constexpr int BM = 128;
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
    constexpr int TM = BM / WAVES_M / 32, TN = 2;
    constexpr int A_RPT = BM / 64, B_RPT = BN / 64;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = NSPLIT * (A_PLANE + B_PLANE);
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
            sstart[i] = p.row_seq_start[m0 + lrow + 64 * i];
            send[i] = p.row_seq_end[m0 + lrow + 64 * i];
        }
    }
    const int a_col0 = CONV ? (int)blockIdx.x * p.conv_group_cols : 0;

    // register staging of the next k-tile (kept in named registers: plain arrays + fully unrolled static indexing)
    u32x4 ra[NSPLIT * A_RPT], rb[NSPLIT * B_RPT];
    const __bf16* a_ptr[NSPLIT];
    const __bf16* w_ptr[NSPLIT];
#pragma unroll
    for (int pl = 0; pl < NSPLIT; pl++) {
        a_ptr[pl] = p.A[pl] + (size_t)(m0 + lrow) * p.lda + lchunk * 8 + a_col0;
        w_ptr[pl] = p.W[pl] + (size_t)(n0 + lrow) * p.K + lchunk * 8;
    }
    const size_t a_row64 = (size_t)64 * p.lda, w_row64 = (size_t)64 * p.K;

#define LOAD_TILES(KT)                                                                                         \
    {                                                                                                          \
        const int kt_ = (KT);                                                                                  \
        int a_off_, shift_ = 0;                                                                                \
        if constexpr (CONV) {                                                                                  \
            const int tap_ = kt_ / p.conv_kpt;                                                                 \
            a_off_ = (kt_ - tap_ * p.conv_kpt) * 32;                                                           \
            shift_ = tap_ - p.conv_center;                                                                     \
        } else {                                                                                               \
            a_off_ = kt_ * 32;                                                                                 \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < A_RPT; i++) {                                                    \
            _Pragma("unroll") for (int pl = 0; pl < NSPLIT; pl++) {                                            \
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
            _Pragma("unroll") for (int pl = 0; pl < NSPLIT; pl++)                                              \
                rb[pl * B_RPT + i] = *reinterpret_cast<const u32x4*>(w_ptr[pl] + i * w_row64 + kt_ * 32);      \
        }                                                                                                      \
    }

#define STORE_TILES(STG)                                                                                       \
    {                                                                                                          \
        char* base_ = smem + (STG) * STAGE;                                                                    \
        _Pragma("unroll") for (int pl = 0; pl < NSPLIT; pl++) {                                                \
            _Pragma("unroll") for (int i = 0; i < A_RPT; i++)                                                  \
                *reinterpret_cast<u32x4*>(base_ + pl * A_PLANE + lds_off(lrow + 64 * i, lchunk)) = ra[pl * A_RPT + i]; \
            _Pragma("unroll") for (int i = 0; i < B_RPT; i++)                                                  \
                *reinterpret_cast<u32x4*>(base_ + NSPLIT * A_PLANE + pl * B_PLANE + lds_off(lrow + 64 * i, lchunk)) = rb[pl * B_RPT + i]; \
        }                                                                                                      \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

    LOAD_TILES(0);
    STORE_TILES(0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const bool more = kt + 1 < nk;
        if (more) LOAD_TILES(kt + 1);
        const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bf16x8 af[NSPLIT][TM], bf[NSPLIT][TN];
            const int chunk = s * 2 + fh;
#pragma unroll
            for (int pl = 0; pl < NSPLIT; pl++) {
#pragma unroll
                for (int i = 0; i < TM; i++)
                    af[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * A_PLANE + lds_off(wm * (TM * 32) + i * 32 + fr, chunk));
#pragma unroll
                for (int j = 0; j < TN; j++)
                    bf[pl][j] = *reinterpret_cast<const bf16x8*>(base + NSPLIT * A_PLANE + pl * B_PLANE + lds_off(wn * 64 + j * 32 + fr, chunk));
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if (NSPLIT == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) STORE_TILES((kt + 1) & 1);
        __syncthreads();
    }
#undef LOAD_TILES
#undef STORE_TILES

    // ---------------------------------------------------------------- epilogue
    // acc[i][j][g] = C[m][n], m = m0 + wm*TM*32 + i*32 + (g&3) + 8*(g>>2) + 4*fh, n = n0 + wn*64 + j*32 + fr
    if (EPI == EPI_GENERIC) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * 64 + j * 32 + fr;   // column in the (possibly group-padded) weight layout
            int no = n;                                  // column in the output / residual / multiplier
            bool nok = n < p.N;
            if (p.group_w) {
                nok = nok && (n & 63) < p.group_w;
                no = (n >> 6) * p.group_w + (n & 63);
            }
            const float bv = (p.bias && nok) ? p.bias[n] : 0.0f;
            const float mv = (p.mul && nok) ? p.mul[no] : 1.0f;
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh;
                    if (nok && m < p.M) {
                        float v = apply_act(acc[i][j][g] + bv, p.act);
                        if (p.row_keep && !p.row_keep[m]) v = 0.0f;
                        v *= mv;
                        if (p.res) v += p.res[(size_t)m * p.ldres + no];
                        if (p.out_f32) p.out_f32[(size_t)m * p.ldo + no] = v;
                        if (p.out_hi) {
                            __bf16 hi, lo;
                            split_bf16(v, hi, lo);
                            p.out_hi[(size_t)m * p.ldob + no] = hi;
                            if (p.out_lo) p.out_lo[(size_t)m * p.ldob + no] = lo;
                        }
                    }
                }
            }
        }
    } else {
        const int D = p.D;
        const int which = n0 / D;             // 0 q, 1 k, 2 v (uniform per workgroup: D % BN == 0)
        const int nd0 = n0 - which * D;       // column offset inside the q/k/v block
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nd = nd0 + wn * 64 + j * 32 + fr;
            const float bv = p.bias[n0 + wn * 64 + j * 32 + fr];
            const bool rot = which < 2 && nd < 64;   // head 0 only (rotary applied before the head split)
#pragma unroll
            for (int i = 0; i < TM; i++) {
                if (which < 2) {
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        const int m = m0 + wm * (TM * 32) + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh;
                        float v = acc[i][j][g] + bv;
                        if (rot) {   // wave-uniform: nd < 64 for a whole 32-column fragment
                            const float other = __shfl_xor(v, 1, 64);
                            const int pos = p.row_pos[m];
                            const float c = p.rope_cos[pos * 32 + (nd >> 1)], sn = p.rope_sin[pos * 32 + (nd >> 1)];
                            v = (nd & 1) ? (v * c + other * sn) : (v * c - other * sn);
                        }
                        if (which == 0) v *= 0.125f;   // softmax scale 1/sqrt(64), exact in bf16
                        if (m < p.M) p.qk[(size_t)m * (2 * D) + which * D + nd] = (__bf16)v;
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const int mb = m0 + wm * (TM * 32) + i * 32 + 8 * a + 4 * fh;
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; e++) pk[e] = (__bf16)(acc[i][j][a * 4 + e] + bv);
                        *reinterpret_cast<bf16x4*>(p.vt + (size_t)nd * p.ldvt + mb) = pk;
                    }
                }
            }
        }
    }
}

template <int NSPLIT, int BN, bool CONV, int EPI>
static hipError_t launch_gemm_t(const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    constexpr int LDS = 2 * NSPLIT * (128 + BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<NSPLIT, BN, CONV, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(n_pad / BN, m_pad / 128);
    hipLaunchKernelGGL((gemm_kernel<NSPLIT, BN, CONV, EPI>), grid, dim3(256), LDS, st, a);
    return hipGetLastError();
}
