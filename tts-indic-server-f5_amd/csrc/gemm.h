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


// Row-contiguous generic epilogue for one 32 x 64 wave sub-tile staged in LDS (fp32): each lane owns 4 columns of 8 rows.
template <int ACT>
F5_DEVICE void epi_generic_rows(const GemmArgs& p, const float* stg, int m_base, int n, int c4, int r0) {
    int no = n;              // column in the output / residual / multiplier
    bool nok = n < ((p.N + 3) & ~3);
    if (p.group_w) {
        nok = nok && (n & 63) < p.group_w;
        no = (n >> 6) * p.group_w + (n & 63);
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, mv = {1.f, 1.f, 1.f, 1.f};
    if (p.bias && nok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.mul && nok) mv = *reinterpret_cast<const f32x4*>(p.mul + no);
    f32x4 rs[8];
    int keep[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int m = m_base + q * 4 + r0;
        rs[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        keep[q] = 1;
        if (p.res && nok && m < p.M) rs[q] = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldres + no);
        if (p.row_keep) keep[q] = p.row_keep[m];
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int m = m_base + q * 4 + r0;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + (q * 4 + r0) * 64 + c4) + bv;
        if (ACT != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = apply_act(v[e], ACT);
        }
        if (!keep[q]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        v = v * mv + rs[q];
        if (nok && m < p.M) {
            if (p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.ldo + no) = v;
            if (p.out_hi) {
                bf16x4 hi, lo;
                const float vv[4] = {v[0], v[1], v[2], v[3]};
                split_bf16x4(vv, hi, lo);
                *reinterpret_cast<bf16x4*>(p.out_hi + (size_t)m * p.ldob + no) = hi;
                if (p.out_lo) *reinterpret_cast<bf16x4*>(p.out_lo + (size_t)m * p.ldob + no) = lo;
            }
        }
    }
}

// ABL (diagnostics only, f5hip_debug_gemm_bench): 0 = normal, 1 = no global loads inside the k-loop, 2 = no LDS reads / MFMAs
template <int NSPLIT, int BN, bool CONV, int EPI, int ABL = 0>
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
        if (more && ABL != 1) LOAD_TILES(kt + 1);
        const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < (ABL == 2 ? 0 : 2); s++) {
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
        if (more && ABL != 1) STORE_TILES((kt + 1) & 1);
        __syncthreads();
    }
#undef LOAD_TILES
#undef STORE_TILES

    // ---------------------------------------------------------------- epilogue
    // acc[i][j][g] = C[m][n], m = m0 + wm*TM*32 + i*32 + (g&3) + 8*(g>>2) + 4*fh, n = n0 + wn*64 + j*32 + fr.
    // Each wave transposes its 32 x 64 sub-tile through a private 8 KB LDS slab so that every lane owns 4 consecutive
    // columns of a row: residual / output traffic becomes 16-byte (fp32) and 8-byte (bf16) row-contiguous accesses,
    // all residual loads are issued before any arithmetic, and the rotary pairs of the QKV epilogue are lane-local.
    float* stg = reinterpret_cast<float*>(smem) + wave * 2048;
    const int n_base = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m_base = m0 + wm * (TM * 32) + i * 32;
        if (EPI == EPI_QKV && n0 >= 2 * p.D) {
            // V block: written transposed ([feature][token]) straight from the accumulators, 4 tokens = 8 bytes per store
            const int nd0 = n0 - 2 * p.D;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int nd = nd0 + wn * 64 + j * 32 + fr;
                const float bv = p.bias[n_base + j * 32 + fr];
#pragma unroll
                for (int a4 = 0; a4 < 4; a4++) {
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; e++) pk[e] = (__bf16)(acc[i][j][a4 * 4 + e] + bv);
                    *reinterpret_cast<bf16x4*>(p.vt + (size_t)nd * p.ldvt + m_base + 8 * a4 + 4 * fh) = pk;
                }
            }
            continue;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) stg[((g & 3) + 8 * (g >> 2) + 4 * fh) * 64 + j * 32 + fr] = acc[i][j][g];
        __syncthreads();
        const int c4 = (lane & 15) * 4, r0 = lane >> 4;
        const int n = n_base + c4;   // column in the (possibly group-padded) weight layout
        if (EPI == EPI_GENERIC) {
            switch (p.act) {
                case ACT_GELU_TANH: epi_generic_rows<ACT_GELU_TANH>(p, stg, m_base, n, c4, r0); break;
                case ACT_GELU_ERF: epi_generic_rows<ACT_GELU_ERF>(p, stg, m_base, n, c4, r0); break;
                case ACT_MISH: epi_generic_rows<ACT_MISH>(p, stg, m_base, n, c4, r0); break;
                case ACT_SILU: epi_generic_rows<ACT_SILU>(p, stg, m_base, n, c4, r0); break;
                default: epi_generic_rows<ACT_NONE>(p, stg, m_base, n, c4, r0); break;
            }
        } else {
            // Q / K blocks: bias, rotary embedding on head 0 (interleaved pairs, lane-local), q * 1/8, bf16 row-major
            const int D = p.D;
            const int which = n0 / D;        // 0 q, 1 k (uniform per workgroup: D % BN == 0)
            const int nd = n - which * D;
            const bool rot = nd < 64;        // head 0 only: rotary is applied before the head split
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            int pos[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pos[q] = rot ? p.row_pos[m_base + q * 4 + r0] : 0;
            float2 cs[8], sn[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                cs[q] = make_float2(1.f, 1.f);
                sn[q] = make_float2(0.f, 0.f);
                if (rot) {
                    cs[q] = *reinterpret_cast<const float2*>(p.rope_cos + pos[q] * 32 + (nd >> 1));
                    sn[q] = *reinterpret_cast<const float2*>(p.rope_sin + pos[q] * 32 + (nd >> 1));
                }
            }
            const float qs = which == 0 ? 0.125f : 1.0f;   // softmax scale 1/sqrt(64), exact in bf16
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int m = m_base + q * 4 + r0;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (q * 4 + r0) * 64 + c4) + bv;
                bf16x4 o;
                o[0] = (__bf16)((v[0] * cs[q].x - v[1] * sn[q].x) * qs);
                o[1] = (__bf16)((v[1] * cs[q].x + v[0] * sn[q].x) * qs);
                o[2] = (__bf16)((v[2] * cs[q].y - v[3] * sn[q].y) * qs);
                o[3] = (__bf16)((v[3] * cs[q].y + v[2] * sn[q].y) * qs);
                if (m < p.M) *reinterpret_cast<bf16x4*>(p.qk + (size_t)m * (2 * D) + which * D + nd) = o;
            }
        }
    }
}

template <int NSPLIT, int BN, bool CONV, int EPI, int ABL = 0>
static hipError_t launch_gemm_t(const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    constexpr int LDS = 2 * NSPLIT * (128 + BN) * 64 < 32768 ? 32768 : 2 * NSPLIT * (128 + BN) * 64;   // >= 4 x 8 KB epilogue slabs
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<NSPLIT, BN, CONV, EPI, ABL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(n_pad / BN, m_pad / 128);
    hipLaunchKernelGGL((gemm_kernel<NSPLIT, BN, CONV, EPI, ABL>), grid, dim3(256), LDS, st, a);
    return hipGetLastError();
}
