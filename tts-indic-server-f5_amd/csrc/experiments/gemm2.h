// gemm2: LDS-DMA ring version of the split-bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  for gfx950.
//
// Why a second kernel: at one utterance per GPU the GEMMs have only 176-528 tiles of 128 x 128 for 256 CUs and the
// register-staged kernel (gemm.h, 4 waves, 2 workgroups / CU) is bound by the ~1.1-1.4 us global->LDS latency with one
// k-tile in flight (tools/gemm_microbench.py).  Here ONE 512-thread workgroup (8 waves, 2 per SIMD) owns a CU:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB pieces, no VGPR staging, no ds_write), NST-deep ring
//     (4 x 32 KiB for 128 x 128, 3 x 48 KiB for 256 x 128) => up to 96 KiB in flight per CU;
//   * counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per k-tile: the wait retires only the oldest tile, the
//     newer ones stay in flight across the barrier (guide: "Pipelining across barriers");
//   * the XOR swizzle of the 64-byte LDS rows is applied on the per-lane SOURCE address (the DMA destination is
//     lane-linear) and on the ds_read_b128 fragment reads (guide rule 21);
//   * 8 waves split the tile 2 x 4 (64 x 32 per wave) or 4 x 2 (64 x 64 per wave); epilogue shared with gemm.h.
// Restrictions: no implicit-conv operand (those GEMMs stay on gemm.h), K % 32 == 0, M_pad % BM == 0, N_pad % BN == 0.
#pragma once
#include "../gemm_epilogue.h"

template <int NSPLIT, int BM, int BN>
struct Gemm2Cfg {
    static constexpr int WAVES_N = (BM == 128) ? 4 : 2;
    static constexpr int WAVES_M = 8 / WAVES_N;
    static constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    static constexpr int STAGE = NSPLIT * (A_PLANE + B_PLANE);
    static constexpr int NST = (4 * STAGE <= 144 * 1024) ? 4 : 3;
    static constexpr int PIECES = STAGE / 1024;   // 1 KiB LDS-DMA pieces per stage
    static constexpr int P = PIECES / 8;          // per wave
    static constexpr int LDS = NST * STAGE > 8 * TM * TN * 4096 ? NST * STAGE : 8 * TM * TN * 4096;   // ring, and room for the epilogue slabs
    static_assert(PIECES % 8 == 0, "pieces must divide over 8 waves");
    static_assert(LDS >= 8 * TM * TN * 4096, "ring must hold the epilogue slabs");
};

template <int NSPLIT, int BM, int BN, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm2_kernel(const GemmArgs p) {
    using C = Gemm2Cfg<NSPLIT, BM, BN>;
    constexpr int TM = C::TM, TN = C::TN, NST = C::NST, P = C::P, STAGE = C::STAGE, A_PLANE = C::A_PLANE, B_PLANE = C::B_PLANE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nk = p.K >> 5;

    // ---- per-lane DMA source pointers (k-tile 0) for this wave's P pieces; destination = piece * 1 KiB + lane * 16
    const char* gsrc[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const int off = (wave * P + j) * 1024;                       // byte offset of the piece inside a stage
        const bool isA = off < NSPLIT * A_PLANE;
        const int rel = isA ? off : off - NSPLIT * A_PLANE;
        const int plane_bytes = isA ? A_PLANE : B_PLANE;
        const int pl = rel / plane_bytes;
        const int row = (rel - pl * plane_bytes) / 64 + (lane >> 2);   // row inside the plane (piece = 16 rows x 64 B)
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);               // logical 16-B chunk that lands in physical slot lane & 3
        // explicit selects: a run-time index into the by-value argument struct makes hipcc copy all of GemmArgs to scratch
        const __bf16* ap = (NSPLIT == 2 && pl) ? p.A[1] : p.A[0];
        const __bf16* wp = (NSPLIT == 2 && pl) ? p.W[1] : p.W[0];
        const __bf16* base = isA ? ap + (size_t)(m0 + row) * p.lda : wp + (size_t)(n0 + row) * p.ldw;
        gsrc[j] = reinterpret_cast<const char*>(base + chunk * 8);
    }
    auto issue_tile = [&](int kt) {
        char* dst = smem + (kt % NST) * STAGE + wave * (P * 1024);
#pragma unroll
        for (int j = 0; j < P; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (size_t)kt * 64),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

#pragma unroll
    for (int t = 0; t < NST - 1; t++)
        if (t < nk) issue_tile(t);

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        // tiles kt+1 .. kt+NST-2 (if they exist) may stay in flight; tile kt must have landed
        const int newer = min(NST - 2, nk - 1 - kt);
        if (newer >= 2) wait_vmcnt<2 * P>();
        else if (newer == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // tile kt visible to all waves; everyone is done reading tile kt-1's stage
        if (kt + NST - 1 < nk) issue_tile(kt + NST - 1);
        const char* st = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            bf16x8 af[NSPLIT][TM], bf[NSPLIT][TN];
            const int chunk = s * 2 + fh;
#pragma unroll
            for (int pl = 0; pl < NSPLIT; pl++) {
#pragma unroll
                for (int i = 0; i < TM; i++)
                    af[pl][i] = *reinterpret_cast<const bf16x8*>(st + pl * A_PLANE + lds_off2(wm * (TM * 32) + i * 32 + fr, chunk));
#pragma unroll
                for (int j = 0; j < TN; j++)
                    bf[pl][j] = *reinterpret_cast<const bf16x8*>(st + NSPLIT * A_PLANE + pl * B_PLANE + lds_off2(wn * (TN * 32) + j * 32 + fr, chunk));
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
    }
    gemm_epilogue<EPI, TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * (TM * TN * 1024), m0 + wm * (TM * 32), n0 + wn * (TN * 32), n0,
                               lane);
}

template <int NSPLIT, int BM, int BN, int EPI>
static hipError_t launch_gemm2_t(const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    using C = Gemm2Cfg<NSPLIT, BM, BN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<NSPLIT, BM, BN, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(n_pad / BN, m_pad / BM);
    hipLaunchKernelGGL((gemm2_kernel<NSPLIT, BM, BN, EPI>), grid, dim3(512), C::LDS, st, a);
    return hipGetLastError();
}
