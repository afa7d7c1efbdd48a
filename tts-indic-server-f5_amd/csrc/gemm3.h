// gemm3: warp-specialised MFMA GEMM for gfx950 (128 x 128 x 32 tiles, 512-thread workgroups) -- the production GEMM of the path.
// Operand precision PREC (template NSPLIT): 1 = bf16, 2 = split bf16 (hi + lo planes, 3 MFMAs per k-step), 3 = fp16 (one plane).
// One plane: 4 x 16 KiB ring, two workgroups per CU; split bf16: 4 x 32 KiB ring, one workgroup per CU.
//
// In-kernel s_memtime stamps of gemm.h (tools/gemm_stamps.py, profiles/) show where a k-step of a lone workgroup goes:
// 543 cycles ISSUING its 8 global loads per wave (the CU's texture-address path moves ~60 B/clk of these 64-byte row
// segments), 940 cycles LDS reads + 24 MFMAs, 640 cycles waiting for the loads and writing LDS, 190 at the barrier --
// 2300 cycles for 768 cycles of MFMA.  The load issue and the MFMAs are serial inside every wave.  Here they are on
// different waves:
//   waves 4-7 (producers): LDS-DMA (global_load_lds_dwordx4) of 8 one-KiB pieces per k-tile each, 3 tiles ahead in a
//                          4 x 32 KiB ring, counted s_waitcnt vmcnt so only the oldest tile is retired per step;
//   waves 0-3 (consumers): ds_read_b128 fragments + v_mfma_f32_32x32x16_{bf16,f16} only (64 x 64 per wave);
//   epilogue: all eight waves (gemm_epilogue.h: the producers take half of every consumer's staged tile).
// One raw s_barrier per k-tile joins both groups: "tile kt has landed" for the consumers, "tile kt-1 is consumed" for
// the producers, which then refill that stage.  Same LDS image / swizzle / epilogue as gemm2.h.
#pragma once
#include "gemm_epilogue.h"

// ABL (diagnostics only): 0 = normal, 1 = producers issue no DMA (consumers read whatever is in LDS), 2 = consumers skip the MFMAs,
// 4 = consumers skip the MFMAs AND the producers re-fetch k-tile 0 every step (cache-hot addresses: the DMA issue rate alone)
// BN = 256 ("wide"): each consumer wave owns 64 x 128 (4 accumulator tiles across): 25 % fewer operand bytes per FLOP and 16 MFMAs
// between barriers, for GEMMs whose 128 x 256 tile count fits one round on the CUs (FF1 at one utterance per GPU: 176 tiles).
template <int NSPLIT, int EPI, int ABL = 0, int BN = 128>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm3_kernel(const GemmArgs p) {
    constexpr int NPL = NSPLIT == 2 ? 2 : 1;   // NSPLIT = operand precision: 1 bf16, 2 split bf16 (3 MFMAs), 3 fp16
    constexpr bool F16 = NSPLIT == 3;
    constexpr int BM = 128, TM = 2, TN = BN / 64, NST = 4;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, STAGE = NPL * (A_PLANE + B_PLANE);
    constexpr int P = STAGE / 1024 / 4;   // DMA pieces per producer wave per k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nk = p.K >> 5;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4;
        const char* gsrc[P];
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int off = (pw * P + j) * 1024;
            const bool isA = off < NPL * A_PLANE;
            const int rel = isA ? off : off - NPL * A_PLANE;
            const int plane_bytes = isA ? A_PLANE : B_PLANE;
            const int pl = rel / plane_bytes;
            const int row = (rel - pl * plane_bytes) / 64 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            // explicit selects: a run-time index into the by-value argument struct makes hipcc copy all of GemmArgs to scratch
            const __bf16* ap = (NPL == 2 && pl) ? p.A[1] : p.A[0];
            const __bf16* wp = (NPL == 2 && pl) ? p.W[1] : p.W[0];
            const __bf16* base = isA ? ap + (size_t)(m0 + row) * p.lda : wp + (size_t)(n0 + row) * p.ldw;
            gsrc[j] = reinterpret_cast<const char*>(base + chunk * 8);
        }
        auto issue_tile = [&](int kt) {
            char* dst = smem + (kt % NST) * STAGE + pw * (P * 1024);
#pragma unroll
            for (int j = 0; j < P; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (size_t)(ABL == 4 ? 0 : kt) * 64),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < NST - 1; t++)
            if (t < nk && ABL != 1) issue_tile(t);
        for (int kt = 0; kt < nk; kt++) {
            const int newer = min(NST - 2, nk - 1 - kt);
            if (newer >= 2) wait_vmcnt<2 * P>();
            else if (newer == 1) wait_vmcnt<P>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NST - 1 < nk && ABL != 1) issue_tile(kt + NST - 1);
        }
        if constexpr (ABL == 0) {
            // the producers are idle now: they take the lower halves of the consumers' epilogue slabs (gemm_epilogue.h)
            const int cw = pw, cwm = cw >> 1, cwn = cw & 1;
            gemm_epilogue8_producer<EPI, TN>(p, reinterpret_cast<const float*>(smem) + cw * (64 * 32 * TN), m0 + cwm * 64, n0 + cwn * (32 * TN), n0, lane);
        } else {
            __syncthreads();   // pairs with the workgroup barrier at the top of gemm_epilogue (consumers)
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

    // Software-pipelined fragments: the LDS reads of the NEXT half k-step (16 k) are always issued before the 12 MFMAs of the
    // current one, so with one consumer wave per SIMD the LDS latency hides behind 384 MFMA cycles.
    bf16x8 fa[2][NPL][TM], fb[2][NPL][TN];
    auto read_frags = [&](int buf, const char* st, int s) {
        const int chunk = s * 2 + fh;
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
            for (int i = 0; i < TM; i++)
                fa[buf][pl][i] = *reinterpret_cast<const bf16x8*>(st + pl * A_PLANE + lds_off2(wm * 64 + i * 32 + fr, chunk));
#pragma unroll
            for (int j = 0; j < TN; j++)
                fb[buf][pl][j] = *reinterpret_cast<const bf16x8*>(st + NPL * A_PLANE + pl * B_PLANE + lds_off2(wn * (32 * TN) + j * 32 + fr, chunk));
        }
    };
    auto mfma_frags = [&](int buf) {
        if constexpr (ABL == 2 || ABL == 4) {
            asm volatile("" :: "v"(fa[buf][0][0]), "v"(fb[buf][0][0]), "v"(fa[buf][NPL - 1][TM - 1]), "v"(fb[buf][NPL - 1][TN - 1]));
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                if (NSPLIT == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][1][i], fb[buf][0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][0][i], fb[buf][1][j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = mfma_32x32x16<F16>(fa[buf][0][i], fb[buf][0][j], acc[i][j]);
            }
    };
    unsigned long long ts[4] = {0, 0, 0, 0};
#define G3_STAMP(I) if constexpr (ABL == 3) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[I])::"memory"); __builtin_amdgcn_sched_barrier(0); }
    G3_STAMP(0);
    __builtin_amdgcn_s_barrier();                         // B_0: tile 0 landed
    G3_STAMP(1);
    read_frags(0, smem, 0);
    for (int kt = 0; kt < nk; kt++) {
        const char* st = smem + (kt % NST) * STAGE;
        read_frags(1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_frags(0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            // every LDS read of tile kt has returned (they were issued 12 MFMAs ago): the producers may refill its stage
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // B_{kt+1}: tile kt+1 landed
            read_frags(0, smem + ((kt + 1) % NST) * STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_frags(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    G3_STAMP(2);
    unsigned long long epi_dbg[4] = {0, 0, 0, 0};
    if constexpr (ABL == 0) gemm_epilogue8_consumer<EPI, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * (64 * 32 * TN), m0 + wm * 64, n0 + wn * (32 * TN), n0, lane);
    else gemm_epilogue<EPI, TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * (TM * TN * 1024), m0 + wm * 64, n0 + wn * (32 * TN), n0, lane, ABL == 3 ? epi_dbg : nullptr);
    G3_STAMP(3);
#undef G3_STAMP
    if constexpr (ABL == 3) {
        if (p.stamps && blockIdx.x == p.stamp_bx && blockIdx.y == p.stamp_by && tid == 0) {
            p.stamps[0] = ts[1] - ts[0]; p.stamps[1] = ts[2] - ts[1]; p.stamps[2] = ts[3] - ts[2]; p.stamps[3] = ts[0];
            for (int i = 0; i < 4; i++) p.stamps[7 + i] = epi_dbg[i];
        }
    }
}

template <int NSPLIT, int EPI, int ABL = 0, int BN = 128>
static hipError_t launch_gemm3_t(const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    constexpr int NPL = NSPLIT == 2 ? 2 : 1;
    constexpr int RING = 4 * NPL * (128 + BN) * 64, SLABS = 4 * 64 * (BN / 2) * 4;   // k-loop ring; four 64 x BN/2 fp32 epilogue slabs
    constexpr int LDS = RING > SLABS ? RING : SLABS;
    static_assert(LDS <= 160 * 1024, "tile does not fit the LDS");
    static unsigned attr_mask = 0;
    if (hipError_t e = f5_set_lds_attr(reinterpret_cast<const void*>(&gemm3_kernel<NSPLIT, EPI, ABL, BN>), LDS, attr_mask); e != hipSuccess) return e;
    dim3 grid(n_pad / BN, m_pad / 128);
    hipLaunchKernelGGL((gemm3_kernel<NSPLIT, EPI, ABL, BN>), grid, dim3(512), LDS, st, a);
    return hipGetLastError();
}
