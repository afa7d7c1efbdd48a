// gemm5: exact-fit, full-line MFMA GEMM for gfx950 -- the production kernel of the transformer-block GEMMs (one 16-bit plane).
//   C[M,N] = A[M,K] * W[N,K]^T, fp16 or bf16 operands, fp32 accumulate, epilogues of gemm_epilogue.h.
//
// Why (profiles/r02_fillrate_microbench.txt, tools/fillrate.hip): at one utterance per GPU the k-loop of these GEMMs is bound by the
// operand fill L2 -> LDS, not by the matrix pipe.  The round-1 kernel (gemm3.h: 128 x 128 tiles, 32-deep k-steps = 64-byte row
// pieces, row-major tile order) fills at 37 GB/s per CU with HBM-cold weights; the same loader with
//   * 64-deep k-steps: every LDS-DMA piece is 8 rows x 128 B = eight FULL cache lines (a 64-byte piece fetches its line twice),
//   * exact-fit tiles: (16 RB) x (16 CB) with RB x CB chosen per shape so that the grid is a whole number of rounds on the 256 CUs
//     (M_pad = 2816 at the C2 config: RB = 11 -> 16 row slabs x 16 column slabs = 256 tiles for out / FF1 / FF2 / QKV alike),
//   * XCD-blocked tile order: blocks b, b + 8, ... share an XCD (round-robin dispatch); that XCD gets a contiguous run of tiles in
//     row-major order (whole row slabs), so its 4 MiB L2 holds its A slabs and every W panel is fetched once per XCD,
// fills at 86-95 GB/s per CU: 5.7 / 7.2 / 8.8 / 10.4 us for out / FF1 / QKV / FF2 against 14 / 12 (x2 rounds) / 14 (x3) / 20.
//
// Structure: 512 threads.  Waves 4-7 (loaders) stream k-steps by LDS-DMA (global_load_lds_dwordx4, source-side XOR swizzle, counted
// s_waitcnt vmcnt) into an NST-deep ring; waves 0-3 (consumers, one per SIMD) read 16-byte fragments and issue
// v_mfma_f32_16x16x32_{f16,bf16}; one raw s_barrier per k-step joins both groups.  16 x 16 MFMA blocks make any multiple of 16 rows a
// legal tile height (176 = 11 x 16).  The consumers tile the block grid WR x (4 / WR): 4 x 1 for the 64-column tiles of out / FF2,
// 2 x 2 for 128 columns, 1 x 4 for 192.  Why the roles are separate waves, and what the consumer loop must look like, is measured
// (profiles/r02_gemm5_ablation.txt): pure loader 0.35 us per k-step, MFMAs alone 0.34, and an 8-wave kernel whose waves all did both
// took 0.59 -- the sum: a wave blocked in VMEM issue cannot issue MFMAs, and its SIMD partner is blocked at the same moment.
// Epilogue: accumulators -> one fp32 slab of the whole tile in LDS (it aliases the dead ring) -> all eight waves run the shared row
// phase on 8-row x 64-column items (16-byte row-contiguous residual loads / stores).  V blocks of the QKV projection are stored
// transposed straight from the accumulators (the 16 x 16 C layout holds 4 consecutive tokens per lane).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "gemm_epilogue.h"

template <bool F16>
F5_DEVICE f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int RB, int CB, int NST>
struct Gemm5Cfg {
    static constexpr int BM = RB * 16, BN = CB * 16;
    static constexpr int PIECES = (BM + BN) / 8;          // 1 KiB LDS-DMA pieces (8 rows x 128 B) per k-step
    static constexpr int STAGE = PIECES * 1024;
    static constexpr int SLD = BN + 4;                    // fp32 slab row stride: +4 floats keeps the transposing ds_write_b32 at 2-way (free)
    static constexpr int RING = NST * STAGE, SLAB = BM * SLD * 4;
    static constexpr int LDS = RING > SLAB ? RING : SLAB;
    static_assert(LDS <= 160 * 1024, "tile does not fit the LDS");
    static_assert(CB % 4 == 0, "column blocks must split over 64-column panels");
};

// tile index of workgroup b: XCD-blocked when the grid divides over the 8 XCDs (speed only: any bijection is correct)
F5_DEVICE int gemm5_tile_of_block(int b, int n_tiles) {
    if (n_tiles & 7) return b;
    return (b & 7) * (n_tiles >> 3) + (b >> 3);
}

// ABL (diagnostics, -DF5HIP_GEMM5_ABL builds + F5HIP_GEMM5_ABL=<n> at run time; results are garbage): 1 = no MFMAs, 2 = no fragment reads and
// no MFMAs, 3 = no LDS-DMA inside the loop, 4 = MFMAs only (no DMA, no fragment reads)
template <bool F16, int EPI, int RB, int CB, int WR, int NST, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm5_kernel(const GemmArgs p, const int tiles_n, const int n_rows_w) {
    using C = Gemm5Cfg<RB, CB, NST>;
    constexpr int BM = C::BM, BN = C::BN, PIECES = C::PIECES, STAGE = C::STAGE, SLD = C::SLD;
    constexpr int WC = 4 / WR;                                 // the 4 consumer waves tile the block grid WR (rows) x WC (columns)
    constexpr int MRB = (RB + WR - 1) / WR, MCB = CB / WC;
    static_assert(CB % WC == 0 && NST >= 3 && NST <= 4, "bad tile configuration");
    constexpr int P_HI = (PIECES + 3) / 4, P_LO = PIECES / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = gemm5_tile_of_block(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nk = p.K >> 6;
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#define G5_STAMP(I) if constexpr (ABL == 5) { if (tid == 0 || tid == 256) ts[I] = __builtin_amdgcn_s_memrealtime(); }
    G5_STAMP(0);
    // consumer geometry (also used by the epilogue): row blocks rb0 .. rb0 + nrb - 1, column blocks cb0 .. cb0 + MCB - 1
    const int cw = wave & 3;
    const int wr = cw / WC, wc = cw % WC;
    const int rb0 = (wr * RB) / WR, nrb = ((wr + 1) * RB) / WR - rb0;   // wave-uniform
    const int cb0 = wc * MCB;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[MRB][MCB];

    // Ring protocol.  All NST stages are filled up front.  Barrier B_{kt+1} sits between the two 32-deep halves of k-step kt: behind it
    // "k-step kt + 1 has landed" (every loader wave waited for its own pieces) and "every read of stage kt % NST has returned" (every
    // consumer waited lgkmcnt(0)), so the loaders issue k-step kt + NST into that stage at once: one stage being consumed, NST - 1 in flight.
    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves: they sit in VMEM issue for the whole k-loop
        // (the CU's address path takes ~26 cycles per 1 KiB piece: profiles/r02_gemm5_ablation.txt), which is why they are not the waves
        // that issue MFMAs: with all eight waves doing both, loader time and MFMA time added up instead of overlapping.
        const int pw = wave - 4;
        const int mine = (PIECES - pw + 3) >> 2;               // pieces pw, pw + 4, ... of every k-step (1 KiB = 8 rows x 128 B each)
        const char* gsrc[P_HI];
#pragma unroll
        for (int j = 0; j < P_HI; j++) {
            const int pc = pw + 4 * j;
            const int row = pc * 8 + (lane >> 3);              // row of the stage image: [0, BM) = A rows, [BM, BM + BN) = W rows
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);   // logical 16-byte chunk that lands in physical slot lane & 7
            const bool isA = row < BM;
            // rows past the matrices (partial last slabs) re-read the last valid row: finite data, never stored
            const int grow = isA ? min(m0 + row, p.M - 1) : min(n0 + row - BM, n_rows_w - 1);
            const __bf16* base = isA ? p.A[0] + (size_t)grow * p.lda : p.W[0] + (size_t)grow * p.ldw;
            gsrc[j] = reinterpret_cast<const char*>(base + chunk * 8);
        }
        auto issue_tile = [&](int kt) {
            char* dst = smem + (kt % NST) * STAGE + pw * 1024;
#pragma unroll
            for (int j = 0; j < P_HI; j++)
                if ((ABL != 3 && ABL != 4) || kt < NST)
                    if (j < P_LO || pw + 4 * j < PIECES)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (size_t)kt * 128),
                                                         (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0, 0);
        };
        // this wave's pieces of a k-step have landed when at most `newer` younger k-steps of its own are still in flight
        auto wait_landed = [&](int newer) {
            if (mine == P_HI) {
                if (newer >= 3) wait_vmcnt<3 * P_HI>(); else if (newer == 2) wait_vmcnt<2 * P_HI>(); else if (newer == 1) wait_vmcnt<P_HI>(); else wait_vmcnt<0>();
            } else {
                if (newer >= 3) wait_vmcnt<3 * P_LO>(); else if (newer == 2) wait_vmcnt<2 * P_LO>(); else if (newer == 1) wait_vmcnt<P_LO>(); else wait_vmcnt<0>();
            }
        };
#pragma unroll
        for (int t = 0; t < NST; t++)
            if (t < nk) issue_tile(t);
        wait_landed(min(NST - 1, nk - 1));
        G5_STAMP(1);
        __builtin_amdgcn_s_barrier();                          // B_0
        for (int kt = 0; kt + 1 < nk; kt++) {
            wait_landed(min(NST - 2, nk - 2 - kt));            // k-step kt + 1 (k-steps kt + 2 .. kt + NST - 1 stay in flight)
            __builtin_amdgcn_s_barrier();                      // B_{kt+1}
            if (kt + NST < nk) issue_tile(kt + NST);
        }
    } else {
        // ------------------------------------------------------------------ consumer waves, one per SIMD: LDS fragment reads + MFMAs only
        // Fragment byte offsets inside a stage: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); block bases are multiples of 16 rows, so the
        // swizzle term depends on the lane only; k-half 1 is k-half 0 with chunk bit 2 flipped (^ 64 bytes).
        const int off0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
#pragma unroll
        for (int i = 0; i < MRB; i++)
#pragma unroll
            for (int j = 0; j < MCB; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Software pipeline over half k-steps (32 deep) with ONE set of A fragments: right behind the MCB MFMAs that consumed row block
        // i, its registers are re-loaded with block i of the next half step ("rolling"), so every ds_read_b128 is in flight for a whole
        // half step (MRB x MCB MFMAs) and the 176 x 192 tile of QKV (132 accumulator registers) still fits 256 VGPRs; the B fragments
        // (MCB <= 6) are double-buffered.  No per-block guard: a wave with fewer than MRB row blocks (RB not a multiple of WR) also
        // multiplies the block after its last one -- in-bounds LDS rows of the same tile, a result that is never stored -- so the loop is
        // branch-free, and with no LDS-DMA in this branch hipcc's lgkmcnt bookkeeping stays exact (counted waits, no lgkmcnt(0)).
        bf16x8 fa[MRB], fb[2][MCB];
        auto a_ptr = [&](int kt, int ks) { return smem + (kt % NST) * STAGE + rb0 * 2048 + (ks ? (off0 ^ 64) : off0); };
        auto b_ptr = [&](int kt, int ks) { return smem + (kt % NST) * STAGE + (BM + cb0 * 16) * 128 + (ks ? (off0 ^ 64) : off0); };
        auto read_b = [&](int buf, const char* sb) {
            if ((ABL == 2 || ABL == 4) && sb != b_ptr(0, 0)) return;
#pragma unroll
            for (int j = 0; j < MCB; j++) fb[buf][j] = *reinterpret_cast<const bf16x8*>(sb + j * 2048);
        };
        // one half step: MFMAs of (fa, fb[buf]) with the rolling reload of fa from `sa_next` (RELOAD = false: last half step)
        auto half_step = [&](auto reload, int buf, const char* sa_next) {
            constexpr bool RELOAD = decltype(reload)::value;
#pragma unroll
            for (int i = 0; i < MRB; i++) {
                if (ABL == 1 || ABL == 2) {
                    asm volatile("" :: "v"(fa[i]), "v"(fb[buf][0]), "v"(fb[buf][MCB - 1]));
                } else {
#pragma unroll
                    for (int j = 0; j < MCB; j++) acc[i][j] = mfma_16x16x32<F16>(fa[i], fb[buf][j], acc[i][j]);
                }
                if (RELOAD && ABL != 2 && ABL != 4) fa[i] = *reinterpret_cast<const bf16x8*>(sa_next + i * 2048);
            }
        };
        constexpr std::integral_constant<bool, true> ROLL{};
        constexpr std::integral_constant<bool, false> LAST{};
        __builtin_amdgcn_s_barrier();                          // B_0: k-step 0 landed
        asm volatile("" ::: "memory");
        read_b(0, b_ptr(0, 0));
#pragma unroll
        for (int i = 0; i < MRB; i++) fa[i] = *reinterpret_cast<const bf16x8*>(a_ptr(0, 0) + i * 2048);
        for (int kt = 0; kt + 1 < nk; kt++) {
            read_b(1, b_ptr(kt, 1));
            half_step(ROLL, 0, a_ptr(kt, 1));
            // every read of stage kt % NST has been issued; the builtin (not an asm statement) lets hipcc know they have all returned
            __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();                      // B_{kt+1}
            asm volatile("" ::: "memory");
            read_b(0, b_ptr(kt + 1, 0));
            half_step(ROLL, 1, a_ptr(kt + 1, 0));
        }
        read_b(1, b_ptr(nk - 1, 1));
        half_step(ROLL, 0, a_ptr(nk - 1, 1));
        half_step(LAST, 1, nullptr);
    }

    G5_STAMP(2);
    __syncthreads();                                               // E1: the ring is dead
    G5_STAMP(3);
    if (wave < 4) {
        float* slab = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int j = 0; j < MCB; j++) {
            const int ncol = n0 + (cb0 + j) * 16 + fr;             // column in the padded weight layout
            if (EPI == EPI_QKV && n0 + (cb0 + j) * 16 >= 2 * p.D) {
                // V block (wave-uniform: 2 D is a multiple of 16): [feature][token] bf16, 4 consecutive tokens = 8 bytes per store
                const float bv = p.bias[ncol];
#pragma unroll
                for (int i = 0; i < MRB; i++) {
                    if (RB % WR == 0 || i < nrb) {
                        const int mrow = m0 + (rb0 + i) * 16 + fq * 4;
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; e++) pk[e] = (__bf16)(acc[i][j][e] + bv);
                        if (mrow < p.M) *reinterpret_cast<bf16x4*>(p.vt + (size_t)(ncol - 2 * p.D) * p.ldvt + mrow) = pk;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < MRB; i++) {
                    if (RB % WR == 0 || i < nrb) {
                        float* d = slab + ((rb0 + i) * 16 + fq * 4) * SLD + (cb0 + j) * 16 + fr;
#pragma unroll
                        for (int e = 0; e < 4; e++) d[e * SLD] = acc[i][j][e];
                    }
                }
            }
        }
    }
    __syncthreads();                                               // E2: the slab is complete
    G5_STAMP(4);

    // ---------------------------------------------------------------------- row phase, all eight waves: items of 8 rows x 64 columns
    constexpr int NPAN = CB / 4, ITEMS = NPAN * RB * 2;
    const float* slab = reinterpret_cast<const float*>(smem);
    for (int it = wave; it < ITEMS; it += 8) {
        const int pan = it % NPAN, ch = it / NPAN;
        const int m_base = m0 + ch * 8, n_base = n0 + pan * 64;
        const float* stg = slab + ch * 8 * SLD + pan * 64;
        if (EPI == EPI_GENERIC) {
            switch (p.act) {
                case ACT_GELU_TANH: epi_generic_rows<ACT_GELU_TANH, 64, 8, SLD>(p, stg, m_base, n_base, lane); break;
                case ACT_GELU_ERF: epi_generic_rows<ACT_GELU_ERF, 64, 8, SLD>(p, stg, m_base, n_base, lane); break;
                case ACT_MISH: epi_generic_rows<ACT_MISH, 64, 8, SLD>(p, stg, m_base, n_base, lane); break;
                case ACT_SILU: epi_generic_rows<ACT_SILU, 64, 8, SLD>(p, stg, m_base, n_base, lane); break;
                default: epi_generic_rows<ACT_NONE, 64, 8, SLD>(p, stg, m_base, n_base, lane); break;
            }
        } else if (n_base < 2 * p.D) {                             // (V panels were stored by the consumers)
            epi_qk_rows<64, 8, SLD>(p, stg, m_base, n_base, lane);
        }
    }
    if constexpr (ABL == 5) {
        if (p.stamps && (tid == 0 || tid == 256)) {
            const unsigned long long t5 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t6 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 2 + (tid >> 8)) * 8;
            for (int i = 0; i < 5; i++) o[i] = ts[i];
            o[5] = t5; o[6] = t6; o[7] = 0;
        }
    }
#undef G5_STAMP
}

template <bool F16, int EPI, int RB, int CB, int WR, int NST, int ABL = 0>
static hipError_t launch_gemm5_t(const GemmArgs& a, int n_pad, hipStream_t st) {
    using C = Gemm5Cfg<RB, CB, NST>;
    static unsigned attr_mask = 0;
    if (hipError_t e = f5_set_lds_attr(reinterpret_cast<const void*>(&gemm5_kernel<F16, EPI, RB, CB, WR, NST, ABL>), C::LDS, attr_mask); e != hipSuccess) return e;
    const int tiles_m = (a.M + C::BM - 1) / C::BM, tiles_n = n_pad / C::BN;
    hipLaunchKernelGGL((gemm5_kernel<F16, EPI, RB, CB, WR, NST, ABL>), dim3(tiles_m * tiles_n), dim3(512), C::LDS, st, a, tiles_n, n_pad);
    return hipGetLastError();
}

// (tile choice and the non-template entry points: gemm_launch.h / tu_gemm5_*.hip)
template <bool F16, int EPI>
static hipError_t launch_gemm5(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st) {
#ifdef F5HIP_GEMM5_ABL
    if (EPI == EPI_GENERIC && rb == 11 && (cb == 4 || cb == 8)) {
        static const int abl = getenv("F5HIP_GEMM5_ABL") ? atoi(getenv("F5HIP_GEMM5_ABL")) : 0;
        if (cb == 4) {
            if (abl == 1) return launch_gemm5_t<F16, EPI_GENERIC, 11, 4, 4, 4, 1>(a, n_pad, st);
            if (abl == 2) return launch_gemm5_t<F16, EPI_GENERIC, 11, 4, 4, 4, 2>(a, n_pad, st);
            if (abl == 3) return launch_gemm5_t<F16, EPI_GENERIC, 11, 4, 4, 4, 3>(a, n_pad, st);
            if (abl == 4) return launch_gemm5_t<F16, EPI_GENERIC, 11, 4, 4, 4, 4>(a, n_pad, st);
            if (abl == 5) return launch_gemm5_t<F16, EPI_GENERIC, 11, 4, 4, 4, 5>(a, n_pad, st);
        } else {
            if (abl == 1) return launch_gemm5_t<F16, EPI_GENERIC, 11, 8, 2, 4, 1>(a, n_pad, st);
            if (abl == 2) return launch_gemm5_t<F16, EPI_GENERIC, 11, 8, 2, 4, 2>(a, n_pad, st);
            if (abl == 3) return launch_gemm5_t<F16, EPI_GENERIC, 11, 8, 2, 4, 3>(a, n_pad, st);
            if (abl == 4) return launch_gemm5_t<F16, EPI_GENERIC, 11, 8, 2, 4, 4>(a, n_pad, st);
            if (abl == 5) return launch_gemm5_t<F16, EPI_GENERIC, 11, 8, 2, 4, 5>(a, n_pad, st);
        }
    }
#endif
    if (rb == 11) {
        if (cb == 4) return launch_gemm5_t<F16, EPI, 11, 4, 4, 4>(a, n_pad, st);
        if (cb == 8) return launch_gemm5_t<F16, EPI, 11, 8, 2, 4>(a, n_pad, st);
        if (cb == 12) return launch_gemm5_t<F16, EPI, 11, 12, 1, 3>(a, n_pad, st);
    }
    if (rb == 8) {
        if (cb == 4) return launch_gemm5_t<F16, EPI, 8, 4, 4, 4>(a, n_pad, st);
        if (cb == 8) return launch_gemm5_t<F16, EPI, 8, 8, 2, 4>(a, n_pad, st);
        if (cb == 12) return launch_gemm5_t<F16, EPI, 8, 12, 2, 4>(a, n_pad, st);
    }
    return hipErrorInvalidValue;
}
