// gemm5: exact-fit, full-line MFMA GEMM for gfx950 -- the production kernel of the transformer-block GEMMs (one 16-bit plane).
//   C[M,N] = A[M,K] * W[N,K]^T, fp16 or bf16 operands, fp32 accumulate, fused epilogues (GemmArgs of gemm_epilogue.h).
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
//
// Epilogue: accumulators -> one fp32 slab of the whole tile in LDS (it aliases the dead ring) -> all eight waves run a row phase with
// 16-byte row-contiguous residual loads / stores.  The in-kernel time lines (profiles/r02_gemm5_timeline_stamps*.txt) showed the
// epilogue's time following the amount of straight-line CODE it executes, not its bytes (V tiles 2.9 us, Q / K tiles 6.2, rotary tiles
// 8.9; 3.4 us for out and FF1 alike): every launch walks it once, instruction-cache cold.  Hence
//   * the W fragment is the MFMA's A operand (SWAP): a lane then holds 4 consecutive FEATURES of one token, and a 16 x 16 block goes to
//     the row-major slab with ONE ds_write_b128 instead of four ds_write_b32 (all-V tiles of the QKV projection keep the other order:
//     4 consecutive tokens per lane = one ds_write_b128 into the transposed slab their [feature][token] output wants);
//   * the row phase is a rolled, one-deep software-pipelined loop over groups of 4 rows (residual of the next group in flight while the
//     current one is finished), the (activation x residual x output x guard) variant chosen once per kernel outside it.
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
    static constexpr int SLD = BN + 4;                    // row-major slab [token][feature]: +4 floats spreads the ds_write_b128 of 8 lanes over all banks
    static constexpr int SLDT = BM + 4;                   // transposed slab [feature][token] (V blocks of the QKV projection)
    static constexpr int NPAN = CB / 4;                   // 64-column panels
    static constexpr int RING = NST * STAGE, SLAB = BM * SLD * 4, SLAB_T = BN * SLDT * 4;
    static constexpr int LDS = RING > SLAB ? (RING > SLAB_T ? RING : SLAB_T) : (SLAB > SLAB_T ? SLAB : SLAB_T);
    static_assert(CB % 4 == 0, "column blocks must split over 64-column panels");   // (LDS <= 160 KiB is checked where a kernel is launched: the W-direct kernels' ring holds A rows only)
};

// tile index of workgroup b: XCD-blocked when the grid divides over the 8 XCDs (speed only: any bijection is correct)
F5_DEVICE int gemm5_tile_of_block(int b, int n_tiles) {
    if (n_tiles & 7) return b;
    return (b & 7) * (n_tiles >> 3) + (b >> 3);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// consumer k-loop.  Software pipeline over half k-steps (32 deep) with ONE set of token-side fragments: right behind the MCB MFMAs that
// consumed row block i, its registers are re-loaded with block i of the next half step ("rolling"), so every ds_read_b128 is in flight
// for a whole half step (MRB x MCB MFMAs) and the 176 x 192 tile of QKV (132 accumulator registers) still fits 256 VGPRs; the W-side
// fragments (MCB <= 6) are double-buffered.  No per-block guard: a wave with fewer than MRB row blocks (RB not a multiple of WR) also
// multiplies the block after its last one -- in-bounds LDS rows of the same tile, a result that is never stored -- so the loop is
// branch-free, and with no LDS-DMA in this branch hipcc's lgkmcnt bookkeeping stays exact (counted waits, no lgkmcnt(0): with the
// builtin global_load_lds in the same loop every wait degraded to lgkmcnt(0), "pending flat").
// SWAP: the W fragment is the MFMA's A operand: acc[i][j][e] = C[token (rb0 + i) 16 + (lane & 15)][feature (cb0 + j) 16 + 4 (lane >> 4) + e];
// !SWAP: acc[i][j][e] = C[token (rb0 + i) 16 + 4 (lane >> 4) + e][feature (cb0 + j) 16 + (lane & 15)].
// ABL (diagnostics, -DF5HIP_GEMM5_ABL builds + F5HIP_GEMM5_ABL=<n> at run time; results are garbage): 1 = no MFMAs, 2 = no fragment
// reads and no MFMAs, 3 = no LDS-DMA inside the loop, 4 = MFMAs only (no DMA, no fragment reads), 5 = s_memrealtime stamps.
template <bool F16, int BM, int STAGE, int NST, int MRB, int MCB, bool SWAP, int ABL>
F5_DEVICE void g5_consume(const char* smem, int nk, int rb0, int cb0, int lane, f32x4 (&acc)[MRB][MCB]) {
    // fragment byte offsets inside a stage: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); block bases are multiples of 16 rows, so the
    // swizzle term depends on the lane only; k-half 1 is k-half 0 with chunk bit 2 flipped (^ 64 bytes)
    const int fr = lane & 15, fq = lane >> 4;
    const int off0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
#pragma unroll
    for (int i = 0; i < MRB; i++)
#pragma unroll
        for (int j = 0; j < MCB; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
                for (int j = 0; j < MCB; j++)
                    acc[i][j] = SWAP ? mfma_16x16x32<F16>(fb[buf][j], fa[i], acc[i][j]) : mfma_16x16x32<F16>(fa[i], fb[buf][j], acc[i][j]);
            }
            if (RELOAD && ABL != 2 && ABL != 4) fa[i] = *reinterpret_cast<const bf16x8*>(sa_next + i * 2048);
        }
    };
    constexpr std::integral_constant<bool, true> ROLL{};
    constexpr std::integral_constant<bool, false> LAST{};
    __builtin_amdgcn_s_barrier();                              // B_0: k-step 0 landed
    asm volatile("" ::: "memory");
    read_b(0, b_ptr(0, 0));
#pragma unroll
    for (int i = 0; i < MRB; i++) fa[i] = *reinterpret_cast<const bf16x8*>(a_ptr(0, 0) + i * 2048);
    for (int kt = 0; kt + 1 < nk; kt++) {
        read_b(1, b_ptr(kt, 1));
        half_step(ROLL, 0, a_ptr(kt, 1));
        // every read of stage kt % NST has been issued; the builtin (not an asm statement) lets hipcc know they have all returned
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                          // B_{kt+1}
        asm volatile("" ::: "memory");
        read_b(0, b_ptr(kt + 1, 0));
        half_step(ROLL, 1, a_ptr(kt + 1, 0));
    }
    read_b(1, b_ptr(nk - 1, 1));
    half_step(ROLL, 0, a_ptr(nk - 1, 1));
    half_step(LAST, 1, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// consumer k-loop with the W fragments STRAIGHT FROM GLOBAL MEMORY ("W-direct", round 3).  In the 1 x 4 consumer layout a wave's column
// blocks are its own: nobody else in the workgroup multiplies those W rows, so staging them through LDS only costs -- half of the LDS-DMA
// pieces of a 176 x 192 tile (the loaders' issue time, ~200 cycles per piece next to busy fragment reads, is what bounds that k-loop:
// 0.94 us per k-step against 0.5 of MFMA work) and a fifth of the LDS reads.  The weights are kept a second time in FRAGMENT ORDER
// (GemmArgs::Wf, packed once at load: block (16 columns, 32 k) = 64 lanes x 16 bytes, lane = (column & 15) + 16 (k chunk), blocks k-major
// within a column block), so a fragment is ONE fully coalesced 1 KiB wave load.  Four half steps of fragments are in flight (L2 latency
// ~1 us against ~0.25 us per half step): a ring of four register sets, hence the loop is unrolled over TWO k-steps (K % 128 == 0) and every
// slot index is a compile-time constant; loads past the end re-read the last half step (branch-free).  The compiler counts vmcnt exactly:
// these waves issue no other vector-memory instruction inside the loop.  Barrier protocol and the rolling A fragments: g5_consume's.
// Registers: the 176 x 192 tile (132 accumulator registers + 48 of W fragments) has no room for g5_consume's eleven rolling A fragments, so
// the row blocks of a half step go in TWO passes over a window of W1 = ceil(MRB / 2) fragments: behind the MFMAs of block i (pass 1) its
// registers take block W1 + i of the SAME half step, behind those of pass 2 block i of the NEXT one.  Consequence for the ring: the second
// half of k-step kt still reads stage kt AFTER barrier B_{kt+1}, so a stage is dead one barrier later than in g5_consume and the loaders
// refill stage kt - 1 (not kt) behind B_{kt+1} -- the ring (A rows only: 22 KiB stages) is two stages deeper instead.
template <bool F16, int BM, int STAGE, int NST, int MRB, int MCB, bool SWAP>
F5_DEVICE void g5_consume_wd(const char* smem, const char* wf, size_t wf_jstride, int nk, int rb0, int lane, f32x4 (&acc)[MRB][MCB]) {
    constexpr int W1 = (MRB + 1) / 2, W2 = MRB - W1;
    const int fr = lane & 15, fq = lane >> 4;
    const int off0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
#pragma unroll
    for (int i = 0; i < MRB; i++)
#pragma unroll
        for (int j = 0; j < MCB; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[W1], fb[4][MCB];
    const int h_last = 2 * nk - 1;
    auto a_ptr = [&](int kt, int ks) { return smem + (kt % NST) * STAGE + rb0 * 2048 + (ks ? (off0 ^ 64) : off0); };
    auto load_w = [&](auto slot_t, int h) {
        constexpr int S = decltype(slot_t)::value;
        const char* src = wf + (size_t)min(h, h_last) * 1024;
#pragma unroll
        for (int j = 0; j < MCB; j++) fb[S][j] = *reinterpret_cast<const bf16x8*>(src + j * wf_jstride);
    };
    // one half step: W fragments of slot S; sa_cur = this half step's A rows (blocks W1 .. of pass 2), sa_next = the next half step's
    auto half_step = [&](auto slot_t, auto reload, const char* sa_cur, const char* sa_next) {
        constexpr int S = decltype(slot_t)::value;
        constexpr bool RELOAD = decltype(reload)::value;
#pragma unroll
        for (int i = 0; i < W1; i++) {
#pragma unroll
            for (int j = 0; j < MCB; j++)
                acc[i][j] = SWAP ? mfma_16x16x32<F16>(fb[S][j], fa[i], acc[i][j]) : mfma_16x16x32<F16>(fa[i], fb[S][j], acc[i][j]);
            if (i < W2) fa[i] = *reinterpret_cast<const bf16x8*>(sa_cur + (W1 + i) * 2048);
            else if (RELOAD) fa[i] = *reinterpret_cast<const bf16x8*>(sa_next + i * 2048);
        }
#pragma unroll
        for (int i = 0; i < W2; i++) {
#pragma unroll
            for (int j = 0; j < MCB; j++)
                acc[W1 + i][j] = SWAP ? mfma_16x16x32<F16>(fb[S][j], fa[i], acc[W1 + i][j]) : mfma_16x16x32<F16>(fa[i], fb[S][j], acc[W1 + i][j]);
            if (RELOAD) fa[i] = *reinterpret_cast<const bf16x8*>(sa_next + i * 2048);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    constexpr std::integral_constant<bool, true> ROLL{};
    constexpr std::integral_constant<bool, false> LAST{};
    load_w(S0{}, 0); load_w(S1{}, 1); load_w(S2{}, 2); load_w(S3{}, 3);
    __builtin_amdgcn_s_barrier();                              // B_0: k-step 0 landed
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < W1; i++) fa[i] = *reinterpret_cast<const bf16x8*>(a_ptr(0, 0) + i * 2048);
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        const int h = 2 * kt;
        half_step(S0{}, ROLL, a_ptr(kt, 0), a_ptr(kt, 1));
        load_w(S0{}, h + 4);
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): every read issued so far has returned (stage kt - 1 is dead)
        __builtin_amdgcn_s_barrier();                          // B_{kt+1}
        asm volatile("" ::: "memory");
        half_step(S1{}, ROLL, a_ptr(kt, 1), a_ptr(kt + 1, 0));
        load_w(S1{}, h + 5);
        half_step(S2{}, ROLL, a_ptr(kt + 1, 0), a_ptr(kt + 1, 1));
        load_w(S2{}, h + 6);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                          // B_{kt+2}
        asm volatile("" ::: "memory");
        half_step(S3{}, ROLL, a_ptr(kt + 1, 1), a_ptr(kt + 2, 0));
        load_w(S3{}, h + 7);
    }
    // the last two k-steps (kt = nk - 2): one barrier, no reload behind the last half step
    half_step(S0{}, ROLL, a_ptr(kt, 0), a_ptr(kt, 1));
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                              // B_{nk-1}
    asm volatile("" ::: "memory");
    half_step(S1{}, ROLL, a_ptr(kt, 1), a_ptr(kt + 1, 0));
    half_step(S2{}, ROLL, a_ptr(kt + 1, 0), a_ptr(kt + 1, 1));
    half_step(S3{}, LAST, a_ptr(kt + 1, 1), nullptr);
}

// consumers: accumulators -> slab.  ROWMAJOR: slab[token][feature] (stride SLD); else slab[feature][token] (stride SLDT, + bias: V blocks).
// One ds_write_b128 per block when the lane's 4 values are contiguous in the target (SWAP & ROWMAJOR, or !SWAP & transposed).
template <int RB, int CB, int NST, int WR, int MRB, int MCB, bool SWAP, bool ROWMAJOR>
F5_DEVICE void g5_write_slab(f32x4 (&acc)[MRB][MCB], float* slab, int rb0, int nrb, int cb0, int lane, const float* bias_tile, int j_first = 0) {
    using C = Gemm5Cfg<RB, CB, NST>;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < MCB; j++) {
        if (j < j_first) continue;                             // (straddling K | V tile: only the V blocks go to the transposed slab)
        f32x4 bj = {0.f, 0.f, 0.f, 0.f};
        if (!ROWMAJOR) {
            if (SWAP) bj = *reinterpret_cast<const f32x4*>(bias_tile + (cb0 + j) * 16 + fq * 4);
            else { const float b1 = bias_tile[(cb0 + j) * 16 + fr]; bj = (f32x4){b1, b1, b1, b1}; }
        }
#pragma unroll
        for (int i = 0; i < MRB; i++) {
            if (RB % WR == 0 || i < nrb) {
                if (SWAP && ROWMAJOR) {
                    *reinterpret_cast<f32x4*>(slab + ((rb0 + i) * 16 + fr) * C::SLD + (cb0 + j) * 16 + fq * 4) = acc[i][j];
                } else if (!SWAP && !ROWMAJOR) {
                    *reinterpret_cast<f32x4*>(slab + ((cb0 + j) * 16 + fr) * C::SLDT + (rb0 + i) * 16 + fq * 4) = acc[i][j] + bj;
                } else if (SWAP) {                             // transposed from the SWAP layout: four scalars (V blocks of a straddling tile)
                    float* d = slab + ((cb0 + j) * 16 + fq * 4) * C::SLDT + (rb0 + i) * 16 + fr;
#pragma unroll
                    for (int e = 0; e < 4; e++) d[e * C::SLDT] = acc[i][j][e] + bj[e];
                } else {                                       // row-major from the !SWAP layout (not used on the path)
                    float* d = slab + ((rb0 + i) * 16 + fq * 4) * C::SLD + (cb0 + j) * 16 + fr;
#pragma unroll
                    for (int e = 0; e < 4; e++) d[e * C::SLD] = acc[i][j][e];
                }
            }
        }
    }
}

// The arithmetic of the generic epilogue on 4 consecutive features of one row, shared by every epilogue form of gemm5 / gemm6 (slab row
// phase, direct-from-accumulator): one function, explicit fma, so that which kernel computed a row cannot change its bits.
template <int ACT, bool RES>
F5_DEVICE f32x4 g5_epi_value(f32x4 acc, f32x4 bias, f32x4 mul, f32x4 res, bool zero_row) {
    f32x4 v = acc + bias;
    if (ACT != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = apply_act(v[e], ACT);
    }
    if (zero_row) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = RES ? __builtin_fmaf(v[e], mul[e], res[e]) : __fmul_rn(v[e], mul[e]);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// generic row phase: v = act(acc + bias); rows with row_keep == 0 -> 0; v = v * mul + res; fp32 and / or 16-bit outputs.
// Work unit = one group of 4 rows x the tile's NPAN panels of 64 columns (lane -> row lane >> 4, columns 4 (lane & 15) .. + 3 of each
// panel); groups are dealt round-robin over the 8 waves; the loop is rolled, the next group's residual is loaded before the current
// group is finished.
template <int ACT, bool RES, bool OUTF, int OUTS, bool GUARD, int RB, int CB, int NST, typename WriteSlab>
F5_DEVICE void g5_generic_tail(const GemmArgs& p, const float* slab, int m0, int n0, int wave, int lane, WriteSlab write_slab) {
    using C = Gemm5Cfg<RB, CB, NST>;
    constexpr int NPAN = C::NPAN, NG = RB * 4;                 // row groups of the tile
    const int r_in = lane >> 4, c4 = (lane & 15) * 4;
    f32x4 bv[NPAN], mv[NPAN];
    bool nok[NPAN];
#pragma unroll
    for (int pn = 0; pn < NPAN; pn++) {
        const int n = n0 + pn * 64 + c4;
        nok[pn] = GUARD ? n < p.N : true;
        bv[pn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mv[pn] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (p.bias && nok[pn]) bv[pn] = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.mul && nok[pn]) mv[pn] = *reinterpret_cast<const f32x4*>(p.mul + n);
    }
    f32x4 rs[RES ? NPAN : 1], rn[RES ? NPAN : 1];
    auto load_res = [&](int g, f32x4 (&dst)[RES ? NPAN : 1]) {
        if (!RES) return;
        const int row = m0 + g * 4 + r_in;
#pragma unroll
        for (int pn = 0; pn < NPAN; pn++) {
            dst[pn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g < NG && (!GUARD || (nok[pn] && row < p.M))) dst[pn] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldres + n0 + pn * 64 + c4);
        }
    };
    load_res(wave, rs);                                            // in flight across the slab write and its barrier
    write_slab();
    __syncthreads();                                               // E2: the slab is complete
#pragma unroll 1
    for (int g = wave; g < NG; g += 8) {
        load_res(g + 8, rn);
        const int rl = g * 4 + r_in, row = m0 + rl;
        int keep = 1;
        if (GUARD && p.row_keep && row < p.M) keep = p.row_keep[row];
#pragma unroll
        for (int pn = 0; pn < NPAN; pn++) {
            const int n = n0 + pn * 64 + c4;
            const f32x4 v = g5_epi_value<ACT, RES>(*reinterpret_cast<const f32x4*>(slab + rl * C::SLD + pn * 64 + c4), bv[pn], mv[pn], rs[RES ? pn : 0], GUARD && !keep);
            if (!GUARD || (nok[pn] && row < p.M)) {
                if (OUTF) *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)row * p.ldo + n) = v;
                const float vv[4] = {v[0], v[1], v[2], v[3]};
                if (OUTS == 2) {
                    store_f16x4(p.out_hi + (size_t)row * p.ldob + n, vv);
                } else if (OUTS == 1) {
                    bf16x4 hi, lo;
                    split_bf16x4(vv, hi, lo);
                    *reinterpret_cast<bf16x4*>(p.out_hi + (size_t)row * p.ldob + n) = hi;
                    if (p.out_lo) *reinterpret_cast<bf16x4*>(p.out_lo + (size_t)row * p.ldob + n) = lo;
                }
            }
        }
        if (RES) {
#pragma unroll
            for (int pn = 0; pn < NPAN; pn++) rs[pn] = rn[pn];
        }
    }
}

template <int ACT, bool GUARD, int RB, int CB, int NST, typename WriteSlab>
F5_DEVICE void g5_generic_variants(const GemmArgs& p, const float* slab, int m0, int n0, int wave, int lane, WriteSlab ws) {
    const bool res = p.res != nullptr, outf = p.out_f32 != nullptr, outs = p.out_hi != nullptr;
    // the (residual, fp32 out, 16-bit out) combinations in use on the path: same table as epi_generic_rows_g (gemm_epilogue.h)
    if (ACT == ACT_NONE) {
        if (res) {
            if (outf && outs) g5_generic_tail<ACT, true, true, 1, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
            else if (outf) g5_generic_tail<ACT, true, true, 0, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
            else g5_generic_tail<ACT, true, false, 1, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
        } else {
            if (outf && outs) g5_generic_tail<ACT, false, true, 1, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
            else if (outf) g5_generic_tail<ACT, false, true, 0, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
            else if (p.f16_out) g5_generic_tail<ACT, false, false, 2, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
            else g5_generic_tail<ACT, false, false, 1, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
        }
    } else {
        if (res) g5_generic_tail<ACT, true, true, 0, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
        else if (outs && !outf && p.f16_out) g5_generic_tail<ACT, false, false, 2, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
        else if (outs && !outf) g5_generic_tail<ACT, false, false, 1, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
        else if (outf && !outs) g5_generic_tail<ACT, false, true, 0, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
        else g5_generic_tail<ACT, false, true, 1, GUARD, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
    }
}

template <int RB, int CB, int NST, typename WriteSlab>
F5_DEVICE void g5_generic_epilogue(const GemmArgs& p, const float* slab, int m0, int n0, int wave, int lane, WriteSlab ws) {
    // workgroup-uniform: interior tile without per-row special cases
    const bool interior = m0 + RB * 16 <= p.M && n0 + CB * 16 <= p.N && !p.row_keep;
#define G5_ACT(A)                                                                                 \
    if (interior) g5_generic_variants<A, false, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);   \
    else g5_generic_variants<A, true, RB, CB, NST>(p, slab, m0, n0, wave, lane, ws);
    switch (p.act) {
        case ACT_GELU_TANH: G5_ACT(ACT_GELU_TANH) break;
        case ACT_GELU_ERF: G5_ACT(ACT_GELU_ERF) break;
        case ACT_MISH: G5_ACT(ACT_MISH) break;
        case ACT_SILU: G5_ACT(ACT_SILU) break;
        default: G5_ACT(ACT_NONE) break;
    }
#undef G5_ACT
}

// Q / K rows of the fused QKV projection: bias, rotary embedding on head 0 (x-transformers interleaved pairs, applied before the head
// split: F/model/modules.py:414-419), q * log2(e) / 8 (softmax scale, base-2 exponents), fp16 row-major [M][2 D].  A 192-column tile can straddle
// the Q | K or the K | V boundary: both are multiples of 64, so every 64-column panel is of one kind.
template <int RB, int CB, int NST>
F5_DEVICE void g5_qk_rows(const GemmArgs& p, const float* slab, int m0, int n0, int wave, int lane) {
    using C = Gemm5Cfg<RB, CB, NST>;
    constexpr int NPAN = C::NPAN, NG = RB * 4, NT = (NG + 7) / 8;
    const int D = p.D, r_in = lane >> 4, c4 = (lane & 15) * 4;
    f32x4 bv[NPAN];
#pragma unroll
    for (int pn = 0; pn < NPAN; pn++) bv[pn] = *reinterpret_cast<const f32x4*>(p.bias + n0 + pn * 64 + c4);
    // head 0 of q = columns [0, 64), of k = [D, D + 64): at most one panel of a tile (workgroup-uniform)
    int rot_pn = -1;
#pragma unroll
    for (int pn = 0; pn < NPAN; pn++) {
        const int n_base = n0 + pn * 64;
        if (n_base == 0 || n_base == D) rot_pn = pn;
    }
    // rotary factors of this wave's rows, all fetched up front (two dependent global loads per row group: inside the rolled loop they
    // were ~1 us per group, and the 32 rotary tiles of the C2 launch finished 3 us behind the other 224)
    float2 cs[NT], sn[NT];
    if (rot_pn >= 0) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int row = m0 + (wave + 8 * t) * 4 + r_in;
            const int pos = (wave + 8 * t < NG && row < p.M) ? (p.row_pos ? p.row_pos[row] : row) : 0;   // (row_pos == null: the tables are per ROW, GemmArgs::rope_cos)
            cs[t] = *reinterpret_cast<const float2*>(p.rope_cos + pos * 32 + (c4 >> 1));
            sn[t] = *reinterpret_cast<const float2*>(p.rope_sin + pos * 32 + (c4 >> 1));
        }
    }
    auto finish = [&](int g, bool rot, float2 c, float2 s2) {
        const int rl = g * 4 + r_in, row = m0 + rl;
        const bool rok = row < p.M;
#pragma unroll
        for (int pn = 0; pn < NPAN; pn++) {
            const int n_base = n0 + pn * 64;
            if (n_base < 2 * D) {                                  // (wave-uniform)
                const int which = n_base / D, nd = n_base - which * D + c4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + rl * C::SLD + pn * 64 + c4) + bv[pn];
                const float qs = which == 0 ? F5_Q_SCALE : 1.0f;
                float o[4];
                if (rot && pn == rot_pn) {
                    // explicit product + fma: left to the compiler, the contraction of a*c - b*s differed between instantiations of the
                    // round-1 epilogue -- a 1-ulp flip in a few q values that the 22-layer sampler amplifies to 5e-4
                    o[0] = __builtin_fmaf(v[0], c.x, -__fmul_rn(v[1], s2.x)) * qs;
                    o[1] = __builtin_fmaf(v[1], c.x, __fmul_rn(v[0], s2.x)) * qs;
                    o[2] = __builtin_fmaf(v[2], c.y, -__fmul_rn(v[3], s2.y)) * qs;
                    o[3] = __builtin_fmaf(v[3], c.y, __fmul_rn(v[2], s2.y)) * qs;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = v[e] * qs;
                }
                if (rok) store_f16x4(p.qk + (size_t)row * (2 * D) + which * D + nd, o);   // fp16, saturating (attention operands: common.h)
            }
        }
    };
    if (rot_pn >= 0) {
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (wave + 8 * t < NG) finish(wave + 8 * t, true, cs[t], sn[t]);
    } else {
#pragma unroll 1
        for (int g = wave; g < NG; g += 8) finish(g, false, make_float2(1.f, 1.f), make_float2(0.f, 0.f));
    }
}

// V rows: slab[feature][token] (bias added by the writer) -> [D][ldvt] fp16: lane-linear (feature, 4 tokens) pairs, 8-byte stores,
// 2 RB x 16-byte runs of tokens per feature row
template <int RB, int CB, int NST>
F5_DEVICE void g5_v_rows(const GemmArgs& p, const float* slab, int m0, int n0, int f_lo, int wave, int lane) {
    using C = Gemm5Cfg<RB, CB, NST>;
    constexpr int T4 = RB * 4;                                     // groups of 4 tokens per feature row
    const int n_items = (CB * 16 - f_lo) * T4;
#pragma unroll 2
    for (int idx = wave * 64 + lane; idx < n_items; idx += 512) {
        const int f = f_lo + idx / T4, t4 = idx % T4;
        const int tok = m0 + t4 * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(slab + f * C::SLDT + t4 * 4);
        const float pk[4] = {v[0], v[1], v[2], v[3]};
        if (tok < p.M) store_f16x4(p.vt + (size_t)(n0 + f - 2 * p.D) * p.ldvt + vt_col(tok), pk);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LNE (EXPERIMENT, instantiated in -DF5HIP_EXPERIMENTS builds only): the LayerNorm that FOLLOWS this residual GEMM (out projection -> norm 2, FF2 -> the next block's norm 1) fused behind its epilogue.
// Only launched as ONE resident wave of workgroups (grid <= CUs, one workgroup per CU by its LDS size) with 16 column tiles per row slab:
// when a slab's 16 workgroups have stored their tiles of the residual stream they meet at a slab-local barrier (an arrival counter in L2:
// the XCD-blocked tile order puts them on one XCD -- probed at start-up, f5hip.hip), then workgroup j normalises rows j * BM / 16 ... of
// the slab (one wave per row, ln_finish = the stand-alone kernel's arithmetic, so the bits are the same) and writes the operand plane
// of the next GEMM.  The rows are read with agent-scope loads (this CU's L1 may still hold the tile's own residual columns from before
// the update).  MEASURED AND NOT SHIPPED: bit-identical, but 31.2 us against 18.8 + 6.1 us for the two kernels it replaces (and the same
// norm fused in FRONT of the following GEMM measured 2-5 % slower end to end): DESIGN.md section 6, profiles/r02_ln_fusion.txt.  The spin is bounded (~50 ms):
// on a time-out the workgroup sets *p.ln_err and carries on (wrong results, no hang); the host then stops using these kernels.
F5_DEVICE f32x4 g5_load_f4_agent(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// WD: W fragments straight from global memory (g5_consume_wd): the ring holds A rows only; needs the 1 x 4 consumer layout, K % 128 == 0, p.Wf.
template <bool F16, int EPI, int RB, int CB, int WR, int NST, int ABL = 0, bool LNE = false, bool WD = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm5_kernel(const GemmArgs p, const int tiles_n, const int n_rows_w) {
    using C = Gemm5Cfg<RB, CB, NST>;
    static_assert(!WD || (WR == 1 && ABL == 0 && !LNE), "W-direct: 1 x 4 consumer layout, production kernels only");
    constexpr int BM = C::BM, BN = C::BN, PIECES = WD ? BM / 8 : C::PIECES, STAGE = PIECES * 1024;
    constexpr int WC = 4 / WR;                                 // the 4 consumer waves tile the block grid WR (rows) x WC (columns)
    constexpr int MRB = (RB + WR - 1) / WR, MCB = CB / WC;
    static_assert(CB % WC == 0 && NST >= 3 && (WD ? NST <= 6 : NST <= 4), "bad tile configuration");
    constexpr int P_HI = (PIECES + 3) / 4, P_LO = PIECES / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = gemm5_tile_of_block(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nk = p.K >> 6;
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    // time line: ABL == 5 builds keep the stamps in registers; the W-direct kernels write them straight to p.stamps when it is set (run time,
    // tools/gemm5_stamps.py with F5HIP_GEMM5_STAMPS=1: no special build)
    unsigned long long* const stamp_o = (WD && p.stamps && (tid == 0 || tid == 256)) ? p.stamps + ((size_t)blockIdx.x * 2 + (tid >> 8)) * 8 : nullptr;
#define G5_STAMP(I) if constexpr (ABL == 5) { if (tid == 0 || tid == 256) ts[I] = __builtin_amdgcn_s_memrealtime(); } else if constexpr (WD) { if (stamp_o) stamp_o[I] = __builtin_amdgcn_s_memrealtime(); }
    G5_STAMP(0);
    // consumer geometry (also used by the epilogue): row blocks rb0 .. rb0 + nrb - 1, column blocks cb0 .. cb0 + MCB - 1
    const int cw = wave & 3;
    const int wr = cw / WC, wc = cw % WC;
    const int rb0 = (wr * RB) / WR, nrb = ((wr + 1) * RB) / WR - rb0;   // wave-uniform
    const int cb0 = wc * MCB;
    // all-V tiles of the QKV projection keep the token on the accumulator registers (their output is [feature][token])
    // (W-direct kernels keep the SWAP layout for V tiles too: one k-loop instantiation instead of two -- the 176 x 192 QKV kernel has no
    // registers to spare -- and the transposed slab is written with four scalar stores per block instead of one 16-byte store)
    const bool swap = WD || !(EPI == EPI_QKV && n0 >= 2 * p.D);
    f32x4 acc[MRB][MCB];

    // Ring protocol.  All NST stages are filled up front.  Barrier B_{kt+1} sits between the two 32-deep halves of k-step kt: behind it
    // "k-step kt + 1 has landed" (every loader wave waited for its own pieces) and "every read of stage kt % NST has returned" (every
    // consumer waited lgkmcnt(0)), so the loaders issue k-step kt + NST into that stage at once: one stage being consumed, NST - 1 in flight.
    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves: they sit in VMEM issue for the whole k-loop
        // (the CU's address path takes ~26 cycles per 1 KiB piece)
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
            static_assert(5 * P_HI <= 63, "vmcnt immediate");
            if (mine == P_HI) {
                if (newer >= 5) wait_vmcnt<5 * P_HI>(); else if (newer == 4) wait_vmcnt<4 * P_HI>(); else
                if (newer == 3) wait_vmcnt<3 * P_HI>(); else if (newer == 2) wait_vmcnt<2 * P_HI>(); else if (newer == 1) wait_vmcnt<P_HI>(); else wait_vmcnt<0>();
            } else {
                if (newer >= 5) wait_vmcnt<5 * P_LO>(); else if (newer == 4) wait_vmcnt<4 * P_LO>(); else
                if (newer == 3) wait_vmcnt<3 * P_LO>(); else if (newer == 2) wait_vmcnt<2 * P_LO>(); else if (newer == 1) wait_vmcnt<P_LO>(); else wait_vmcnt<0>();
            }
        };
#pragma unroll
        for (int t = 0; t < NST; t++)
            if (t < nk) issue_tile(t);
        wait_landed(min(NST - 1, nk - 1));
        G5_STAMP(1);
        __builtin_amdgcn_s_barrier();                          // B_0
        if constexpr (WD) {
            // g5_consume_wd reads stage kt until B_{kt+2}: behind B_{kt+1} the free stage is that of k-step kt - 1
            for (int kt = 0; kt + 1 < nk; kt++) {
                wait_landed(min(NST - 3, nk - 2 - kt));        // k-step kt + 1 (issued so far: up to kt + NST - 2)
                __builtin_amdgcn_s_barrier();                  // B_{kt+1}
                if (kt >= 1 && kt - 1 + NST < nk) issue_tile(kt - 1 + NST);
            }
        } else
        for (int kt = 0; kt + 1 < nk; kt++) {
            wait_landed(min(NST - 2, nk - 2 - kt));            // k-step kt + 1 (k-steps kt + 2 .. kt + NST - 1 stay in flight)
            __builtin_amdgcn_s_barrier();                      // B_{kt+1}
            if (kt + NST < nk) issue_tile(kt + NST);
        }
    } else if constexpr (WD) {
        const size_t jstride = (size_t)(p.K >> 5) * 1024;      // bytes between the fragments of two column blocks
        const char* wf = reinterpret_cast<const char*>(p.Wf) + (size_t)(n0 / 16 + cb0) * jstride + lane * 16;
        g5_consume_wd<F16, BM, STAGE, NST, MRB, MCB, true>(smem, wf, jstride, nk, rb0, lane, acc);
    } else if (swap) {
        g5_consume<F16, BM, STAGE, NST, MRB, MCB, true, ABL>(smem, nk, rb0, cb0, lane, acc);
    } else {
        g5_consume<F16, BM, STAGE, NST, MRB, MCB, false, ABL>(smem, nk, rb0, cb0, lane, acc);
    }

    G5_STAMP(2);
    __syncthreads();                                               // E1: the ring is dead
    G5_STAMP(3);
    float* slab = reinterpret_cast<float*>(smem);
    if constexpr (EPI == EPI_GENERIC) {
        g5_generic_epilogue<RB, CB, NST>(p, slab, m0, n0, wave, lane, [&]() {
            if (wave < 4) g5_write_slab<RB, CB, NST, WR, MRB, MCB, true, true>(acc, slab, rb0, nrb, cb0, lane, nullptr);
        });
    } else if (swap) {
        // Q / K tile (possibly with V blocks behind the K | V boundary): row-major slab, rolled row phase; then the V blocks, if any
        const int f_lo = max(0, 2 * p.D - n0);                     // first V column of this tile (>= BN: none; 0: an all-V tile of a W-direct kernel)
        if (f_lo > 0) {                                            // (workgroup-uniform)
            if (wave < 4) g5_write_slab<RB, CB, NST, WR, MRB, MCB, true, true>(acc, slab, rb0, nrb, cb0, lane, nullptr);
            __syncthreads();                                       // E2
            g5_qk_rows<RB, CB, NST>(p, slab, m0, n0, wave, lane);
        }
        if (f_lo < BN) {                                           // (workgroup-uniform)
            if (f_lo > 0) __syncthreads();                         // the Q / K rows are done with the slab
            // column blocks at or behind the boundary (a multiple of 16) go to the transposed slab
            if (wave < 4) g5_write_slab<RB, CB, NST, WR, MRB, MCB, true, false>(acc, slab, rb0, nrb, cb0, lane, p.bias + n0, max(0, f_lo / 16 - cb0));
            __syncthreads();
            g5_v_rows<RB, CB, NST>(p, slab, m0, n0, f_lo, wave, lane);
        }
    } else {
        if (wave < 4) g5_write_slab<RB, CB, NST, WR, MRB, MCB, false, false>(acc, slab, rb0, nrb, cb0, lane, p.bias + n0);
        __syncthreads();                                           // E2
        g5_v_rows<RB, CB, NST>(p, slab, m0, n0, 0, wave, lane);
    }
    if constexpr (LNE) {
        static_assert(EPI == EPI_GENERIC && Gemm5Cfg<RB, CB, NST>::BM % 16 == 0, "16 workgroups share the rows of a slab of the residual stream");
        constexpr int ROWS = Gemm5Cfg<RB, CB, NST>::BM / 16;       // rows this workgroup normalises (11 or 8)
        static_assert(ROWS <= 16, "at most two rows per wave");
        const int slab_i = tile / tiles_n, j = tile - slab_i * tiles_n;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's part of the tile is in L2 (write-through L1)
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(p.ln_sync + slab_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ((int)(__hip_atomic_load(p.ln_sync + slab_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.ln_target) < 0) {
                if (++spins > (1 << 15)) { *p.ln_err = 1; break; }  // ~50 ms: the slab's other workgroups are not running
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // rows wave and wave + 8 of this workgroup's share: every load (both rows, scale, shift) in flight before the first use
        const int r0 = m0 + j * ROWS + wave, r1 = wave + 8 < ROWS ? r0 + 8 : p.ln.M;   // (r1 == M: no second row)
        float4 v0[4], v1[4], sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = (i * 64 + lane) * 4;
            const f32x4 a = g5_load_f4_agent(p.ln.x + (size_t)min(r0, p.ln.M - 1) * p.ln.ldx + c);
            const f32x4 b = g5_load_f4_agent(p.ln.x + (size_t)min(r1, p.ln.M - 1) * p.ln.ldx + c);
            v0[i] = make_float4(a[0], a[1], a[2], a[3]);
            v1[i] = make_float4(b[0], b[1], b[2], b[3]);
        }
        ln_load_mod<4>(p.ln, lane, sc, sh);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the asm loads are invisible to the compiler's own counting)
        ln_finish<4>(p.ln, r0, lane, v0, sc, sh);
        ln_finish<4>(p.ln, r1, lane, v1, sc, sh);
    }
    G5_STAMP(4);
    if constexpr (WD) {
        if (stamp_o) {
            stamp_o[5] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp_o[6] = __builtin_amdgcn_s_memrealtime();
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

template <bool F16, int EPI, int RB, int CB, int WR, int NST, int ABL = 0, bool LNE = false, bool WD = false>
static hipError_t launch_gemm5_t(const GemmArgs& a, int n_pad, hipStream_t st) {
    using C = Gemm5Cfg<RB, CB, NST>;
    // LDS: the ring (A rows only with W-direct) or the epilogue slab that aliases it, whichever is larger
    constexpr int ring = WD ? NST * (C::BM / 8) * 1024 : C::RING;
    constexpr int lds = ring > C::SLAB ? (ring > C::SLAB_T ? ring : C::SLAB_T) : (C::SLAB > C::SLAB_T ? C::SLAB : C::SLAB_T);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    if (WD && (!a.Wf || a.K % 128)) return hipErrorInvalidValue;
    static unsigned attr_mask = 0;
    if (hipError_t e = f5_set_lds_attr(reinterpret_cast<const void*>(&gemm5_kernel<F16, EPI, RB, CB, WR, NST, ABL, LNE, WD>), lds, attr_mask); e != hipSuccess) return e;
    const int tiles_m = (a.M + C::BM - 1) / C::BM, tiles_n = n_pad / C::BN;
    if (LNE && (tiles_n != 16 || !a.ln_sync || !a.ln_err || a.ln.D != 1024 || a.ln.x != a.out_f32 || a.ln.dw_w)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm5_kernel<F16, EPI, RB, CB, WR, NST, ABL, LNE, WD>), dim3(tiles_m * tiles_n), dim3(512), lds, st, a, tiles_n, n_pad);
    return hipGetLastError();
}

// residual GEMM + the LayerNorm behind it (out projection, FF2: generic epilogue, 64 columns = 16 column tiles of a 1024-wide stream)
template <bool F16>
static hipError_t launch_gemm5_lne(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st) {
    if (cb != 4) return hipErrorInvalidValue;
    if (rb == 11) return launch_gemm5_t<F16, EPI_GENERIC, 11, 4, 4, 4, 0, true>(a, n_pad, st);
    if (rb == 8) return launch_gemm5_t<F16, EPI_GENERIC, 8, 4, 4, 4, 0, true>(a, n_pad, st);
    return hipErrorInvalidValue;
}

// (tile choice and the non-template entry points: gemm_launch.h / tu_gemm5_*.hip)
template <bool F16, int EPI>
static hipError_t launch_gemm5(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st) {
#ifdef F5HIP_GEMM5_ABL
    static const int abl = getenv("F5HIP_GEMM5_ABL") ? atoi(getenv("F5HIP_GEMM5_ABL")) : 0;
    if (EPI == EPI_QKV && rb == 11 && cb == 12) {
        if (abl == 5) return launch_gemm5_t<F16, EPI_QKV, 11, 12, 1, 3, 5>(a, n_pad, st);
        if (abl == 2) return launch_gemm5_t<F16, EPI_QKV, 11, 12, 1, 3, 2>(a, n_pad, st);
        if (abl == 4) return launch_gemm5_t<F16, EPI_QKV, 11, 12, 1, 3, 4>(a, n_pad, st);
    }
    if (EPI == EPI_GENERIC && rb == 11 && (cb == 4 || cb == 8)) {
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
    // W-direct (fragment-ordered weights present, K a multiple of 128): the wide tiles, in the 1 x 4 consumer layout (F5HIP_GEMM5_WD=0: off, A/B)
    static const bool wd_on = !(getenv("F5HIP_GEMM5_WD") && atoi(getenv("F5HIP_GEMM5_WD")) == 0);
    if (wd_on && a.Wf && a.K % 128 == 0 && cb >= 8) {
        if (rb == 11 && cb == 8) return launch_gemm5_t<F16, EPI, 11, 8, 1, 6, 0, false, true>(a, n_pad, st);
        if (rb == 11 && cb == 12) return launch_gemm5_t<F16, EPI, 11, 12, 1, 6, 0, false, true>(a, n_pad, st);
        if (rb == 8 && cb == 8) return launch_gemm5_t<F16, EPI, 8, 8, 1, 6, 0, false, true>(a, n_pad, st);
        if (rb == 8 && cb == 12) return launch_gemm5_t<F16, EPI, 8, 12, 1, 6, 0, false, true>(a, n_pad, st);
    }
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
