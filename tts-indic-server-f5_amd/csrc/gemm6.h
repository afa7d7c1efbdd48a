// gemm6: 256 x 256 x 64 ping-pong MFMA GEMM for the BATCH-MODE shapes of the transformer-block GEMMs (one fp16 plane per operand).
//   C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate, the fused epilogues of gemm5.h (same functions: the arithmetic per output element, and the
//   k order of every dot product -- k-tile by k-tile, 32-deep half by half, v_mfma_f32_16x16x32_f16 -- are those of gemm5, so a row's result does
//   not depend on which of the two kernels computed it: the "batch of copies == single utterance" tests hold bit for bit).
//
// Why another GEMM (profiles/r03_gemm_batch_*): at M = 22 528 (C3's per-GPU share: 8 utterances x 2 CFG branches) gemm5's 176 x 128 / 176 x 192
// tiles deliver 0.27-0.32 of the MFMA roof although launch, fill and epilogue are amortised over 8 rounds.  What a CU can pull from L2 into
// LDS (86-95 GB/s measured, tools/fillrate.hip: 36-40 B / clk) against the matrix pipes' 4096 FLOP / clk / CU needs >= 105-115 FLOP per operand
// byte; a tile delivers BM BN / (BM + BN): 74 at 176 x 128, 92 at 176 x 192 -- fill-bound before anything else -- and 128 at 256 x 256.  A 256 x 256
// fp32 accumulator tile is exactly the register budget of eight waves (128 registers each), which leaves no room for dedicated loader waves:
// every wave stages AND multiplies, and the two waves of a SIMD alternate (MI355X_MICROARCH.md "Two waves per SIMD", the 8-phase structure of
// cdna_hip_programming.md section 5): waves 0-3 (rows 0-127 of the tile) and waves 4-7 (rows 128-255) run half a phase apart, so that while one
// group sits in its MFMA cluster (16 MFMAs = one 64 x 32 quadrant x K = 64, ~256 cycles of the SIMD's matrix pipe) the other issues its LDS-DMA
// pieces and fragment reads, and the pipe is handed over at every s_barrier.
//
// Geometry: wave w -> wr = w >> 2 (row half), wc = w & 3 (64-column quarter): 8 x 4 accumulator blocks of 16 x 16 (128 registers).
// LDS: two K-tile buffers of four 16 KiB HALF-TILES [A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255], 128-byte rows (64 k),
// XOR swizzle chunk ^ ((row >> 1) & 7) on the DMA's source side and on the fragment reads (conflict-free ds_read_b128, full-line DMA pieces).
// A half-tile is 16 LDS-DMA pieces of 1 KiB: two per wave.
// Schedule per K-tile kt (buffer kt & 1), four phases = the four quadrants (0,0) (0,1) (1,1) (1,0) of the wave's 128 x 64 output:
//   phase 0  stage A-lo(kt + 1)   read A frags of row blocks 0-3 (8) + W frags of column blocks 0-1 (4)      MFMAs (0,0)
//   phase 1  stage A-hi(kt + 1)   read W frags of column blocks 2-3 (4)                                       MFMAs (0,1)
//   phase 2  stage W-lo(kt + 2)   read A frags of row blocks 4-7 (8, over the first ones)                     MFMAs (1,1)
//   phase 3  stage W-hi(kt + 2)   no reads (both W fragment sets stay in registers); s_waitcnt vmcnt(4)       MFMAs (1,0)
// i.e. a phase is [stage, read, s_waitcnt lgkmcnt(0), s_barrier | 16 MFMAs, s_barrier].
//   WAR: the W half-tiles of buffer b are last read in phase 1, the A half-tiles in phase 2, every read is retired (lgkmcnt(0)) BEFORE the
//        barrier that ends its section, and the later group's section ends one barrier later: W-lo(kt + 2) goes into buffer b in phase 2,
//        one phase after its last read (legal with the reads retired before the barrier); A-lo(kt + 1) into the other buffer two phases after.
//   RAW: all four half-tiles of K-tile kt + 1 are first read in ITS phase 0.  Each wave waits for its own pieces in the load section of
//        phase 3 of K-tile kt -- vmcnt(4): the only younger DMAs are the four of W-lo / W-hi(kt + 2) -- which is followed by that section's
//        barrier for BOTH groups before the earlier group's next load section begins (the later group's wait sits one barrier later than
//        the earlier group's, and the earlier group's first read two barriers later).
// The most critical piece (A-hi(kt + 1), staged in phase 1, needed after phase 3) has two phases ~ 1000+ cycles to land.
#pragma once
#include "attn_common.h"
#include "gemm5.h"

// RBW = 16-row blocks per wave: 8 -> 256-row tiles; 6 -> tiles of 176 rows in a 192-row LDS image (the last row block of the second wave
// group multiplies the first 16 rows of the NEXT tile -- real, finite data -- and is never stored): 176 divides the row counts of this
// path (M_pad = 1408 per sequence: 22 528 = 128 x 176), so out / FF2 / QKV at the C3 share are 2 / 2 / 6 FULL rounds on the 256 CUs instead
// of 1.375 / 1.375 / 4.125 rounds of 256-row tiles that cost 2 / 2 / 5.
template <int RBW>
struct Gemm6Cfg {
    static_assert(RBW == 8 || RBW == 6, "8 = 256-row tiles, 6 = 176-row tiles");
    static constexpr int QR = RBW / 2;                      // row blocks per quadrant
    static constexpr int BM = RBW == 8 ? 256 : 176;         // rows a tile owns (its stride)
    static constexpr int BN = 256;
    static constexpr int AH = RBW * 2048, WH = 16384;       // half-tile bytes: RBW * 16 rows (A), 128 rows (W), 128-byte rows
    static constexpr int APIECES = RBW * 2;                 // 1 KiB pieces of an A half-tile
    static constexpr int BUF = 2 * AH + 2 * WH, LDS = 2 * BUF;
    using Q = Gemm5Cfg<QR, 16, 3>;   // the geometry of one quarter (QR row blocks) of the tile as the gemm5 epilogue functions see it (slab strides)
    static_assert(Q::SLAB <= LDS && Q::SLAB_T <= LDS, "the quarter-tile slabs alias the dead ring");
};

// one K-loop; SWAP as in gemm5 (true: the W fragment is the MFMA's A operand, a lane holds 4 consecutive FEATURES of a token)
// One 1 KiB LDS-DMA piece with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: the eight source addresses of a
// wave then cost 8 VGPRs instead of 16 (the 256-row kernels sit at the register limit).  M0 handling as attn_lds_dma16 (attn_common.h).
F5_DEVICE unsigned g6_lds_addr(const char* smem) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem; }   // (once per kernel: a generic -> LDS cast carries a null check)
F5_DEVICE void g6_dma16(const void* base, unsigned off, unsigned lds_dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(lds_dst);
    const unsigned long long b = (unsigned long long)base;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long bs = ((unsigned long long)bhi << 32) | blo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(bs), "s"(d) : "memory");
}

// DMA sources of a tile: half-tile h (0 A-lo, 1 A-hi, 2 W-lo, 3 W-hi), piece pp (0 / 1) = rows (wave + 8 pp) * 8 + (lane >> 3) of the half-tile.
// Rows past the matrices (a partial last slab) re-read the last valid row: finite data, never stored.
template <int RBW>
F5_DEVICE void g6_sources(const GemmArgs& p, int m0, int n0, int n_rows_w, int wave, int lane, unsigned (&src)[4][2]) {
#pragma unroll
    for (int h = 0; h < 4; h++)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            const int row = (wave + 8 * pp) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const bool isA = h < 2;
            const int grow = isA ? min(m0 + (h & 1) * (RBW * 16) + row, p.M - 1) : min(n0 + (h & 1) * 128 + row, n_rows_w - 1);
            src[h][pp] = ((unsigned)grow * (unsigned)(isA ? p.lda : p.ldw) + (unsigned)chunk * 8u) * 2u;   // byte offset from A[0] / W[0] (< 4 GiB: checked at launch)
        }
}
template <int RBW>
F5_DEVICE void g6_stage(const GemmArgs& p, const unsigned (&src)[4][2], unsigned lds0, int wave, int h, int kt) {   // half-tile h of K-tile kt -> buffer kt & 1
    using C = Gemm6Cfg<RBW>;
    const unsigned dst = lds0 + (kt & 1) * C::BUF + (h < 2 ? h * C::AH : 2 * C::AH + (h - 2) * C::WH) + wave * 1024;
    const void* base = h < 2 ? (const void*)p.A[0] : (const void*)p.W[0];
    g6_dma16(base, src[h][0] + (unsigned)kt * 128u, dst);
    if (C::APIECES == 16 || h >= 2 || wave < C::APIECES - 8) g6_dma16(base, src[h][1] + (unsigned)kt * 128u, dst + 8192);   // (an A half-tile has 16 or 12 pieces)
}
// prologue of a tile: K-tile 0 and the W halves of K-tile 1 (the persistent kernel issues it for the NEXT tile before the epilogue of the
// current one, so the first K-tiles land while the accumulators are being written out)
template <int RBW>
F5_DEVICE void g6_prologue(const GemmArgs& p, unsigned lds0, int m0, int n0, int n_rows_w, int wave, int lane) {
    using C = Gemm6Cfg<RBW>;
    const bool two = (p.K >> 6) > 1;
    // (one source pointer alive at a time: the accumulators of the tile being finished own the registers)
#pragma unroll
    for (int h = 0; h < 4; h++)
#pragma unroll
        for (int pp = 0; pp < 2; pp++) {
            if (C::APIECES < 16 && h < 2 && pp == 1 && wave >= C::APIECES - 8) continue;   // (an A half-tile has 16 or 12 pieces; wave-uniform)
            const int row = (wave + 8 * pp) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const bool isA = h < 2;
            const int grow = isA ? min(m0 + (h & 1) * (RBW * 16) + row, p.M - 1) : min(n0 + (h & 1) * 128 + row, n_rows_w - 1);
            const unsigned off = ((unsigned)grow * (unsigned)(isA ? p.lda : p.ldw) + (unsigned)chunk * 8u) * 2u;
            const void* base = isA ? (const void*)p.A[0] : (const void*)p.W[0];
            const unsigned dst = lds0 + (h < 2 ? h * C::AH : 2 * C::AH + (h - 2) * C::WH) + wave * 1024 + pp * 8192;
            g6_dma16(base, off, dst);                                   // K-tile 0 -> buffer 0
            if (!isA && two) g6_dma16(base, off + 128u, dst + C::BUF);  // the W halves of K-tile 1 -> buffer 1
            asm volatile("" ::: "memory");
        }
}

// STAGED: the prologue of this tile was issued earlier (g6_prologue, with other vector-memory traffic behind it: wait for everything)
template <bool F16, bool SWAP, int RBW>
F5_DEVICE void g6_kloop(const GemmArgs& p, char* smem, unsigned lds0, int m0, int n0, int n_rows_w, int wave, int lane, f32x4 (&acc)[RBW][4], bool staged) {
    using C = Gemm6Cfg<RBW>;
    constexpr int AH = C::AH, WH = C::WH, BUF = C::BUF, QR = C::QR;
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = p.K >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int off0 = fr * 128 + ((fq ^ (fr >> 1)) << 4);
#pragma unroll
    for (int i = 0; i < RBW; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    unsigned src[4][2];
    g6_sources<RBW>(p, m0, n0, n_rows_w, wave, lane, src);
    auto stage = [&](int h, int kt) { g6_stage<RBW>(p, src, lds0, wave, h, kt); };
    bf16x8 fa[2 * QR], fb[2][4];   // fa[kh * QR + i]: row block i of the current row half; fb[qn][kh * 2 + j]
    auto read_a = [&](int kt, int qm) {
        const char* b = smem + (kt & 1) * BUF + wr * AH + qm * (QR * 2048);
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int i = 0; i < QR; i++) fa[kh * QR + i] = *reinterpret_cast<const bf16x8*>(b + i * 2048 + (off0 ^ (kh << 6)));
    };
    auto read_b = [&](int kt, int qn) {
        const char* b = smem + (kt & 1) * BUF + 2 * AH + (wc >> 1) * WH + (wc & 1) * 8192 + qn * 4096;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int j = 0; j < 2; j++) fb[qn][kh * 2 + j] = *reinterpret_cast<const bf16x8*>(b + j * 2048 + (off0 ^ (kh << 6)));
    };
    auto load_end = [&]() {   // every fragment read of the section has returned, then the hand-over barrier
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_quadrant = [&](auto qm_t, auto qn_t) {
        constexpr int QM = decltype(qm_t)::value, QN = decltype(qn_t)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int i = 0; i < QR; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[QM * QR + i][QN * 2 + j] = SWAP ? mfma_16x16x32<F16>(fb[QN][kh * 2 + j], fa[kh * QR + i], acc[QM * QR + i][QN * 2 + j])
                                                        : mfma_16x16x32<F16>(fa[kh * QR + i], fb[QN][kh * 2 + j], acc[QM * QR + i][QN * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // one K-tile; NEXT1: K-tile kt + 1 exists (stage its A halves), NEXT2: kt + 2 exists (stage its W halves)
    auto ktile = [&](int kt, auto next1_t, auto next2_t) {
        constexpr bool NEXT1 = decltype(next1_t)::value, NEXT2 = decltype(next2_t)::value;
        if (NEXT1) stage(0, kt + 1);
        read_b(kt, 0);
        read_a(kt, 0);
        load_end();
        mfma_quadrant(I0{}, I0{});
        if (NEXT1) stage(1, kt + 1);
        read_b(kt, 1);
        load_end();
        mfma_quadrant(I0{}, I1{});
        if (NEXT2) stage(2, kt + 2);
        read_a(kt, 1);
        load_end();
        mfma_quadrant(I1{}, I1{});
        if (NEXT2) stage(3, kt + 2);
        if (NEXT2) attn_wait_vmcnt<4>(); else if (NEXT1) attn_wait_vmcnt<0>();
        load_end();
        mfma_quadrant(I1{}, I0{});
    };
    // prologue: K-tile 0 and the W halves of K-tile 1
    if (staged) {
        attn_wait_vmcnt<0>();
    } else {
        stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
        if (nk > 1) { stage(2, 1); stage(3, 1); attn_wait_vmcnt<4>(); } else attn_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }   // the later group: half a phase behind from here on
    using T_ = std::true_type;
    using F_ = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; kt++) ktile(kt, T_{}, T_{});
    if (kt + 1 < nk) { ktile(kt, T_{}, F_{}); kt++; }
    ktile(kt, F_{}, F_{});
    if (wr == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }   // (pairs with the later group's last barrier: the ring is dead)
}

// ---------------------------------------------------------------------------------------------------------------------------------
// generic epilogue straight from the accumulators (SWAP layout: a lane holds features 4 fq .. + 3 of column block j for token fr of row block
// i): no slab, no barrier -- all eight waves hold a 128 x 64 part of the tile, so all of them stream their own residual rows (64-byte
// segments, the next row block's in flight while the current one is finished).  The per-element arithmetic is g5_epi_value, gemm5's.
// (Through the slab, four quarters of [write | barrier | row phase | barrier], the epilogue took 13-17 us of a 39-43 us tile:
// profiles/r03_gemm6_stamps_slab_epilogue.txt.)
// n_blk = row blocks of this wave that belong to the tile (RBW, or one fewer for the second group of a 176-row tile)
template <int ACT, bool RES, bool OUTF, int OUTS, bool GUARD, int RBW>
F5_DEVICE void g6_direct_tail(const GemmArgs& p, f32x4 (&acc)[RBW][4], int m_w, int n_w, int n_blk, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 bv[4], mv[4];
    bool nok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = n_w + j * 16 + fq * 4;
        nok[j] = GUARD ? n < p.N : true;
        bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mv[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (p.bias && nok[j]) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.mul && nok[j]) mv[j] = *reinterpret_cast<const f32x4*>(p.mul + n);
    }
    f32x4 rs[RES ? 4 : 1], rn[RES ? 4 : 1];
    auto load_res = [&](int i, f32x4 (&dst)[RES ? 4 : 1]) {
        if (!RES) return;
        const int row = m_w + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            dst[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i < n_blk && (!GUARD || (nok[j] && row < p.M))) dst[j] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldres + n_w + j * 16 + fq * 4);
        }
    };
    load_res(0, rs);
#pragma unroll
    for (int i = 0; i < RBW; i++) {
        if (i >= n_blk) break;                                 // (wave-uniform)
        load_res(i + 1, rn);
        const int row = m_w + i * 16 + fr;
        int keep = 1;
        if (GUARD && p.row_keep && row < p.M) keep = p.row_keep[row];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n_w + j * 16 + fq * 4;
            const f32x4 v = g5_epi_value<ACT, RES>(acc[i][j], bv[j], mv[j], rs[RES ? j : 0], GUARD && !keep);
            if (!GUARD || (nok[j] && row < p.M)) {
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
            for (int j = 0; j < 4; j++) rs[j] = rn[j];
        }
    }
}

template <int ACT, bool GUARD, int RBW>
F5_DEVICE void g6_direct_variants(const GemmArgs& p, f32x4 (&acc)[RBW][4], int m_w, int n_w, int n_blk, int lane) {
    const bool res = p.res != nullptr, outf = p.out_f32 != nullptr, outs = p.out_hi != nullptr;
    // the (residual, fp32 out, 16-bit out) combinations in use on the path: the table of g5_generic_variants
    if (ACT == ACT_NONE) {
        if (res) {
            if (outf && outs) g6_direct_tail<ACT, true, true, 1, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
            else if (outf) g6_direct_tail<ACT, true, true, 0, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
            else g6_direct_tail<ACT, true, false, 1, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
        } else {
            if (outf && outs) g6_direct_tail<ACT, false, true, 1, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
            else if (outf) g6_direct_tail<ACT, false, true, 0, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
            else if (p.f16_out) g6_direct_tail<ACT, false, false, 2, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
            else g6_direct_tail<ACT, false, false, 1, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
        }
    } else {
        if (res) g6_direct_tail<ACT, true, true, 0, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
        else if (outs && !outf && p.f16_out) g6_direct_tail<ACT, false, false, 2, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
        else if (outs && !outf) g6_direct_tail<ACT, false, false, 1, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
        else if (outf && !outs) g6_direct_tail<ACT, false, true, 0, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
        else g6_direct_tail<ACT, false, true, 1, GUARD, RBW>(p, acc, m_w, n_w, n_blk, lane);
    }
}

template <int RBW>
F5_DEVICE void g6_direct_epilogue(const GemmArgs& p, f32x4 (&acc)[RBW][4], int m0, int n0, int m_w, int n_w, int n_blk, int lane) {
    const bool interior = m0 + Gemm6Cfg<RBW>::BM <= p.M && n0 + Gemm6Cfg<RBW>::BN <= p.N && !p.row_keep;   // workgroup-uniform
#define G6_ACT(A)                                                                         \
    if (interior) g6_direct_variants<A, false, RBW>(p, acc, m_w, n_w, n_blk, lane);       \
    else g6_direct_variants<A, true, RBW>(p, acc, m_w, n_w, n_blk, lane);
    switch (p.act) {
        case ACT_GELU_TANH: G6_ACT(ACT_GELU_TANH) break;
        case ACT_GELU_ERF: G6_ACT(ACT_GELU_ERF) break;
        case ACT_MISH: G6_ACT(ACT_MISH) break;
        case ACT_SILU: G6_ACT(ACT_SILU) break;
        default: G6_ACT(ACT_NONE) break;
    }
#undef G6_ACT
}

// Q / K tiles of the QKV projection straight from the accumulators: g5_qk_rows' arithmetic (bias, rotary on head 0 -- the first 64 columns
// of the Q and of the K block: only the wc = 0 waves of the tiles at n0 = 0 and n0 = D -- the softmax scale on q, saturated fp16), per lane
// 4 consecutive features of one token per (row block, column block).
template <int RBW>
F5_DEVICE void g6_qk_direct(const GemmArgs& p, f32x4 (&acc)[RBW][4], int n0, int m_w, int n_w, int n_blk, int m_end, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const int D = p.D, which = n0 / D;                         // 0 = q, 1 = k (a tile is all one kind: D % 256 == 0)
    const int ndw = n_w - which * D;                           // the wave's first column inside the q / k block
    const bool rot = ndw == 0;                                 // head 0 (wave-uniform)
    const float qs = which == 0 ? F5_Q_SCALE : 1.0f;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + n_w + j * 16 + fq * 4);
#pragma unroll
    for (int i = 0; i < RBW; i++) {
        if (i >= n_blk) break;                                 // (wave-uniform)
        const int row = m_w + i * 16 + fr;
        const bool rok = row < m_end;
        float2 cs[4], sn[4];
        if (rot) {
            const int pos = rok ? (p.row_pos ? p.row_pos[row] : row) : 0;   // (row_pos == null: per-row tables)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cs[j] = *reinterpret_cast<const float2*>(p.rope_cos + pos * 32 + j * 8 + fq * 2);
                sn[j] = *reinterpret_cast<const float2*>(p.rope_sin + pos * 32 + j * 8 + fq * 2);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x4 v = acc[i][j] + bv[j];
            float o[4];
            if (rot) {
                o[0] = __builtin_fmaf(v[0], cs[j].x, -__fmul_rn(v[1], sn[j].x)) * qs;
                o[1] = __builtin_fmaf(v[1], cs[j].x, __fmul_rn(v[0], sn[j].x)) * qs;
                o[2] = __builtin_fmaf(v[2], cs[j].y, -__fmul_rn(v[3], sn[j].y)) * qs;
                o[3] = __builtin_fmaf(v[3], cs[j].y, __fmul_rn(v[2], sn[j].y)) * qs;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = v[e] * qs;
            }
            if (rok) store_f16x4(p.qk + (size_t)row * (2 * D) + which * D + ndw + j * 16 + fq * 4, o);
        }
    }
}

// PERSISTENT grid: workgroup b works through tiles b, b + gridDim.x, ... (the grid is one workgroup per CU, or the tile count if smaller).
// When the next tile's epilogue-free start is known -- the current tile writes its results straight from the accumulators (generic and Q / K
// tiles; V tiles need the LDS for their transposed slab) -- the NEXT tile's prologue (K-tile 0 + the W halves of K-tile 1: 12 LDS-DMA pieces
// per wave) is issued BEFORE the current epilogue: the ring is dead, the accumulators are in registers, and the ~2 us the first K-tile takes
// to land from beyond L2 pass behind 5-10 us of epilogue instead of in front of the next k-loop.
template <bool F16, int EPI, int RBW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm6_kernel(const GemmArgs p, const int tiles_n, const int n_rows_w, const int n_tiles) {
    using C = Gemm6Cfg<RBW>;
    using Q = typename C::Q;
    constexpr int QR = C::QR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = g6_lds_addr(smem);
    // diagnostics (run time: p.stamps != null, tools/gemm6_stamps.py): s_memrealtime (100 MHz) of waves 0 and 4 at [0] start, [1] k-loop done,
    // [2..5] epilogue done (quarter s of a V tile), [6] stores drained, [7] shader cycles of the k-loop -- of the workgroup's FIRST tile
    // (kept as a wave-uniform flag and re-derived at every use: a per-lane pointer alive across the k-loop costs the 256-row kernels registers
    // they do not have)
    bool stamping = p.stamps != nullptr;
#define G6_STAMP_PTR() (p.stamps + ((size_t)blockIdx.x * 2 + (tid >> 8)) * 8)
#define G6_STAMP(EXPR) if (stamping && (tid == 0 || tid == 256)) { unsigned long long* stamp = G6_STAMP_PTR(); EXPR; }
    bool staged = false;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tile = gemm5_tile_of_block(t, n_tiles);
        const int m0 = (tile / tiles_n) * C::BM, n0 = (tile % tiles_n) * C::BN;
        // all-V tiles of the QKV projection keep the token on the accumulator registers (their output is [feature][token]); D % 256 == 0, so a
        // tile is all Q, all K or all V
        const bool swap = !(EPI == EPI_QKV && n0 >= 2 * p.D);
        G6_STAMP(stamp[0] = __builtin_amdgcn_s_memrealtime(); stamp[7] = __builtin_amdgcn_s_memtime())
        f32x4 acc[RBW][4];
        if (swap) g6_kloop<F16, true, RBW>(p, smem, lds0, m0, n0, n_rows_w, wave, lane, acc, staged);
        else g6_kloop<F16, false, RBW>(p, smem, lds0, m0, n0, n_rows_w, wave, lane, acc, staged);
        G6_STAMP(stamp[7] = __builtin_amdgcn_s_memtime() - stamp[7]; stamp[1] = __builtin_amdgcn_s_memrealtime())
        // the next tile's first K-tiles, if this tile's epilogue leaves the LDS alone
        staged = false;
        if (swap && t + (int)gridDim.x < n_tiles) {                 // (workgroup-uniform)
            const int tn = gemm5_tile_of_block(t + gridDim.x, n_tiles);
            g6_prologue<RBW>(p, lds0, (tn / tiles_n) * C::BM, (tn % tiles_n) * C::BN, n_rows_w, wave, lane);
            staged = true;
        }
        // rows of the LDS image this tile owns: all of them, or 176 of the 192 (the last row block of the second wave group is the next tile's)
        const int m_end = min(p.M, m0 + C::BM);
        const int n_blk = min(RBW, (C::BM - wr * RBW * 16) / 16);
        if (EPI == EPI_GENERIC || swap) {
            if constexpr (EPI == EPI_GENERIC) g6_direct_epilogue<RBW>(p, acc, m0, n0, m0 + wr * RBW * 16, n0 + wc * 64, n_blk, lane);
            else g6_qk_direct<RBW>(p, acc, n0, m0 + wr * RBW * 16, n0 + wc * 64, n_blk, m_end, lane);   // Q / K tile
            G6_STAMP(stamp[2] = stamp[3] = stamp[4] = stamp[5] = __builtin_amdgcn_s_memrealtime())
        } else {
            // V tiles ([feature][token] output): four quarters of QR row blocks through one transposed slab that aliases the dead ring; the row
            // phase is gemm5's, which bounds its stores by p.M: it sees the tile's own end instead
            GemmArgs pe = p;
            pe.M = m_end;
            float* slab = reinterpret_cast<float*>(smem);
            auto write_cols = [&](auto half_t) {   // !SWAP layout -> transposed slab [256 features][16 QR tokens], + bias
                constexpr int H = decltype(half_t)::value;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float b1 = p.bias[n0 + wc * 64 + j * 16 + fr];
#pragma unroll
                    for (int i = 0; i < QR; i++)
                        *reinterpret_cast<f32x4*>(slab + (wc * 64 + j * 16 + fr) * Q::SLDT + i * 16 + fq * 4) = acc[H * QR + i][j] + (f32x4){b1, b1, b1, b1};
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
#pragma unroll 1
            for (int s4 = 0; s4 < 4; s4++) {
                const int mq = m0 + s4 * (QR * 16);
                if (mq >= m_end) break;                             // (workgroup-uniform: quarters past the last row)
                const bool mine = wr == (s4 >> 1);
                if (mine) { if (s4 & 1) write_cols(I1{}); else write_cols(I0{}); }
                __syncthreads();
                g5_v_rows<QR, 16, 3>(pe, slab, mq, n0, 0, wave, lane);
                __syncthreads();                                    // the slab is free for the next quarter (and, after the last one, for the ring)
                G6_STAMP(stamp[2 + s4] = __builtin_amdgcn_s_memrealtime())
            }
        }
        if (stamping) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            G6_STAMP(stamp[6] = __builtin_amdgcn_s_memrealtime())
            stamping = false;                                       // (first tile only)
        }
    }
#undef G6_STAMP
#undef G6_STAMP_PTR
}

template <bool F16, int EPI, int RBW>
static hipError_t launch_gemm6_t(const GemmArgs& a, int n_pad, hipStream_t st) {
    using C = Gemm6Cfg<RBW>;
    if (n_pad % C::BN || a.K % 64 || a.K < 64 || (EPI == EPI_QKV && a.D % C::BN)) return hipErrorInvalidValue;
    if ((unsigned long long)a.M * a.lda * 2ull >= (1ull << 32) || (unsigned long long)n_pad * a.ldw * 2ull >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit DMA offsets
    static unsigned attr_mask = 0;
    if (hipError_t e = f5_set_lds_attr(reinterpret_cast<const void*>(&gemm6_kernel<F16, EPI, RBW>), C::LDS, attr_mask); e != hipSuccess) return e;
    const int tiles_m = (a.M + C::BM - 1) / C::BM, tiles_n = n_pad / C::BN, n_tiles = tiles_m * tiles_n;
    static int n_cu = 0;                                            // persistent grid: one workgroup per CU (F5HIP_GEMM6_PERSIST=0: one per tile, A/B)
    if (!n_cu) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu = (getenv("F5HIP_GEMM6_PERSIST") && atoi(getenv("F5HIP_GEMM6_PERSIST")) == 0) ? (1 << 30) : cus;
    }
    hipLaunchKernelGGL((gemm6_kernel<F16, EPI, RBW>), dim3(n_tiles < n_cu ? n_tiles : n_cu), dim3(512), C::LDS, st, a, tiles_n, n_pad, n_tiles);
    return hipGetLastError();
}
