// RECORD of a measured-and-rejected variant (round 3): attn3.h with a "Q2" step -- 64 queries per wave so that every K / V^T fragment read from LDS
// feeds two MFMAs, the two query blocks pipelined against each other inside the wave (MFMAs as asm with explicit VGPR / AGPR files, exp2 / converts
// of the other block hand-placed between them behind sched_barrier fences, one fragment set refilled in place).  Correct, slower than attn3's
// 32-query waves at two per SIMD in every shape; numbers and the reasons in profiles/r03_attn_ablate.txt.  Not included by any translation unit.
// attn3: flash-style attention forward.  S^T = K Q^T, so a query is a lane and its row statistics are in-register; exp(S) is directly the
// B operand of O^T += V^T P^T.  A wave owns QB blocks of 32 queries; a workgroup = NW waves = one query tile of 32 QB NW queries; K / V^T
// tiles of 64 keys stream through an LDS ring by LDS-DMA.
//
// History of the inner loop (MI355X, C2 shape N = 1404 x 16 heads x 2, tools/attn_stamps.py / attn_bench.py, profiles/r02_attn_bench.txt):
//   attn2   ~3800 cycles per KV tile and SIMD against ~1000 of MFMA and ~2200 of VALU: the per-tile barrier keeps the two waves of a SIMD
//           in lockstep, so matrix and vector phases add up instead of overlapping;
//   attn3a  each wave carries two score tiles, S_next = K_{j+1} Q^T issued before the softmax of S_cur: 2940 per tile, 45.7 us -- the
//           stamps showed the overlap never happened: a wave is in-order, the 8 score MFMAs went out back to back (the wave sits in MFMA
//           issue for 256+ cycles with its VALU idle), then the max chain ran with the matrix pipe idle, then exp2 + PV;
//   now     (1) fixed-offset softmax (below): a third fewer VALU issue cycles, no data-dependent branch in the loop;
//           (2) half-tile steps (32 keys = one 32 x 32 score block per query block), each ONE straight-line scheduling region in which
//               sched_group_barrier spreads the MFMAs one : two exp2 : four plain VALU;
//           (3) V^T stored in vt_col order (common.h): a PV fragment is one ds_read_b128, no register shuffle; fragment addresses from
//               one lane constant by XOR;
//           (4) fragments prefetched one half tile ahead (no MFMA waits on LDS latency);
//           (5) XCD-aware workgroup numbering: the query tiles of one (sequence, head) share an L2;
//           (6) QB = 2: 64 queries per wave, one wave per SIMD with the 512-register budget -- every K / V^T fragment read from LDS feeds two
//               MFMAs, there is no partner wave to share the SIMD's issue slots with, and NW = 4 puts exactly one wave on each SIMD
//               (with 6 waves of 32 queries two SIMDs carried two waves, two carried one, and every tile ended at a barrier).
#pragma once
#include <type_traits>
#include "attn_common.h"

// NW = waves per workgroup (4, 6 or 8), QB = 32-query blocks per wave (1 or 2), NST = ring stages of 16 KiB.
// SEG2: the keys are two row ranges (AttnArgs::seq_kv_row0 / seq_kv2_*): tile kt covers 64 rows of the first range while kt < nkt1, of the
// second after it; the last tile of EACH range is masked.  SEG2 = false is the single-range kernel of the DiT / UNetT path.
//
// Operands are FP16 (round 3): q (pre-scaled), k, v^T arrive as saturated fp16 from the QKV epilogue and exp(S) is rounded to fp16 for the
// P V product -- v_mfma_f32_32x32x16_f16 issues at the bf16 rate and carries 11 significand bits instead of 8.  tools/attn_ladder.py: on the
// reference's tiny UNetT / MMDiT CFM.sample fixtures bf16 attention operands alone cost 9.7e-4 / 1.08e-3 rms (above north_star's 1e-3),
// fp16 ones 1.2e-4 / 1.4e-4; they were also what made a request's result depend on the batch it ran in.
//
// Softmax with a FIXED per-query offset folded into the score MFMAs.  q arrives scaled by log2(e) / 8 (F5_Q_SCALE), so a score is already
// a base-2 exponent.  Floating point is scale-invariant: the running maximum of online softmax only has to keep exp2 inside the range of
// the P operand, it does not have to be the maximum.  So the offset of a query is fixed BEFORE the loop: A3_OFF_MARGIN above its maximum score
// over 32 SAMPLE keys (16 spread over the whole key range, 16 around the query block's own position -- neighbouring frames are where a
// speech model's attention peaks) and stays put: the first MFMA of every score block takes C = splat(-off) instead of 0, the accumulator
// leaves the matrix pipe as s - off, and exp2 applies to it directly -- per score one exp2, one add, half a convert; no max, no subtract,
// no scale, no rescaling of O, no branch.  fp16 normals span 2^-14 .. 2^16: with the offset 2 binades above the sample maximum the sampled
// keys have P <= 1/4, the true row maximum has P >= 1/4 (samples are keys), and every P within 2^-12 of it is still a normal number.
// Whether every probability stayed below fp16's largest finite value is decided ONCE, after the loop, from the row sums themselves (they are
// accumulated by the matrix pipe from the fp16 probabilities: an overflowed one makes its row sum inf; the inf it put into O is discarded with
// it): if one did not -- a logit more than 18 binades = 12.5 nats above the query's maximum over its sample -- the WHOLE workgroup redoes its
// tile with a plain running-maximum loop, correct
// for any input and exercised by tests/test_gpu_ops.py (k_gain cases).  Because every variant of the kernel (and both key halves of a BAL
// block) derives the offset from the same sample, variants differ only in the association of fp32 sums.
// (The running-maximum formulation in the hot loop measured 45.7 us at C2 against 41.4 us for the fixed offset, before any of the
// scheduling work.)
#define A3_OFF_MARGIN 2.0f
#ifndef A3_ABL
#define A3_ABL 0   // timing ablations of the hot loop (WRONG results; tools/attn_ablate.sh): bit 0 no ring step, bit 1 exp2 -> v_mul, bit 2 no LDS reads, bit 3 no P V
#endif
#define A3_P_LIMIT 0x1p15f
// BAL (NW = 8, QB = 1, 192 queries per workgroup): the SIMD-balanced form of the 6-block tile.  Eight waves put two on every SIMD; waves 0-3 own a
// query block each over all keys, waves 4 / 5 own blocks 4 / 5 over keys 0-31 of every tile and waves 6 / 7 the SAME blocks over keys 32-63,
// so every SIMD carries three half tiles per tile (with six whole-block waves two SIMDs carry four and two carry two, and the per-tile barrier
// makes the four the pace).  The two halves of blocks 4 / 5 keep their own offset, O and l and are merged once at the end through LDS.
template <int NW, bool SEG2 = false, bool STAMPS = false, int NST = 5, int QB = 1, bool BAL = false>
static __global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((QB == 2 && NST >= 9) ? 1 : 2, (QB == 2 && NST >= 9) ? 1 : 2))) void attn3_fwd_kernel(const AttnArgs p) {
    constexpr int STAGE = 16384;   // one 64-key tile: 8 KiB of K rows + 8 KiB of V^T rows
    constexpr int P_HI = (16 + NW - 1) / NW, P_LO = 16 / NW;   // 1 KiB pieces of a KV tile per wave (pieces w, w + NW, ...)
    static_assert(!BAL || (NW == 8 && QB == 1), "BAL is the 8-wave, 6-block form");
    constexpr int QT = BAL ? 192 : 32 * QB * NW;                // queries per workgroup
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    // Workgroup -> (query tile, head, sequence), XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, which would
    // put the query tiles of one (sequence, head) -- the workgroups that read the same K / V^T rows -- on 8 different L2s, each of them pulling
    // every row from beyond L2.  Re-number so that an XCD gets a contiguous run of (sequence, head, tile) triples, tile fastest.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z, per = total >> 3;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        if (lin < per * 8) {
            const unsigned v = (lin & 7) * per + (lin >> 3);
            bx = (int)(v % gx); by = (int)((v / gx) % gy); bz = (int)(v / (gx * gy));
        }
    }
    const int seq = bz, head = by;
    const int len = p.seq_len[seq], kvlen = p.seq_kvlen[seq], row0 = p.seq_row0[seq];
    const int kv_row0 = SEG2 ? p.seq_kv_row0[seq] : row0, kv2_row0 = SEG2 ? p.seq_kv2_row0[seq] : 0, kv2_len = SEG2 ? p.seq_kv2_len[seq] : 0;
    const int nkt1 = (kvlen + 63) >> 6;
    const int q0 = bx * QT;
    if (q0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int D = p.D;
    // BAL roles: 0 = whole tiles, 1 = keys 0-31 of every tile, 2 = keys 32-63; qblk = the 32-query block of the wave
    const int role = !BAL ? 0 : wave < 4 ? 0 : wave < 6 ? 1 : 2;
    const int qblk = !BAL ? wave * QB : wave < 6 ? wave : wave - 2;

    // queries of this wave (rows beyond the sequence stay inside its 128-row padding or the next sequence: finite data, never stored).
    // q0 + QT - 1 can exceed the padded rows of the LAST sequence only by < 256 rows: the workspace has that much slack.
    f16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const __bf16* qrow = p.qk + (size_t)(row0 + q0 + (qblk + qb) * 32 + fr) * (2 * D) + head * 64 + fh * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[qb][s] = *reinterpret_cast<const f16x8*>(qrow + s * 16);
    }
    // The 32 SAMPLE keys that fix the softmax offsets of this wave's queries (below): 16 spread evenly over the (first) key range and 16 at
    // every second position (every fourth at QB = 2) of the wave's own query block, clamped into the range -- all of them valid keys.  Their K
    // rows come straight from global memory in the fragment layout of a score MFMA's A operand (lane = key fr, 8 features per k-step).
    f16x8 ksamp[4];
    {
        const int sidx = fr < 16 ? (fr * kvlen) >> 4 : min(q0 + qblk * 32 + (fr - 16) * 2 * QB, kvlen - 1);
        const __bf16* krow = p.qk + (size_t)(kv_row0 + sidx) * (2 * D) + D + head * 64 + fh * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) ksamp[s] = *reinterpret_cast<const f16x8*>(krow + s * 16);
    }
    // Retire the Q / sample loads BEFORE the first LDS-DMA is issued: with a DMA in flight hipcc can only wait vmcnt(0) for an
    // ordinary VGPR load, and it would put that wait inside the KV loop, draining the ring every tile.
#pragma unroll
    for (int qb = 0; qb < QB; qb++) asm volatile("" ::"v"(qf[qb][0]), "v"(qf[qb][1]), "v"(qf[qb][2]), "v"(qf[qb][3]) : "memory");
    asm volatile("" ::"v"(ksamp[0]), "v"(ksamp[1]), "v"(ksamp[2]), "v"(ksamp[3]) : "memory");

    // LDS-DMA: a KV tile is 8 K pieces + 8 V^T pieces of 1 KiB (8 rows x 128 B); wave w moves pieces w, w + NW, ... (0-7 = K, 8-15 = V^T).
    // Physical 16-B slot (lane & 7) of row r holds logical chunk (lane & 7) ^ ((r >> 1) & 7)  (same swizzle as the fragment reads).
    const int mine = (16 - wave + NW - 1) / NW;
    const char* src[P_HI];
    size_t step[P_HI];
#pragma unroll
    for (int j = 0; j < P_HI; j++) {
        const int pc = wave + NW * j;
        const int prow = (pc & 7) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((prow >> 1) & 7);
        const bool isK = pc < 8;
        src[j] = isK ? reinterpret_cast<const char*>(p.qk + (size_t)(kv_row0 + prow) * (2 * D) + D + head * 64 + chunk * 8)
                     : reinterpret_cast<const char*>(p.vt + (size_t)(head * 64 + prow) * p.ldvt + kv_row0 + chunk * 8);
        step[j] = isK ? (size_t)64 * (2 * D) * 2 : (size_t)64 * 2;   // bytes per KV tile
    }
    auto issue_tile = [&](int kt) {
        char* dst = smem + (kt % NST) * STAGE + wave * 1024;
        // tile kt starts (in rows, relative to kv_row0) at 64 kt in the first range, at kv2_row0 - kv_row0 + 64 (kt - nkt1) in the second
        const long long rel = (!SEG2 || kt < nkt1) ? (long long)kt * 64 : (long long)(kv2_row0 - kv_row0) + (long long)(kt - nkt1) * 64;
#pragma unroll
        for (int j = 0; j < P_HI; j++)
            if (j < P_LO || wave + NW * j < 16)
                attn_lds_dma16(src[j] + rel * (long long)(step[j] / 64), dst + j * NW * 1024);   // (asm: keeps hipcc's lgkmcnt waits exact, see attn_common.h)
    };
    // this wave's pieces of a tile have landed when at most `newer` younger tiles of its own are in flight
    auto wait_landed = [&](int newer) {
        switch (newer * mine) {   // (wave-uniform; vmcnt takes an immediate)
#define A3_W(N) case N: attn_wait_vmcnt<N>(); break;
            A3_W(0) A3_W(1) A3_W(2) A3_W(3) A3_W(4) A3_W(5) A3_W(6) A3_W(7) A3_W(8) A3_W(9) A3_W(10) A3_W(11) A3_W(12) A3_W(13) A3_W(14) A3_W(15) A3_W(16)
            A3_W(17) A3_W(18) A3_W(19) A3_W(20) A3_W(21) A3_W(22) A3_W(23) A3_W(24) A3_W(25) A3_W(26) A3_W(27) A3_W(28) A3_W(29) A3_W(30) A3_W(31) A3_W(32)
#undef A3_W
            default: attn_wait_vmcnt<0>(); break;
        }
    };
    static_assert((NST - 2) * P_HI <= 32, "wait_landed covers 32 outstanding pieces");
    // the same wait with a compile-time tile count: the steady state of the ring.  (The switch above compiles to a chain of ~27 scalar
    // compare / branch pairs; tools/attn_stamps.py showed 240-500 clocks per KV tile between the end of a tile and the end of its wait with
    // every piece long landed -- the hot loop takes this path, the switch serves the first and last tiles.)
    auto wait_landed_steady = [&](auto t_) {
        constexpr int T = decltype(t_)::value;
        if constexpr (P_HI == P_LO) attn_wait_vmcnt<T * P_HI>();
        else if (mine == P_HI) attn_wait_vmcnt<T * P_HI>();
        else attn_wait_vmcnt<T * P_LO>();
    };

    f32x16 oacc[QB][2];
    f32x16 negm[QB];                   // C operand of the first score MFMA of a block = splat(-offset); zero for the first block, fixed after it
    float lrun[QB];
    // Row sums through the matrix pipe (round 3: the loop is bound by VECTOR issue at head dimension 64 -- 32 v_exp, ~75 other VALU per tile and
    // wave against 16 MFMAs -- and the sixteen v_add per half tile were its largest item after the exponentials).  One v_mfma_f32_16x16x32_f16
    // per 16-key k-step with a constant 0 / 1 A operand: seen as a 16 x 16 x 32 B operand, the P^T fragment of lane (fr, fh) is column fr & 15,
    // k-group 2 fh + (fr >> 4), i.e. k-groups {0, 2} carry query fr & 15 and {1, 3} query 16 + (fr & 15); row 0 of A is ones on k-groups {0, 2},
    // row 1 on {1, 3}, the rest zero: D[0][n] and D[1][n] accumulate the row sums of queries n and 16 + n (lanes 0-15, registers 0 and 1).  The
    // sums are those of the fp16-ROUNDED probabilities -- exactly what the P V product multiplies -- and a probability that overflowed fp16 makes
    // its row sum inf, which is the range check (no v_max in the loop either).
    f32x4 lacc[QB];
    f16x8 lones;
    {
        const int m16 = lane & 15, kq = lane >> 4;
        const _Float16 one = ((m16 == 0 && (kq & 1) == 0) || (m16 == 1 && (kq & 1) == 1)) ? (_Float16)1.0f : (_Float16)0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) lones[j] = one;
    }
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        lrun[qb] = 0.0f;
        lacc[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 16; g++) { oacc[qb][0][g] = 0.0f; oacc[qb][1][g] = 0.0f; negm[qb][g] = 0.0f; }
    }

    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0, st_t0 = 0, st_pro = 0;
    unsigned long long st_r0 = 0;
    if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory"); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    const bool st_on = STAMPS && p.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0;
#define A3_STAMP(I)                                                                                  \
    if (STAMPS && st_on) {                                                                           \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if ((I) >= 0) st_acc[(I) < 0 ? 0 : (I)] += t_ - st_prev;                                     \
        st_prev = t_;                                                                                \
    }
    const int nkt = nkt1 + (SEG2 ? (kv2_len + 63) >> 6 : 0);
#pragma unroll
    for (int t = 0; t < (NST >= 9 ? NST : NST - 1); t++)
        if (t < nkt) issue_tile(t);

    // The unit of the inner loop is a HALF tile: 32 keys = one 32 x 32 score block per query block (16 registers; two whole 64-key score
    // tiles in flight cost 64 and pushed the kernel into spills and accumulator copies).
    // Fragment addresses: lds_off128(h * 32 + fr, 2 sI + fh) = fr * 128 + h * 4096 + (((2 sI) ^ (fh ^ sw)) << 4) with sw = (fr >> 1) & 7 -- ONE lane
    // constant, the k-step is an XOR on address bits 5-6, the key half an immediate offset.  The V^T fragments (feature rows, 16-key k-steps in
    // vt_col order) have the same form.
    const unsigned k_lane = (unsigned)(fr * 128 + ((fh ^ ((fr >> 1) & 7)) << 4));
    const unsigned v_lane = k_lane + 8192u;
    auto stage_of = [&](int kt) { return (unsigned)((kt % NST) * STAGE); };
    auto qk_read = [&](f16x8 (&kf)[4], int kt, int h) {
        const unsigned kb = stage_of(kt) + k_lane;
#pragma unroll
        for (int sI = 0; sI < 4; sI++) kf[sI] = *reinterpret_cast<const f16x8*>(smem + (kb ^ (unsigned)(sI << 5)) + h * 4096);
    };
    // S^T block = K_half Q^T - offset (four chained MFMAs; C of the first = negm)
    auto qk_mfma = [&](f32x16& s, const f16x8 (&kf)[4], int qb) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[qb][0], negm[qb], 0, 0, 0);
#pragma unroll
        for (int sI = 1; sI < 4; sI++) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sI], qf[qb][sI], s, 0, 0, 0);
    };
    // V^T fragments of k-step s2 of key half h (16 keys, stored in vt_col order: a lane's 8 keys are contiguous): one 16-byte read per 32 features
    auto v_read = [&](f16x8 (&vf)[2], unsigned vb, int h, int s2) {
#pragma unroll
        for (int dt = 0; dt < 2; dt++) vf[dt] = *reinterpret_cast<const f16x8*>(smem + (vb ^ (unsigned)((h * 2 + s2) << 5)) + dt * 4096);
    };
    // key-padding mask of half h of tile kt (only the last tile of a key range is partial)
    auto mask_half = [&](f32x16& s, int kt, int h) {
        const int valid = ((!SEG2 || kt < nkt1) ? kvlen - kt * 64 : kv2_len - (kt - nkt1) * 64) - h * 32;   // (a tile never straddles the two ranges)
        if (valid < 32) {
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int key = (g & 3) + 8 * (g >> 2) + 4 * fh;
                if (key >= valid) s[g] = -1e30f;
            }
        }
    };
    auto row_max = [&](const f32x16& s) {
        float m = s[0];
#pragma unroll
        for (int g = 1; g < 16; g++) m = fmaxf(m, s[g]);
        return fmaxf(m, __shfl_xor(m, 32, 64));
    };
    // ring step kt: tile kt + 1 landed (its K feeds the next score block); every wave is past tile kt - 1, so stage (kt - 1) % NST is free.
    // PAIR (the 9-stage ring of a lone workgroup): one step per TWO tiles -- at even kt tiles kt + 1 and kt + 2 are made visible and the
    // stages of tiles kt - 2 and kt - 1 refilled; half the barriers (38 % of the wave cycles at C2 were parked at s_waitcnt / s_barrier:
    // profiles/r02_pmc_attention_sq.txt).  The prologue fills the whole ring and makes tiles 0-2 visible.
    constexpr bool PAIR = NST >= 9;
    auto ring_step = [&](int kt) {
        if (PAIR) {
            if ((kt & 1) || kt == 0) return;
            // issued so far: tiles up to kt - 3 + NST (capped at nkt - 1); tiles kt + 1, kt + 2 must have landed
            if (kt - 3 + NST <= nkt - 1) wait_landed_steady(std::integral_constant<int, NST - 5>{});
            else wait_landed(max(0, nkt - 1 - (kt + 2)));
            A3_STAMP(5);
            __builtin_amdgcn_s_barrier();
            A3_STAMP(4);
            if (kt - 2 + NST < nkt) issue_tile(kt - 2 + NST);
            if (kt - 1 + NST < nkt) issue_tile(kt - 1 + NST);
            return;
        }
        if (nkt - 2 - kt >= NST - 3) wait_landed_steady(std::integral_constant<int, NST - 3>{});   // tiles kt + 2 .. kt + NST - 2 may stay in flight
        else wait_landed(nkt - 2 - kt);
        A3_STAMP(5);
        __builtin_amdgcn_s_barrier();
        A3_STAMP(4);
        if (kt + NST - 1 < nkt) issue_tile(kt + NST - 1);
    };
    // ---- one half tile of the FAST loop: straight-line, no branch on the data ------------------------------------------------------------
    // ONE scheduling region: per query block the four score MFMAs of the NEXT half and the four O^T += V^T P^T MFMAs of THIS half, spread one
    // MFMA : two exp2 : a few plain VALU (an in-order wave that issues MFMAs back to back sits in MFMA issue with its VALU idle, and one
    // that runs its VALU in one block leaves the matrix pipe idle); exp2 / row sum / convert of this half (scores are s - off: exp2 directly).
    // The fragments a half tile multiplies were read from LDS during the PREVIOUS half tile (two register sets that swap roles), so no MFMA
    // waits on LDS latency; ring step kt (tile kt + 1 landed and visible) therefore comes at the top of tile kt, before its first half
    // prefetches K rows of tile kt + 1.
    // H: which half of tile kt `sc` holds.  The block produced here is (kt, 1) for H = 0 and (kt + 1, 0) for H = 1.
    // NEXT_TILE: tile kt + 1 exists.  MASK: the score block produced here belongs to a possibly partial tile.
    struct Frags { f16x8 k[4]; f16x8 v[2][2]; };
    auto fast_half = [&](f32x16 (&sc)[QB], int kt, Frags& use, Frags& fill, auto h_t, auto next_tile_t, auto mask_t) {
        constexpr int H = decltype(h_t)::value;
        constexpr bool NEXT_TILE = decltype(next_tile_t)::value, MASK = decltype(mask_t)::value;
        constexpr bool HAS_NEXT = H == 0 || NEXT_TILE;
        if (H == 0) {
            if (!(A3_ABL & 1)) ring_step(kt);
            A3_STAMP(0);
        }
        // prefetch for the following half tile: H = 0 -> (kt, 1) multiplies K(kt + 1, rows 0-31) and V(kt, keys 32-63);
        //                                      H = 1 -> (kt + 1, 0) multiplies K(kt + 1, rows 32-63) and V(kt + 1, keys 0-31)
        if (false) {
        } else if (A3_ABL & 4) {
            fill = use;
        } else {
        if (NEXT_TILE) qk_read(fill.k, kt + 1, H == 0 ? 0 : 1);
        if (H == 0) {
            const unsigned vb = stage_of(kt) + v_lane;
            v_read(fill.v[0], vb, 1, 0);
            v_read(fill.v[1], vb, 1, 1);
        } else if (NEXT_TILE) {
            const unsigned vb = stage_of(kt + 1) + v_lane;
            v_read(fill.v[0], vb, 0, 0);
            v_read(fill.v[1], vb, 0, 1);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            f32x16 acc;
            if (HAS_NEXT) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(use.k[0], qf[qb][0], negm[qb], 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(use.k[1], qf[qb][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(use.k[2], qf[qb][2], acc, 0, 0, 0);
            }
            float pe[16];
            f16x8 pf[2];
#pragma unroll
            for (int g = 0; g < 16; g++) pe[g] = (A3_ABL & 2) ? sc[qb][g] * 0.001f : __builtin_amdgcn_exp2f(sc[qb][g]);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
                for (int j = 0; j < 8; j++) pf[s2][j] = (_Float16)pe[8 * s2 + j];
                if (A3_ABL & 8) { asm volatile("" ::"v"(pf[s2])); continue; }
#pragma unroll
                for (int dt = 0; dt < 2; dt++) oacc[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use.v[s2][dt], pf[s2], oacc[qb][dt], 0, 0, 0);
                lacc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lones, pf[s2], lacc[qb], 0, 0, 0);   // row sums (see lacc)
            }
            if (HAS_NEXT) sc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(use.k[3], qf[qb][3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < QB * (HAS_NEXT ? 10 : 6); i++) {   // (8 + 2 row-sum MFMAs per query block; the VALU left: 16 v_exp, 8 converts, a few moves)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        if (HAS_NEXT && (SEG2 || MASK)) {
#pragma unroll
            for (int qb = 0; qb < QB; qb++) mask_half(sc[qb], H == 0 ? kt : kt + 1, H == 0 ? 1 : 0);
        }
        A3_STAMP(H == 0 ? 1 : 2);
    };

    // BAL, roles 1 / 2: one half tile per tile -- consumes block (kt, H), produces block (kt + 1, H); fragments read at the top of the step
    // (no prefetch across steps: K rows of tile kt + 2 are not visible yet; the whole-block wave on the same SIMD covers the latency)
    auto half_only = [&](f32x16& sc, int kt, auto h_t, auto next_tile_t) {
        constexpr int H = decltype(h_t)::value;
        constexpr bool NEXT = decltype(next_tile_t)::value;
        ring_step(kt);
        f16x8 kf[4], vf[2][2], pf[2];
        if (NEXT) qk_read(kf, kt + 1, H);
        const unsigned vb = stage_of(kt) + v_lane;
        v_read(vf[0], vb, H, 0);
        v_read(vf[1], vb, H, 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
        if (NEXT) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0][0], negm[0], 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[0][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[2], qf[0][2], acc, 0, 0, 0);
        }
        float pe[16];
#pragma unroll
        for (int g = 0; g < 16; g++) pe[g] = __builtin_amdgcn_exp2f(sc[g]);
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
            for (int j = 0; j < 8; j++) pf[s2][j] = (_Float16)pe[8 * s2 + j];
#pragma unroll
            for (int dt = 0; dt < 2; dt++) oacc[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s2][dt], pf[s2], oacc[0][dt], 0, 0, 0);
            lacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lones, pf[s2], lacc[0], 0, 0, 0);
        }
        if (NEXT) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[3], qf[0][3], acc, 0, 0, 0);
        // the first exp2 / converts run while the fragments are on their way from LDS, then one MFMA : two exp2 : a few VALU
        __builtin_amdgcn_sched_group_barrier(0x400, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
        for (int i = 0; i < (NEXT ? 10 : 6); i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        if (NEXT) mask_half(sc, kt + 1, H);   // (tests the tile's key count itself)
    };

    // ---- Q2 (QB = 2 at two waves per SIMD, NST = 5): 64 queries per wave, so every K / V^T fragment read from LDS feeds TWO MFMAs -- the LDS
    // pipe was the co-bottleneck of the 32-query wave (the premise was wrong -- CDNA4's LDS is 256 B / clk and the array 21 % busy, profiles/r03_pmc_attention_lds.txt; ablations in
    // profiles/r03_attn_ablate.txt: no LDS reads -12..14 %, exp2 -> v_mul only -3 %).  256 registers hold ONE fragment set and no spare score
    // block, so a half-tile step is two regions that pipeline the two query blocks against each other inside the wave:
    //   region 1: MFMAs of block 0 (next scores -> sc[0], O[0] += V P0, row sums)  ||  exp2 / convert of block 1 (sc[1] -> p1)
    //   region 2: MFMAs of block 1 (next scores -> sc[1], O[1] += V P1, row sums)  ||  exp2 / convert of block 0's NEW scores (sc[0] -> p0)
    // p0 (8 registers) crosses the step boundary; the partner wave on the SIMD covers LDS latency (fragments are read at the top of the step).
    constexpr bool Q2 = QB == 2;
    // MFMAs of the Q2 step as asm with explicit register files: score blocks (read by v_exp) and the offsets (their C operand: acc_cd is ONE bit
    // for C and D) in VGPRs, O and the row sums in AGPRs.  Left to hipcc the offsets went to AGPRs, so the scores did too, and every score cost a
    // v_accvgpr_read before its exp2 (64 extra VALU per tile, s_nop 11 behind the chain).  asm volatile also pins the MFMA order; the groups
    // below are fenced with sched_barrier(0), which hand-places the VALU of the other query block between them.
    auto mf_first = [&](f32x16& d, const f16x8& a_, const f16x8& b_, const f32x16& c_) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "+v"(d) : "v"(a_), "v"(b_), "v"(c_));   // ("+": the block keeps its registers round the loop)
    };
    auto mf_acc_v = [&](f32x16& d, const f16x8& a_, const f16x8& b_) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a_), "v"(b_));
    };
    // (at two waves per SIMD hipcc splits the 256 registers 128 : 128 as soon as anything is pinned to an AGPR: everything in VGPRs there)
    auto mf_acc_a = [&](f32x16& d, const f16x8& a_, const f16x8& b_) {
        if constexpr (NST >= 9) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a_), "v"(b_));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a_), "v"(b_));
    };
    auto mf_sum_a = [&](f32x4& d, const f16x8& a_, const f16x8& b_) {
        if constexpr (NST >= 9) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a_), "v"(b_));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a_), "v"(b_));
    };
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    f16x8 p0[2];
    // exp2 of scores [G0, G1) of a block and the packed converts of the pairs [C0, C1) (a pair = scores 2c, 2c + 1 -> element c of the P fragments)
    auto q2_exp = [&](f32x16& sc, auto g0_, auto g1_) {
#pragma unroll
        for (int g = decltype(g0_)::value; g < decltype(g1_)::value; g++) sc[g] = (A3_ABL & 2) ? sc[g] * 0.001f : __builtin_amdgcn_exp2f(sc[g]);
    };
    auto q2_cvt = [&](const f32x16& sc, f16x8 (&pf)[2], auto c0_, auto c1_) {
#pragma unroll
        for (int c = decltype(c0_)::value; c < decltype(c1_)::value; c++) {
            const f16x2_ h = __builtin_convertvector((f32x2_){sc[2 * c], sc[2 * c + 1]}, f16x2_);
            pf[c >> 2][(c & 3) * 2] = h[0];
            pf[c >> 2][(c & 3) * 2 + 1] = h[1];
        }
    };
    auto q2_softmax = [&](f32x16& sc, f16x8 (&pf)[2]) {
        using I0 = std::integral_constant<int, 0>;
        q2_exp(sc, I0{}, std::integral_constant<int, 16>{});
        q2_cvt(sc, pf, I0{}, std::integral_constant<int, 8>{});
    };
#define A3_FENCE() __builtin_amdgcn_sched_barrier(0)
#define A3_I(N) std::integral_constant<int, N> {}
    // one region: MFMAs of block A (next scores -> sa, O[A] += V pa, row sums)  ||  exp2 / convert of block B (sb -> pb)
    auto q2_region = [&](f32x16& sa_, const f32x16& negm_a, const f16x8 (&qa)[4], f32x16 (&oa)[2], f32x4& la, const f16x8 (&pa)[2],
                         f32x16& sb_, f16x8 (&pb)[2], Frags& fr, auto next_t, auto soft_t, auto&& after_qk) {
        constexpr bool NEXT = decltype(next_t)::value, SOFT = decltype(soft_t)::value;
        A3_FENCE();
        if (NEXT) mf_first(sa_, fr.k[0], qa[0], negm_a);
        if (SOFT) q2_exp(sb_, A3_I(0), A3_I(2));
        A3_FENCE();
        if (NEXT) mf_acc_v(sa_, fr.k[1], qa[1]);
        if (SOFT) { q2_exp(sb_, A3_I(2), A3_I(4)); q2_cvt(sb_, pb, A3_I(0), A3_I(1)); }
        A3_FENCE();
        if (NEXT) mf_acc_v(sa_, fr.k[2], qa[2]);
        if (SOFT) { q2_exp(sb_, A3_I(4), A3_I(6)); q2_cvt(sb_, pb, A3_I(1), A3_I(2)); }
        A3_FENCE();
        if (NEXT) mf_acc_v(sa_, fr.k[3], qa[3]);
        if (SOFT) { q2_exp(sb_, A3_I(6), A3_I(8)); q2_cvt(sb_, pb, A3_I(2), A3_I(3)); }
        A3_FENCE();
        after_qk();
        mf_acc_a(oa[0], fr.v[0][0], pa[0]);
        if (SOFT) { q2_exp(sb_, A3_I(8), A3_I(10)); q2_cvt(sb_, pb, A3_I(3), A3_I(4)); }
        A3_FENCE();
        mf_acc_a(oa[1], fr.v[0][1], pa[0]);
        if (SOFT) { q2_exp(sb_, A3_I(10), A3_I(12)); q2_cvt(sb_, pb, A3_I(4), A3_I(5)); }
        A3_FENCE();
        mf_sum_a(la, lones, pa[0]);
        if (SOFT) q2_exp(sb_, A3_I(12), A3_I(14));
        A3_FENCE();
        mf_acc_a(oa[0], fr.v[1][0], pa[1]);
        if (SOFT) { q2_exp(sb_, A3_I(14), A3_I(16)); q2_cvt(sb_, pb, A3_I(5), A3_I(6)); }
        A3_FENCE();
        mf_acc_a(oa[1], fr.v[1][1], pa[1]);
        if (SOFT) q2_cvt(sb_, pb, A3_I(6), A3_I(8));
        A3_FENCE();
        mf_sum_a(la, lones, pa[1]);
        A3_FENCE();
    };
    // Q2 (QB = 2: 64 queries per wave, one wave per SIMD): every K / V^T fragment read from LDS feeds TWO MFMAs -- the LDS pipe was the
    // co-bottleneck of the 32-query wave (premise wrong: the LDS array is 21 % busy, profiles/r03_pmc_attention_lds.txt; profiles/r03_attn_ablate.txt:
    // no LDS reads -12..14 %, exp2 -> v_mul only -3 %).  A half-tile step is two regions that pipeline the two query blocks against each
    // other inside the wave: region 1 = MFMAs of block 0 || softmax of block 1, region 2 = MFMAs of block 1 || softmax of block 0's NEW scores.
    // p0 (the converted probabilities of block 0) crosses the step boundary.
    auto q2_step = [&](f32x16 (&sc)[QB], int kt, Frags& fr, auto h_t, auto next_tile_t, auto mask_t) {
        constexpr int H = decltype(h_t)::value;
        constexpr bool NEXT_TILE = decltype(next_tile_t)::value, MASK = decltype(mask_t)::value;
        constexpr bool HAS_NEXT = H == 0 || NEXT_TILE;
        constexpr int QB1 = QB - 1;   // (= 1; keeps the QB = 1 instantiations well-formed)
        if (H == 0 && !(A3_ABL & 1)) ring_step(kt);
        // ONE fragment set, refilled in place: the V^T fragments of this step here (first used five MFMA groups on), the K fragments of the
        // FOLLOWING step in the middle of region 2, when the last score MFMA of this step has read them (six groups before their first use)
        if (!(A3_ABL & 4)) {
            const unsigned vb = stage_of(kt) + v_lane;
            v_read(fr.v[0], vb, H, 0);
            v_read(fr.v[1], vb, H, 1);
        }
        f16x8 p1[2];
        using N_ = std::integral_constant<bool, HAS_NEXT>;
        q2_region(sc[0], negm[0], qf[0], oacc[0], lacc[0], p0, sc[QB1], p1, fr, N_{}, std::true_type{}, [] {});
        if (HAS_NEXT && (SEG2 || MASK)) { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); mask_half(sc[0], H == 0 ? kt : kt + 1, H == 0 ? 1 : 0); }
        q2_region(sc[QB1], negm[QB1], qf[QB1], oacc[QB1], lacc[QB1], p1, sc[0], p0, fr, N_{}, N_{}, [&] {
            if (NEXT_TILE && !(A3_ABL & 4)) qk_read(fr.k, kt + 1, H == 0 ? 0 : 1);
        });
        if (HAS_NEXT && (SEG2 || MASK)) { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); mask_half(sc[QB1], H == 0 ? kt : kt + 1, H == 0 ? 1 : 0); }
    };
#undef A3_I
#undef A3_FENCE

    f32x16 sa[QB];   // the score blocks of the half tile about to be consumed
    Frags fa, fb;
    const int h_first = (BAL && role == 2) ? 1 : 0;   // the half of tile 0 this wave starts with
    // tile 0 must have landed before the first score block: tiles 1 .. NST - 2 may stay in flight (PAIR: tiles 0-2 landed, 3 .. NST - 1 in flight)
    if (PAIR) wait_landed(max(0, min(NST - 1, nkt - 1) - 2));
    else wait_landed(min(NST - 2, nkt - 1));
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {   // the queries' offsets = 2 + their maxima over the sample keys (C = 0: negm is still zero)
        f32x16 ss;
        qk_mfma(ss, ksamp, qb);
        const float m0 = row_max(ss) + A3_OFF_MARGIN;
#pragma unroll
        for (int g = 0; g < 16; g++) negm[qb][g] = -m0;
    }
    {
        f16x8 kf0[4];
        if (BAL && role == 2) qk_read(kf0, 0, 1); else qk_read(kf0, 0, 0);
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            qk_mfma(sa[qb], kf0, qb);   // = s - offset
            mask_half(sa[qb], 0, h_first);   // (role 2 with at most 32 keys: nothing valid in its half -- its P are exp2(-1e30) = 0, its O and l stay 0)
        }
    }
    if (!BAL || role == 0) {
        qk_read(fa.k, 0, 1);    // what half (0, 0) multiplies: K rows 32-63 and V keys 0-31 of tile 0
        v_read(fa.v[0], stage_of(0) + v_lane, 0, 0);
        v_read(fa.v[1], stage_of(0) + v_lane, 0, 1);
    }
    A3_STAMP(-1);
    st_pro = st_prev - st_t0;
    using T_ = std::true_type;
    using F_ = std::false_type;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    if (Q2) {
        q2_softmax(sa[0], p0);   // block 0 of the first half: its probabilities enter the loop converted
        int kt = 0;
        for (; kt + 2 < nkt; kt++) {
            q2_step(sa, kt, fa, H0{}, T_{}, F_{});
            q2_step(sa, kt, fa, H1{}, T_{}, F_{});
        }
        for (; kt < nkt; kt++) {
            if (kt + 1 < nkt) {
                q2_step(sa, kt, fa, H0{}, T_{}, T_{});
                q2_step(sa, kt, fa, H1{}, T_{}, T_{});
            } else {
                q2_step(sa, kt, fa, H0{}, F_{}, T_{});
                q2_step(sa, kt, fa, H1{}, F_{}, F_{});
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last asm MFMAs' results (O, row sums) are read by compiler-scheduled code next
    } else if (!BAL || role == 0) {
        int kt = 0;
        for (; kt + 2 < nkt; kt++) {   // tiles whose successors are complete tiles: no masks
            fast_half(sa, kt, fa, fb, H0{}, T_{}, F_{});
            fast_half(sa, kt, fb, fa, H1{}, T_{}, F_{});
        }
        for (; kt < nkt; kt++) {   // the last one or two tiles
            if (kt + 1 < nkt) {
                fast_half(sa, kt, fa, fb, H0{}, T_{}, T_{});
                fast_half(sa, kt, fb, fa, H1{}, T_{}, T_{});
            } else {
                fast_half(sa, kt, fa, fb, H0{}, F_{}, T_{});
                fast_half(sa, kt, fb, fa, H1{}, F_{}, F_{});
            }
        }
    } else if (role == 1) {   // (one ring step per tile, like the whole-block waves: the barriers pair up)
        for (int kt = 0; kt + 1 < nkt; kt++) half_only(sa[0], kt, H0{}, T_{});
        half_only(sa[0], nkt - 1, H0{}, F_{});
    } else {
        for (int kt = 0; kt + 1 < nkt; kt++) half_only(sa[0], kt, H1{}, T_{});
        half_only(sa[0], nkt - 1, H1{}, F_{});
    }
    // Did every probability stay in range (l_bad above)?  (The flag lives in the first bytes of the ring -- 5 x 16 KiB is exactly half a CU's LDS, two workgroups
    // per CU -- hence the barriers: everybody done with the ring | flag cleared | flag set | flag read.)
    // the row sums of this wave's queries out of the 16 x 16 accumulators: query fr sits in lane fr & 15, register fr >> 4.  Kept as before as a
    // per-lane partial that the epilogue (and the BAL merge) adds over the two lane halves: the whole sum in lanes 0-31, zero in lanes 32-63.
    bool l_bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const float l0 = __shfl(lacc[qb][0], fr & 15, 64), l1 = __shfl(lacc[qb][1], fr & 15, 64);
        const float lq = (fr >> 4) ? l1 : l0;
        l_bad = l_bad || !(lq <= 3.0e38f);                 // inf (a probability overflowed fp16) or NaN
        lrun[qb] = fh ? 0.0f : lq;
    }
    int* redo_flag = reinterpret_cast<int*>(smem);
    attn_wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) *redo_flag = 0;
    __syncthreads();
    if (__any(l_bad) && lane == 0) *redo_flag = 1;
    __syncthreads();
    const bool redo = *redo_flag != 0 && !A3_ABL;
    __syncthreads();
    if (redo) {   // GENERAL loop: running maximum, any input; plain code, one query block at a time
        float mrun[QB];
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            lrun[qb] = 0.0f;
            mrun[qb] = -1e30f;
#pragma unroll
            for (int g = 0; g < 16; g++) { oacc[qb][0][g] = 0.0f; oacc[qb][1][g] = 0.0f; negm[qb][g] = 0.0f; }
        }
#pragma unroll
        for (int t = 0; t < NST - 1; t++)
            if (t < nkt) issue_tile(t);
        for (int gk = 0; gk < nkt; gk++) {
            // tile gk landed for everybody; every wave is past tile gk - 1, so its stage can take tile gk + NST - 2
            wait_landed(min(gk == 0 ? NST - 2 : NST - 3, nkt - 1 - gk));   // issued so far: tiles 0 .. NST - 2 at gk = 0, up to gk + NST - 3 after
            __builtin_amdgcn_s_barrier();
            if (gk > 0 && gk + NST - 2 < nkt) issue_tile(gk + NST - 2);
            const unsigned vb = stage_of(gk) + v_lane;
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                if (BAL && role == 2) break;   // (waves 6 / 7 only move tiles and keep the barriers; blocks 4 / 5 are redone whole by waves 4 / 5)
                f16x8 kf[4];
                qk_read(kf, gk, h);
#pragma unroll
                for (int qb = 0; qb < QB; qb++) {
                    f32x16 sg;
                    qk_mfma(sg, kf, qb);
                    mask_half(sg, gk, h);
                    const float mnew = fmaxf(mrun[qb], row_max(sg));
                    const float alpha = __builtin_amdgcn_exp2f(mrun[qb] - mnew);
                    mrun[qb] = mnew;
                    float rs = 0.0f;
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        sg[g] = __builtin_amdgcn_exp2f(sg[g] - mnew);
                        rs += sg[g];
                    }
                    lrun[qb] = lrun[qb] * alpha + rs;
#pragma unroll
                    for (int dt = 0; dt < 2; dt++)
#pragma unroll
                        for (int g = 0; g < 16; g++) oacc[qb][dt][g] *= alpha;
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) {
                        f16x8 pf, vf[2];
#pragma unroll
                        for (int j = 0; j < 8; j++) pf[j] = (_Float16)sg[8 * s2 + j];
                        v_read(vf, vb, h, s2);
#pragma unroll
                        for (int dt = 0; dt < 2; dt++) oacc[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dt], pf, oacc[qb][dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (BAL) {
        // merge the key halves of blocks 4 / 5: waves 6 / 7 hand (offset, l, O) to waves 4 / 5 through LDS (the ring is idle), which bring both to the
        // larger offset -- exact power-of-two factors <= 1 -- and add.  After a redo the whole-block result of waves 4 / 5 stands alone.
        float* mbuf = reinterpret_cast<float*>(smem) + (size_t)(wave & 1) * 34 * 64;
        __syncthreads();
        if (role == 2 && !redo) {
            mbuf[lane] = -negm[0][0];
            mbuf[64 + lane] = lrun[0];
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int g = 0; g < 16; g++) mbuf[(2 + dt * 16 + g) * 64 + lane] = oacc[0][dt][g];
        }
        __syncthreads();
        if (role == 1 && !redo) {
            const float off_a = -negm[0][0], off_b = mbuf[lane];
            const float m = fmaxf(off_a, off_b);
            const float fa_ = __builtin_amdgcn_exp2f(off_a - m), fb_ = __builtin_amdgcn_exp2f(off_b - m);
            lrun[0] = lrun[0] * fa_ + mbuf[64 + lane] * fb_;
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int g = 0; g < 16; g++) oacc[0][dt][g] = oacc[0][dt][g] * fa_ + mbuf[(2 + dt * 16 + g) * 64 + lane] * fb_;
        }
        if (role == 2) return;   // (no barrier after this point)
    }
    if (STAMPS && st_on && lane == 0) { p.dbg[0] = st_acc[0]; p.dbg[1] = st_acc[1]; p.dbg[2] = st_acc[2]; p.dbg[3] = (unsigned long long)nkt; p.dbg[4] = st_acc[4]; p.dbg[5] = st_acc[5]; p.dbg[6] = st_pro; }

#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const float ltot = lrun[qb] + __shfl_xor(lrun[qb], 32, 64);
        const float inv = 1.0f / ltot;
        const int q = q0 + (qblk + qb) * 32 + fr;
        if (q < len) {
            const size_t obase = (size_t)(row0 + q) * D + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    bf16x4 hi4, lo4;
                    float ov[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) ov[e] = oacc[qb][dt][a * 4 + e] * inv;
                    const int d = dt * 32 + 8 * a + 4 * fh;
                    if (p.f16_out) {
                        store_f16x4(p.out_hi + obase + d, ov);
                        continue;
                    }
                    split_bf16x4(ov, hi4, lo4);
                    *reinterpret_cast<bf16x4*>(p.out_hi + obase + d) = hi4;
                    if (p.out_lo) *reinterpret_cast<bf16x4*>(p.out_lo + obase + d) = lo4;
                }
        }
    }
    A3_STAMP(-1);
    if (STAMPS && st_on && lane == 0) { p.dbg[7] = st_prev - st_t0; p.dbg[8] = __builtin_amdgcn_s_memrealtime() - st_r0; }   // (stores issued, not retired)
#undef A3_STAMP
}
