// attn5: flash-style attention forward as a PING-PONG of the two waves of every SIMD (round 3; single key range: the DiT / UNetT path).
// MEASURED AND NOT SHIPPED (-DF5HIP_EXPERIMENTS builds, F5HIP_ATTN5=1 / f5hip_op_attention impl 5; profiles/r03_attn5_pingpong.txt): parity-identical
// to attn3 on every test shape (same maximum errors to the digit) and 11-40 % SLOWER -- 212-222 us against 183-199 at 16 x 1404, 37-38 against
// 31-33 at C2.  Why: at head dimension 64 the softmax, not the matrix pipe, is the longer half of a tile (per wave 32 v_exp at 8 issue cycles,
// ~75 other vector instructions at 4, 16 fragment reads: ~800 cycles against 512 of MFMA), and a phase structure that lets only ONE wave per SIMD
// issue vector work at a time runs that half at the single-wave issue rate; attn3's two interleaving waves share the vector pipe at its
// aggregate rate.  The pure-phase structure pays where the matrix phase is the long one (gemm6.h), not here.
//
// Same mathematics, operand layouts and conventions as attn3.h (S^T = K Q^T so a query is a lane; fp16 operands; fixed per-query softmax offset
// from 32 sample keys folded into the first score MFMA, range-checked once after the loop, running-maximum redo; V^T in vt_col order; K / V^T
// tiles of 64 keys through an LDS ring by LDS-DMA).  What differs is WHEN a wave does what.  attn3 interleaves, inside every wave, the score
// MFMAs of the next half tile with the exponentials of the current one (sched_group_barrier): per 64-key tile and wave 16 MFMAs, 32 v_exp,
// ~75 other VALU, 16 ds_read_b128 -- ~1 070 issue cycles per SIMD (two waves) against 1 024 of matrix pipe, balanced on paper -- and runs in
// ~2 500: an in-order wave that waits for its MFMA results, its LDS fragments or the ring barrier blocks BOTH streams, and the per-tile barrier
// keeps the two waves of a SIMD in lock-step, so they want the matrix pipe, and then the vector pipe, at the same time.
//
// Here a wave alternates two PURE phases per tile, and the two wave groups (waves 0-3, waves 4-7: one of each on every SIMD) run half a tile
// apart (MI355X_MICROARCH.md "Two waves per SIMD"; the structure of gemm6.h):
//   V(i)  vector / memory phase:  P(i) = exp2(S(i)) (+ row sums, fp16 convert, key-padding mask on the last tile); read the V^T fragments of
//         tile i and the K fragments of tile i + 1 from LDS; issue the LDS-DMA pieces of tile i + NST - 1; wait for this wave's pieces of
//         tile i + 2; s_waitcnt lgkmcnt(0); s_barrier.
//   M(i)  matrix phase: S(i + 1) = K(i + 1) Q^T - offset (8 MFMAs) and O^T += V^T(i) P^T(i) (8 MFMAs): 512 cycles of the SIMD's matrix pipe,
//         nothing else; s_barrier.
// While one group sits in M the other sits in V: the matrix pipe is handed over at every barrier and each wave's exponentials, LDS reads and DMA
// issue happen beside its partner's MFMAs.  Every value is single-buffered (S, P, the K / V^T fragments are produced in one phase and consumed
// in the next).  A workgroup is 8 waves x 32 queries = 256 queries.
// Ring: tile i + 1 must be visible when V(i) reads its K rows.  Each wave waits for its own pieces of tile i + 2 at the end of V(i); the later
// group's V(i) ends one barrier after the earlier group's and one barrier before the earlier group's V(i + 1) begins, so every piece of tile
// i + 2 has landed before anybody reads it.  The stage of tile i - 1 is last read (V^T fragments) in V(i - 1) of the later group, which ends
// before the earlier group's V(i) begins: V(i) refills it with tile i - 1 + NST.
#pragma once
#include <type_traits>
#include "../attn3.h"

template <int NST>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn5_fwd_kernel(const AttnArgs p) {
    constexpr int STAGE = 16384;
    static_assert(NST >= 4 && NST <= 9, "ring depth");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {   // XCD-aware numbering: the query tiles of one (sequence, head) share an L2 (attn3.h)
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z, per = total >> 3;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        if (lin < per * 8) {
            const unsigned v = (lin & 7) * per + (lin >> 3);
            bx = (int)(v % gx); by = (int)((v / gx) % gy); bz = (int)(v / (gx * gy));
        }
    }
    const int seq = bz, head = by;
    const int len = p.seq_len[seq], kvlen = p.seq_kvlen[seq], row0 = p.seq_row0[seq];
    const int nkt = (kvlen + 63) >> 6;
    const int q0 = bx * 256;
    if (q0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int late = wave >> 2;                                  // the later group: half a tile behind
    const int fr = lane & 31, fh = lane >> 5;
    const int D = p.D;

    // queries of this wave + the 32 sample keys of its softmax offsets (attn3.h), loaded before any LDS-DMA is in flight
    f16x8 qf[4], ksamp[4];
    {
        const __bf16* qrow = p.qk + (size_t)(row0 + q0 + wave * 32 + fr) * (2 * D) + head * 64 + fh * 8;
        const int sidx = fr < 16 ? (fr * kvlen) >> 4 : min(q0 + wave * 32 + (fr - 16) * 2, kvlen - 1);
        const __bf16* krow = p.qk + (size_t)(row0 + sidx) * (2 * D) + D + head * 64 + fh * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) { qf[s] = *reinterpret_cast<const f16x8*>(qrow + s * 16); ksamp[s] = *reinterpret_cast<const f16x8*>(krow + s * 16); }
    }
    asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(ksamp[0]), "v"(ksamp[1]), "v"(ksamp[2]), "v"(ksamp[3]) : "memory");

    // LDS-DMA: wave w moves piece w (K rows 8 w ..) and piece 8 + w (V^T rows 8 w ..) of every tile; source-side XOR swizzle as attn3
    const char *src_k, *src_v;
    {
        const int prow = wave * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((prow >> 1) & 7);
        src_k = reinterpret_cast<const char*>(p.qk + (size_t)(row0 + prow) * (2 * D) + D + head * 64 + chunk * 8);
        src_v = reinterpret_cast<const char*>(p.vt + (size_t)(head * 64 + prow) * p.ldvt + row0 + chunk * 8);
    }
    const size_t step_k = (size_t)64 * (2 * D) * 2, step_v = 128;
    auto issue_tile = [&](int kt) {
        char* dst = smem + (kt % NST) * STAGE + wave * 1024;
        attn_lds_dma16(src_k + (size_t)kt * step_k, dst);
        attn_lds_dma16(src_v + (size_t)kt * step_v, dst + 8192);
    };
    auto wait_newer = [&](int newer) {   // this wave's pieces of a tile have landed when at most `newer` younger tiles of its own are in flight
        switch (newer) {
            case 0: attn_wait_vmcnt<0>(); break;
            case 1: attn_wait_vmcnt<2>(); break;
            case 2: attn_wait_vmcnt<4>(); break;
            case 3: attn_wait_vmcnt<6>(); break;
            case 4: attn_wait_vmcnt<8>(); break;
            case 5: attn_wait_vmcnt<10>(); break;
            default: attn_wait_vmcnt<12>(); break;
        }
    };
    static_assert(NST - 3 <= 6, "wait_newer covers 6 tiles in flight");

    const unsigned k_lane = (unsigned)(fr * 128 + ((fh ^ ((fr >> 1) & 7)) << 4));
    const unsigned v_lane = k_lane + 8192u;
    auto stage_of = [&](int kt) { return (unsigned)((kt % NST) * STAGE); };
    f16x8 kf[2][4], vf[2][2][2], pf[2][2];      // K fragments of the next tile [half][k-step], V^T fragments of this tile [half][k-step][32 features], P^T [half][k-step]
    auto read_k = [&](int kt) {
        const unsigned kb = stage_of(kt) + k_lane;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int sI = 0; sI < 4; sI++) kf[h][sI] = *reinterpret_cast<const f16x8*>(smem + (kb ^ (unsigned)(sI << 5)) + h * 4096);
    };
    auto read_v = [&](int kt) {
        const unsigned vb = stage_of(kt) + v_lane;
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int dt = 0; dt < 2; dt++) vf[h][s2][dt] = *reinterpret_cast<const f16x8*>(smem + (vb ^ (unsigned)((h * 2 + s2) << 5)) + dt * 4096);
    };
    auto mask_tile = [&](f32x16 (&s)[2], int kt) {   // key-padding mask of tile kt (only the last tile is partial)
        const int valid = kvlen - kt * 64;
        if (valid < 64) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int key = h * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh;
                    if (key >= valid) s[h][g] = -1e30f;
                }
        }
    };
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16 oacc[2], negm, sc[2];
    float lrun = 0.0f, rs_max = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; g++) { oacc[0][g] = 0.0f; oacc[1][g] = 0.0f; negm[g] = 0.0f; }
    {   // the queries' offsets = 2 + their maxima over the sample keys
        f32x16 ss = __builtin_amdgcn_mfma_f32_32x32x16_f16(ksamp[0], qf[0], negm, 0, 0, 0);
#pragma unroll
        for (int sI = 1; sI < 4; sI++) ss = __builtin_amdgcn_mfma_f32_32x32x16_f16(ksamp[sI], qf[sI], ss, 0, 0, 0);
        float m = ss[0];
#pragma unroll
        for (int g = 1; g < 16; g++) m = fmaxf(m, ss[g]);
        m = fmaxf(m, __shfl_xor(m, 32, 64)) + A3_OFF_MARGIN;
#pragma unroll
        for (int g = 0; g < 16; g++) negm[g] = -m;
    }

    // ---- prologue: tiles 0 .. NST - 2 in flight, tiles 0 and 1 landed for everybody
    const int n_pre = min(NST - 1, nkt);
#pragma unroll
    for (int t = 0; t < NST - 1; t++)
        if (t < nkt) issue_tile(t);
    wait_newer(max(0, n_pre - 2));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (late) phase_end();                                       // the later group: one phase behind from here on

    // matrix phase: S(next) = K(next) Q^T - offset (HAS_NEXT) and O^T += V^T P^T of the current tile (HAS_CUR)
    auto m_phase = [&](auto has_next_t, auto has_cur_t) {
        constexpr bool HAS_NEXT = decltype(has_next_t)::value, HAS_CUR = decltype(has_cur_t)::value;
        __builtin_amdgcn_s_setprio(1);
        if (HAS_NEXT) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                sc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[h][0], qf[0], negm, 0, 0, 0);
#pragma unroll
                for (int sI = 1; sI < 4; sI++) sc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[h][sI], qf[sI], sc[h], 0, 0, 0);
            }
        }
        if (HAS_CUR) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int dt = 0; dt < 2; dt++) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[h][s2][dt], pf[h][s2], oacc[dt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        phase_end();
    };
    // vector / memory phase of tile i (i = -1: only the first K fragments)
    auto v_phase = [&](int i) {
        if (i >= 0) {
            if (i == nkt - 1) mask_tile(sc, i);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float pe[16], r4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    pe[g] = __builtin_amdgcn_exp2f(sc[h][g]);
                    r4[g & 3] += pe[g];
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                    for (int j = 0; j < 8; j++) pf[h][s2][j] = (_Float16)pe[8 * s2 + j];
                const float rs = (r4[0] + r4[1]) + (r4[2] + r4[3]);
                lrun += rs;
                rs_max = fmaxf(rs_max, rs);
            }
            read_v(i);
        }
        if (i + 1 < nkt) read_k(i + 1);
        if (i >= 0) {
            if (i - 1 + NST < nkt) issue_tile(i - 1 + NST);       // into the stage of tile i - 1 (V(0): the one stage the prologue left empty)
            // tile i + 2 (first read in V(i + 1)): issued so far are the tiles up to min(i - 1 + NST, nkt - 1)
            if (i + 2 < nkt) wait_newer(min(i - 1 + NST, nkt - 1) - (i + 2));
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the fragments are in registers, this phase's LDS reads are retired
        phase_end();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    v_phase(-1);
    m_phase(T_{}, F_{});                                         // S(0)
    for (int i = 0; i + 1 < nkt; i++) {
        v_phase(i);
        m_phase(T_{}, T_{});
    }
    v_phase(nkt - 1);
    m_phase(F_{}, T_{});
    if (!late) phase_end();                                      // (pairs with the later group's last barrier)

    // ---- range check (attn3.h): did every probability stay a finite fp16?  else the workgroup redoes its tile with a running maximum
    int* redo_flag = reinterpret_cast<int*>(smem);
    attn_wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) *redo_flag = 0;
    __syncthreads();
    if (__any(!(rs_max <= A3_P_LIMIT)) && lane == 0) *redo_flag = 1;
    __syncthreads();
    const bool redo = *redo_flag != 0;
    __syncthreads();
    if (redo) {   // GENERAL loop: all eight waves in step, plain code
        float mrun = -1e30f;
        lrun = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; g++) { oacc[0][g] = 0.0f; oacc[1][g] = 0.0f; negm[g] = 0.0f; }
#pragma unroll
        for (int t = 0; t < NST - 1; t++)
            if (t < nkt) issue_tile(t);
        for (int gk = 0; gk < nkt; gk++) {
            // tile gk landed for everybody; every wave is past tile gk - 1, so its stage can take tile gk + NST - 2
            wait_newer(min(gk == 0 ? NST - 2 : NST - 3, nkt - 1 - gk));
            __builtin_amdgcn_s_barrier();
            if (gk > 0 && gk + NST - 2 < nkt) issue_tile(gk + NST - 2);
            read_k(gk);
            read_v(gk);
            f32x16 sg[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                sg[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[h][0], qf[0], negm, 0, 0, 0);
#pragma unroll
                for (int sI = 1; sI < 4; sI++) sg[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[h][sI], qf[sI], sg[h], 0, 0, 0);
            }
            mask_tile(sg, gk);
            float mnew = mrun;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int g = 0; g < 16; g++) mnew = fmaxf(mnew, sg[h][g]);
            mnew = fmaxf(mnew, __shfl_xor(mnew, 32, 64));
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            mrun = mnew;
            float rs = 0.0f;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int g = 0; g < 16; g++) { sg[h][g] = __builtin_amdgcn_exp2f(sg[h][g] - mnew); rs += sg[h][g]; }
            lrun = lrun * alpha + rs;
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int g = 0; g < 16; g++) oacc[dt][g] *= alpha;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    f16x8 pg;
#pragma unroll
                    for (int j = 0; j < 8; j++) pg[j] = (_Float16)sg[h][8 * s2 + j];
#pragma unroll
                    for (int dt = 0; dt < 2; dt++) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[h][s2][dt], pg, oacc[dt], 0, 0, 0);
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    }

    const float ltot = lrun + __shfl_xor(lrun, 32, 64);
    const float inv = 1.0f / ltot;
    const int q = q0 + wave * 32 + fr;
    if (q < len) {
        const size_t obase = (size_t)(row0 + q) * D + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int a = 0; a < 4; a++) {
                bf16x4 hi4, lo4;
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; e++) ov[e] = oacc[dt][a * 4 + e] * inv;
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
