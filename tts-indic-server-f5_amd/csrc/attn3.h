// attn3: attn2 with the QK^T MFMAs of KV tile j+1 issued BEFORE the softmax of tile j (software pipelining inside each wave).
//
// rocprof + cycle counts on attn2 (MI355X, N = 1404): ~3800 cycles per KV tile and SIMD against ~1000 cycles of MFMA
// (2 waves x 16 MFMAs) and ~2200 cycles of VALU (2 waves x 32 exp2 at quarter rate + max / sum / scale / convert):
// the per-tile barrier keeps the two waves of a SIMD in lockstep, so matrix and vector phases add up instead of
// overlapping.  Here each wave carries two score accumulators: S_next = K_{j+1} Q^T goes to the matrix pipe, then the
// softmax of S_cur runs on the VALU while those MFMAs execute, then O += V_j P.  The LDS ring is 5 deep so the K tile
// one step ahead is resident while 3 more tiles stay in flight.
#pragma once
#include "attn_common.h"

// NW = waves per workgroup = 32-query slices per query tile (4, 6 or 8).  The launcher picks NW so that the grid fills the 256 CUs in
// whole rounds: at the C2 shape (2 sequences x 16 heads x 1404 queries) 256-query tiles give 192 workgroups -- 64 CUs idle for the whole
// launch -- while 192-query tiles (NW = 6) give exactly 256 workgroups with 3/4 of the work each.
// SEG2: the keys are two row ranges (AttnArgs::seq_kv_row0 / seq_kv2_*): tile kt covers 64 rows of the first range while kt < nkt1, of
// the second after it; the last tile of EACH range is masked.  SEG2 = false is the single-range kernel of the DiT / UNetT path, unchanged.
template <int NW, bool SEG2 = false>
static __global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn3_fwd_kernel(const AttnArgs p) {
    constexpr int NST = 5, STAGE = 16384;
    constexpr int P_HI = (16 + NW - 1) / NW, P_LO = 16 / NW;   // 1 KiB pieces of a KV tile per wave (pieces w, w + NW, ...)
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    const int seq = blockIdx.z, head = blockIdx.y;
    const int len = p.seq_len[seq], kvlen = p.seq_kvlen[seq], row0 = p.seq_row0[seq];
    const int kv_row0 = SEG2 ? p.seq_kv_row0[seq] : row0, kv2_row0 = SEG2 ? p.seq_kv2_row0[seq] : 0, kv2_len = SEG2 ? p.seq_kv2_len[seq] : 0;
    const int nkt1 = (kvlen + 63) >> 6;
    const int q0 = blockIdx.x * (32 * NW);
    if (q0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int D = p.D;
    const float LOG2E = 1.4426950408889634f;

    // queries of this wave (rows beyond the sequence stay inside its 128-row padding or the next sequence: finite data,
    // never stored).  q0 + 255 can exceed the padded rows of the LAST sequence only by < 256 rows: workspace has slack.
    bf16x8 qf[4];
    {
        const __bf16* qrow = p.qk + (size_t)(row0 + q0 + wave * 32 + fr) * (2 * D) + head * 64 + fh * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + s * 16);
    }
    // Retire the Q loads BEFORE the first LDS-DMA is issued: with a DMA in flight hipcc can only wait vmcnt(0) for an
    // ordinary VGPR load, and it would put that wait inside the KV loop, draining the ring every tile.
    asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]) : "memory");

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
        if (mine == P_HI) {
            if (newer >= 3) attn_wait_vmcnt<3 * P_HI>(); else if (newer == 2) attn_wait_vmcnt<2 * P_HI>(); else if (newer == 1) attn_wait_vmcnt<P_HI>(); else attn_wait_vmcnt<0>();
        } else {
            if (newer >= 3) attn_wait_vmcnt<3 * P_LO>(); else if (newer == 2) attn_wait_vmcnt<2 * P_LO>(); else if (newer == 1) attn_wait_vmcnt<P_LO>(); else attn_wait_vmcnt<0>();
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int g = 0; g < 16; g++) oacc[dt][g] = 0.0f;
    float mrun = -1e30f, lrun = 0.0f;

    const int nkt = nkt1 + (SEG2 ? (kv2_len + 63) >> 6 : 0);
#pragma unroll
    for (int t = 0; t < NST - 1; t++)
        if (t < nkt) issue_tile(t);

    // S^T tile = K_tile Q^T for the 64 keys of ring stage `st` (8 MFMAs), masked on the last, partial tile
    auto qk_tile = [&](f32x16 (&s)[2], int kt) {
        const char* kst = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
#pragma unroll
            for (int g = 0; g < 16; g++) s[kh][g] = 0.0f;
#pragma unroll
            for (int sI = 0; sI < 4; sI++) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(kst + lds_off128(kh * 32 + fr, 2 * sI + fh));
                s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[sI], s[kh], 0, 0, 0);
            }
        }
        // keys of this tile that exist: the rest of its range (a tile never straddles the two ranges)
        const int valid = (!SEG2 || kt < nkt1) ? kvlen - kt * 64 : kv2_len - (kt - nkt1) * 64;
        if (valid < 64) {   // key-padding mask
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int key = kh * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh;
                    if (key >= valid) s[kh][g] = -1e30f;
                }
        }
    };

    typedef __attribute__((ext_vector_type(2))) float f32x2;
    // softmax of one score tile + O += V P for ring stage of tile `kt` (packed fp32 VALU ops: two scores per instruction)
    auto softmax_pv = [&](f32x16 (&sacc)[2], int kt) {
        const char* vst = smem + (kt % NST) * STAGE + 8192;
        float mloc = sacc[0][0];
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int g = 0; g < 16; g++) mloc = fmaxf(mloc, sacc[kh][g]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(mrun, mloc);
        const bool moved = mnew != mrun;
        const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * LOG2E);
        mrun = mnew;
        const f32x2 msc2 = {-mnew * LOG2E, -mnew * LOG2E}, l2e2 = {LOG2E, LOG2E};
        f32x2 rs2 = {0.0f, 0.0f};
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                f32x2 t = {sacc[kh][g], sacc[kh][g + 1]};
                t = __builtin_elementwise_fma(t, l2e2, msc2);
                t[0] = __builtin_amdgcn_exp2f(t[0]);
                t[1] = __builtin_amdgcn_exp2f(t[1]);
                sacc[kh][g] = t[0];
                sacc[kh][g + 1] = t[1];
                rs2 += t;
            }
        lrun = lrun * alpha + (rs2[0] + rs2[1]);
        if (__any(moved)) {   // wave-uniform; alpha == 1 exactly for every query whose maximum did not move
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int g = 0; g < 16; g++) oacc[dt][g] *= alpha;
        }
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; j++) pf[j] = (__bf16)sacc[kh][8 * s2 + j];
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const int row = dt * 32 + fr, c0 = kh * 4 + s2 * 2;
                    const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vst + lds_off128(row, c0) + fh * 8);
                    const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vst + lds_off128(row, c0 + 1) + fh * 8);
                    bf16x8 vf;
#pragma unroll
                    for (int e = 0; e < 4; e++) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
                }
            }
        }
    };
    // ring step kt: tile kt + 1 landed (its K feeds the next score tile); every wave is past step kt - 1, so stage (kt - 1) % NST is free
    auto ring_step = [&](int kt) {
        wait_landed(min(2, nkt - 2 - kt));   // tiles kt + 2, kt + 3 may stay in flight
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nkt) issue_tile(kt + NST - 1);
    };

    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
    const bool st_on = p.dbg != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0;
#define A3_STAMP(I)                                                                                  \
    if (st_on) {                                                                                     \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if ((I) >= 0) st_acc[(I) < 0 ? 0 : (I)] += t_ - st_prev;                                     \
        st_prev = t_;                                                                                \
    }
    f32x16 sa[2], sb[2];   // the two score tiles swap roles every step (loop unrolled by two: no register copies)
    // tile 0 must have landed before the first score tile: tiles 1..3 may stay in flight
    wait_landed(min(3, nkt - 1));
    __builtin_amdgcn_s_barrier();
    qk_tile(sa, 0);
    A3_STAMP(-1);
    for (int kt = 0; kt < nkt; kt += 2) {
        ring_step(kt);
        A3_STAMP(0);
        if (kt + 1 < nkt) qk_tile(sb, kt + 1);
        A3_STAMP(1);
        softmax_pv(sa, kt);
        A3_STAMP(2);
        if (kt + 1 >= nkt) break;
        ring_step(kt + 1);
        A3_STAMP(0);
        if (kt + 2 < nkt) qk_tile(sa, kt + 2);
        A3_STAMP(1);
        softmax_pv(sb, kt + 1);
        A3_STAMP(2);
    }
    if (st_on && lane == 0) { p.dbg[0] = st_acc[0]; p.dbg[1] = st_acc[1]; p.dbg[2] = st_acc[2]; p.dbg[3] = (unsigned long long)nkt; }
#undef A3_STAMP

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
