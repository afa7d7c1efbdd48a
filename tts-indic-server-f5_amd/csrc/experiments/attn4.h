// attn4: flash-style non-causal self-attention forward for gfx950, head dim 64, bf16 MFMA (16 x 16 x 32), fp32 online softmax --
// the production attention kernel (F/model/modules.py:424-436: SDPA with the key-padding mask).
//
// Why a different shape than attn3 (8 waves x 32 queries): the kernel is bound by VALU work per SIMD (exp2, max, sum, scale, convert:
// profiles/r01 attention stamps), and at the C2 shape the 2 x 16 x ceil(1404 / 32) = 1408 query slices of 32 do not divide over the
// chip's 1024 SIMDs: some SIMD always carries two slices (64 queries), whatever the tile height (measured: 192-query tiles = exactly
// 256 workgroups took the same 35 us as 256-query tiles on 192 CUs).  16 x 16 x 32 MFMA blocks make 16 queries the unit, and a
// workgroup is 8 waves of UNEQUAL height: waves 0-3 own 16 QA queries, waves 4-7 own 16 QC.  QA = 2, QC = 1 puts 48 queries on every
// SIMD (192 per workgroup, exactly 256 workgroups at C2: 0.75 of the per-SIMD work of attn3) and still two instruction streams per
// SIMD to cover each other's latencies -- a single 48-query wave per SIMD (first version of this file) measured SLOWER per query than
// attn3: with one wave per SIMD every exp2 / MFMA-result / LDS latency is exposed.
//   * every wave streams 2 of the 16 one-KiB pieces of a KV tile (64 keys: 8 K pieces + 8 V^T pieces) by LDS-DMA into a 5-deep ring
//     behind a counted s_waitcnt vmcnt; one raw s_barrier per tile.  The DMA is issued from an asm statement: with the builtin in the
//     same loop as the fragment reads hipcc degrades every lgkmcnt wait to lgkmcnt(0) (see gemm5.h);
//   * S^T = K Q^T per 16-key x 16-query block (keys on the accumulator registers, queries on the lanes): the row maximum / sum of a
//     query are in-register reductions plus two v_permlane swaps, and the exponentiated block is directly the B operand of
//     O^T += V^T P^T (the key order inside a k-step is permuted identically on both operands: two 8-byte LDS reads of V^T);
//   * the score blocks of tile j + 1 are computed into the registers of tile j's blocks as those are exponentiated and packed (one
//     score buffer), so the matrix pipe works under the softmax's VALU stream;
//   * Q is pre-scaled by 1/8 in the QKV epilogue; keys >= kv_len are masked to -1e30 before the maximum; rows >= len are never stored.
#pragma once
#include <type_traits>

#include "../attn_common.h"

// all-reduce over the 4 lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48) that hold one query: v_permlane16_swap / v_permlane32_swap exchange
// 16- / 32-lane halves between two registers in the VALU (no LDS round trip: __shfl_xor is a ds_bpermute, ~100 exposed cycles each with one
// wave per SIMD, six per KV tile)
// (inline asm: through the __builtin_amdgcn_permlane*_swap builtins hipcc -- ROCm 7.2 -- folded the max / add of the two results into
//  the first result alone, even with the operands made distinct by an empty asm: tools/permlane_probe.hip shows the instruction itself
//  does what the ISA says.  s_nop 1 = the two wait states a VALU write of an operand needs before the swap reads it.)
F5_DEVICE void xgroup_swap16(float v, float& x, float& y) {
    x = v; y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
F5_DEVICE void xgroup_swap32(float v, float& x, float& y) {
    x = v; y = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
}
F5_DEVICE float xgroup_max(float v) {
    float x, y;
    xgroup_swap16(v, x, y);
    v = fmaxf(x, y);
    xgroup_swap32(v, x, y);
    return fmaxf(x, y);
}
F5_DEVICE float xgroup_sum(float v) {
    float x, y;
    xgroup_swap16(v, x, y);
    v = x + y;
    xgroup_swap32(v, x, y);
    return x + y;
}

// the whole KV loop of one wave that owns 16 QB queries starting at qw0
template <int QB>
F5_DEVICE void attn4_wave(const AttnArgs& p, char* smem, int wave, int lane, int head, int row0, int len, int kvlen, int qw0) {
    constexpr int NST = 5, STAGE = 16384;
    const int D = p.D;
    const int nkt = (kvlen + 63) >> 6;
    // loader role: pieces wave and wave + 8 of every KV tile (0-7 = K rows, 8-15 = V^T rows; 8 rows x 128 B each).  Physical 16-byte slot
    // (lane & 7) of row r holds logical chunk (lane & 7) ^ ((r >> 1) & 7): lds_off128 on the read side.
    const int prow = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((prow >> 1) & 7);
    const char* ksrc = reinterpret_cast<const char*>(p.qk + (size_t)(row0 + prow) * (2 * D) + D + head * 64 + chunk * 8);
    const char* vsrc = reinterpret_cast<const char*>(p.vt + (size_t)(head * 64 + prow) * p.ldvt + row0 + chunk * 8);
    const size_t kstep = (size_t)64 * (2 * D) * 2, vstep = 64 * 2;   // bytes per KV tile
    auto issue_tile = [&](int kt) {
        char* dst = smem + (kt % NST) * STAGE + wave * 1024;
        attn_lds_dma16(ksrc + kt * kstep, dst);
        attn_lds_dma16(vsrc + kt * vstep, dst + 8192);
    };
    auto wait_landed = [&](int newer) {   // this wave's pieces of a tile have landed when at most `newer` younger tiles of its own are in flight
        if (newer >= 3) attn_wait_vmcnt<6>(); else if (newer == 2) attn_wait_vmcnt<4>(); else if (newer == 1) attn_wait_vmcnt<2>(); else attn_wait_vmcnt<0>();
    };
    // ---- compute role
    const int fr = lane & 15, fq = lane >> 4;
    const float LOG2E = 1.4426950408889634f;
    // Q^T fragments (B operand: lane holds query fr of block qb, dims 32 ks + 8 fq .. + 7).  Rows beyond the sequence stay inside its
    // 128-row padding, the next sequence or the workspace slack (qk has 256 rows of it): finite data, never stored.
    bf16x8 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const __bf16* qrow = p.qk + (size_t)(row0 + qw0 + qb * 16 + fr) * (2 * D) + head * 64 + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 32);
    }
    f32x4 oacc[4][QB];
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
        for (int qb = 0; qb < QB; qb++) oacc[db][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[QB], lrun[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) { mrun[qb] = -1e30f; lrun[qb] = 0.0f; }

    // one 16-key block of S^T = K_tile Q^T from ring stage kt: s[qb][r] = score(key 16 kb + 4 fq + r, query 16 qb + fr)
    auto qk_block = [&](auto mask, f32x4 (&sk)[QB], int kt, int kb) {
        const char* kst = smem + (kt % NST) * STAGE;
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kst + lds_off128(kb * 16 + fr, fq));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kst + lds_off128(kb * 16 + fr, 4 + fq));
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            sk[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qb][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            sk[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qb][1], sk[qb], 0, 0, 0);
        }
        if (decltype(mask)::value) {   // key-padding mask (last, possibly partial tile)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool dead = kt * 64 + kb * 16 + fq * 4 + r >= kvlen;
#pragma unroll
                for (int qb = 0; qb < QB; qb++)
                    if (dead) sk[qb][r] = -1e30f;
            }
        }
    };

    // One KV tile: online softmax of the score tile s (tile kt), the score tile of tile kt + 1 computed INTO the same registers block by
    // block as soon as a block's probabilities are packed to bf16 (its MFMAs run under the exp2 stream of the following blocks; one
    // score buffer instead of two: 250 -> ~190 VGPRs at QB = 3), then O^T += V^T P^T.
    // NEXT: 0 = no next tile (last step), 1 = next tile is full, 2 = next tile is the last one (masked).  Compile-time: a run-time
    // `if (has_next)` around the block MFMAs made hipcc copy the score registers at every merge point (140 v_mov per tile).
    auto step = [&](auto next, f32x4 (&s)[4][QB], int kt) {
        constexpr int NEXT = decltype(next)::value;
        const char* vst = smem + (kt % NST) * STAGE + 8192;
        float alpha[QB], msc[QB], rs[QB];
        bool moved = false;
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
            float mloc = s[0][qb][0];
#pragma unroll
            for (int kb = 0; kb < 4; kb++)
#pragma unroll
                for (int r = 0; r < 4; r++) mloc = fmaxf(mloc, s[kb][qb][r]);
            mloc = xgroup_max(mloc);
            const float mnew = fmaxf(mrun[qb], mloc);
            moved = moved || (mnew != mrun[qb]);
            alpha[qb] = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * LOG2E);
            mrun[qb] = mnew;
            msc[qb] = -mnew * LOG2E;
            rs[qb] = 0.0f;
        }
        // pf[ks][qb]: B operand of the PV k-step ks (32 keys = key blocks 2 ks, 2 ks + 1): element j of lane group fq is key
        // 32 ks + 4 fq + j (j < 4) or 32 ks + 16 + 4 fq + j - 4
        bf16x8 pf[2][QB];
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
#pragma unroll
            for (int qb = 0; qb < QB; qb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][qb][r], LOG2E, msc[qb]));
                    rs[qb] += pv;
                    pf[kb >> 1][qb][(kb & 1) * 4 + r] = (__bf16)pv;
                }
            if constexpr (NEXT == 1) qk_block(std::false_type{}, s[kb], kt + 1, kb);
            else if constexpr (NEXT == 2) qk_block(std::true_type{}, s[kb], kt + 1, kb);
        }
#pragma unroll
        for (int qb = 0; qb < QB; qb++) lrun[qb] = lrun[qb] * alpha[qb] + rs[qb];   // per-lane partial sums (lane groups added at the end)
        if (__any(moved)) {   // wave-uniform; alpha == 1 exactly for every query whose maximum did not move
#pragma unroll
            for (int db = 0; db < 4; db++)
#pragma unroll
                for (int qb = 0; qb < QB; qb++) oacc[db][qb] *= alpha[qb];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int db = 0; db < 4; db++) {
                const int row = db * 16 + fr;
                const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vst + lds_off128(row, 4 * ks + (fq >> 1)) + (fq & 1) * 8);
                const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vst + lds_off128(row, 4 * ks + 2 + (fq >> 1)) + (fq & 1) * 8);
                bf16x8 vf;
#pragma unroll
                for (int e = 0; e < 4; e++) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
#pragma unroll
                for (int qb = 0; qb < QB; qb++) oacc[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks][qb], oacc[db][qb], 0, 0, 0);
            }
    };

    // (the Q loads above are ordinary VGPR loads: retire them before the first DMA is in flight, or hipcc's vmcnt(0) for them would
    //  land inside the loop and drain the ring every tile)
#pragma unroll
    for (int qb = 0; qb < QB; qb++) asm volatile("" ::"v"(qf[qb][0]), "v"(qf[qb][1]) : "memory");
    f32x4 s[4][QB];
#pragma unroll
    for (int t = 0; t < NST - 1; t++)
        if (t < nkt) issue_tile(t);
    wait_landed(min(3, nkt - 1));
    __builtin_amdgcn_s_barrier();                              // P: tile 0 landed
    asm volatile("" ::: "memory");
    if (nkt == 1) {
#pragma unroll
        for (int kb = 0; kb < 4; kb++) qk_block(std::true_type{}, s[kb], 0, kb);
    } else {
#pragma unroll
        for (int kb = 0; kb < 4; kb++) qk_block(std::false_type{}, s[kb], 0, kb);
    }
    auto ring = [&](int kt) {
        wait_landed(max(0, min(2, nkt - 2 - kt)));             // my pieces of tile kt + 1 (tiles kt + 2, kt + 3 may stay in flight)
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // every LDS read of step kt - 1 has returned
        __builtin_amdgcn_s_barrier();                          // R_kt: tile kt + 1 landed, stage (kt - 1) % NST is free
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < nkt) issue_tile(kt + NST - 1);
    };
    for (int kt = 0; kt + 2 < nkt; kt++) {
        ring(kt);
        step(std::integral_constant<int, 1>{}, s, kt);
    }
    if (nkt >= 2) {
        ring(nkt - 2);
        step(std::integral_constant<int, 2>{}, s, nkt - 2);
    }
    ring(nkt - 1);
    step(std::integral_constant<int, 0>{}, s, nkt - 1);

#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
        const float inv = 1.0f / xgroup_sum(lrun[qb]);
        const int q = qw0 + qb * 16 + fr;
        if (q < len) {
            const size_t obase = (size_t)(row0 + q) * D + head * 64 + fq * 4;
#pragma unroll
            for (int db = 0; db < 4; db++) {
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; e++) ov[e] = oacc[db][qb][e] * inv;
                if (p.f16_out) {
                    store_f16x4(p.out_hi + obase + db * 16, ov);
                } else {
                    bf16x4 hi4, lo4;
                    split_bf16x4(ov, hi4, lo4);
                    *reinterpret_cast<bf16x4*>(p.out_hi + obase + db * 16) = hi4;
                    if (p.out_lo) *reinterpret_cast<bf16x4*>(p.out_lo + obase + db * 16) = lo4;
                }
            }
        }
    }
}

// QA / QC: 16-query blocks per wave for waves 0-3 / waves 4-7 (two waves per SIMD: w and w + 4 share one)
template <int QA, int QC>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn4_fwd_kernel(const AttnArgs p) {
    constexpr int NST = 5, STAGE = 16384, QT = 64 * (QA + QC);
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    const int seq = blockIdx.z, head = blockIdx.y;
    const int len = p.seq_len[seq], kvlen = p.seq_kvlen[seq], row0 = p.seq_row0[seq];
    const int q0 = blockIdx.x * QT;
    if (q0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave < 4) attn4_wave<QA>(p, smem, wave, lane, head, row0, len, kvlen, q0 + wave * 16 * QA);
    else attn4_wave<QC>(p, smem, wave, lane, head, row0, len, kvlen, q0 + 64 * QA + (wave - 4) * 16 * QC);
}
