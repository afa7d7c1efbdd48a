// attn2: 8-wave, LDS-DMA-fed version of the flash attention forward (same math and fragment scheme as attn.h).
//
// rocprof on attn.h (4 waves, 128 queries, register-staged K/V with one tile in flight) showed ~2500 cycles per KV
// tile against 512 MFMA cycles per wave: the loop waits on the ~1 us global->LDS latency every tile, and at one
// utterance per GPU its 352 workgroups need two rounds on 256 CUs.  Here one 512-thread workgroup owns 256 queries
// (192 workgroups at N = 1404: one round), the K and V^T tiles of 64 keys arrive by LDS-DMA (global_load_lds_dwordx4,
// 2 pieces per wave per tile) into a 4-deep ring (3 tiles = 48 KiB in flight), with a counted s_waitcnt vmcnt and one
// raw s_barrier per tile; the O rescale is skipped whenever no running maximum of the wave moved (exact: alpha == 1).
#pragma once
#include "attn.h"

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn2_fwd_kernel(const AttnArgs p) {
    constexpr int NST = 4, STAGE = 16384;
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    const int seq = blockIdx.z, head = blockIdx.y;
    const int len = p.seq_len[seq], kvlen = p.seq_kvlen[seq], row0 = p.seq_row0[seq];
    const int q0 = blockIdx.x * 256;
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

    // LDS-DMA: a KV tile is 8 K pieces + 8 V^T pieces of 1 KiB (8 rows x 128 B); wave w moves K piece w and V piece w.
    // Physical 16-B slot (lane & 7) of row r holds logical chunk (lane & 7) ^ ((r >> 1) & 7)  (same swizzle as attn.h).
    const int prow = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((prow >> 1) & 7);
    const char* ksrc = reinterpret_cast<const char*>(p.qk + (size_t)(row0 + prow) * (2 * D) + D + head * 64 + chunk * 8);
    const char* vsrc = reinterpret_cast<const char*>(p.vt + (size_t)(head * 64 + prow) * p.ldvt + row0 + chunk * 8);
    const size_t kstep = (size_t)64 * (2 * D) * 2, vstep = 64 * 2;   // bytes per KV tile
    auto issue_tile = [&](int kt) {
        char* dst = smem + (kt % NST) * STAGE + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + kt * kstep),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + kt * vstep),
                                         (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
    };

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int g = 0; g < 16; g++) oacc[dt][g] = 0.0f;
    float mrun = -1e30f, lrun = 0.0f;

    const int nkt = (kvlen + 63) >> 6;
#pragma unroll
    for (int t = 0; t < NST - 1; t++)
        if (t < nkt) issue_tile(t);

    for (int kt = 0; kt < nkt; kt++) {
        const int newer = min(NST - 2, nkt - 1 - kt);
        if (newer >= 2) attn_wait_vmcnt<4>();
        else if (newer == 1) attn_wait_vmcnt<2>();
        else attn_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nkt) issue_tile(kt + NST - 1);
        const char* kst = smem + (kt % NST) * STAGE;
        const char* vst = kst + 8192;

        f32x16 sacc[2];
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
#pragma unroll
            for (int g = 0; g < 16; g++) sacc[kh][g] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(kst + lds_off128(kh * 32 + fr, 2 * s + fh));
                sacc[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kh], 0, 0, 0);
            }
        }
        if (kt * 64 + 64 > kvlen) {   // last, partial tile: key-padding mask
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int key = kt * 64 + kh * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh;
                    if (key >= kvlen) sacc[kh][g] = -1e30f;
                }
        }
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
        const float msc = mnew * LOG2E;
        float rowsum = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kh][g], LOG2E, -msc));
                sacc[kh][g] = pv;
                rowsum += pv;
            }
        lrun = lrun * alpha + rowsum;
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
