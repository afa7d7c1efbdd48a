// Flash-style non-causal self-attention forward for gfx950, head dim 64, bf16 MFMA, fp32 online softmax.
//
//  one workgroup = 128 queries of one (sequence, head); 4 waves x 32 queries; KV tiles of 64 keys.
//  Per KV tile each wave computes S^T = K Q^T (keys on accumulator rows, queries on lanes), so the row max /
//  row sum of a query are in-register reductions plus one lane^32 exchange, and the exponentiated tile is
//  already the B operand of O^T += V^T P^T (guide: "an accumulator tile as the next MFMA's operand") -- no
//  LDS round trip for P.  V arrives transposed ([d][token], written by the QKV GEMM epilogue) so the permuted
//  k order of that operand is two 8-byte LDS reads.  Q is pre-scaled by 1/8 in the QKV epilogue.
//  Keys >= kv_len are masked to -1e30 before the max (key-padding mask, F/model/modules.py:429-434).
#pragma once
#include "../attn_common.h"

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];
    const int seq = blockIdx.z, head = blockIdx.y;
    const int len = p.seq_len[seq], kvlen = p.seq_kvlen[seq], row0 = p.seq_row0[seq];
    const int q0 = blockIdx.x * 128;
    if (q0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int D = p.D;
    const float LOG2E = 1.4426950408889634f;

    bf16x8 qf[4];
    {
        const __bf16* qrow = p.qk + (size_t)(row0 + q0 + wave * 32 + fr) * (2 * D) + head * 64 + fh * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + s * 16);
    }

    const int nkt = (kvlen + 63) >> 6;
    u32x4 rk[2], rv[2];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + 256 * i, r = idx >> 3, c = idx & 7;
            rk[i] = *reinterpret_cast<const u32x4*>(p.qk + (size_t)(row0 + kt * 64 + r) * (2 * D) + D + head * 64 + c * 8);
            rv[i] = *reinterpret_cast<const u32x4*>(p.vt + (size_t)(head * 64 + r) * p.ldvt + row0 + kt * 64 + c * 8);
        }
    };
    auto store_kv = [&](int stage) {
        char* base = smem + stage * 16384;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + 256 * i, r = idx >> 3, c = idx & 7;
            *reinterpret_cast<u32x4*>(base + lds_off128(r, c)) = rk[i];
            *reinterpret_cast<u32x4*>(base + 8192 + lds_off128(r, c)) = rv[i];
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
        for (int g = 0; g < 16; g++) oacc[dt][g] = 0.0f;
    float mrun = -1e30f, lrun = 0.0f;

    load_kv(0);
    store_kv(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; kt++) {
        const bool more = kt + 1 < nkt;
        if (more) load_kv(kt + 1);
        const char* kst = smem + (kt & 1) * 16384;
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
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g = 0; g < 16; g++) oacc[dt][g] *= alpha;

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
        if (more) store_kv((kt + 1) & 1);
        __syncthreads();
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
