// LayerNorm of one row by one wave: shared by the stand-alone kernel (elementwise.h) and the fused prologue of gemm5 (gemm5.h).
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (biased variance, eps inside the sqrt, like torch), then
//   y = n * (gain_off + scale[c]) + shift[c]
// AdaLN-Zero: gain_off = 1, scale/shift from the modulation vector (F/model/modules.py:289,568,310);
// affine LayerNorm: gain_off = 0, scale = weight, shift = bias (ConvNeXt blocks).
// With dw_w != null the row is first replaced by a depthwise Conv1d(k=7, pad=3) over the frame axis
// (zero padding at the sequence bounds): F/model/modules.py:262 / vocos ConvNeXtBlock.
struct LnArgs {
    const float* x; int ldx; int M; int D;
    const float* scale; const float* shift; float gain_off; float eps;
    const float* dw_w; const float* dw_b; const int* row_seq_start; const int* row_seq_end;
    __bf16* out_hi; __bf16* out_lo; int ldo;
    float* out_f32; int ldof;
    int f16_out;   // 1: out_hi receives one fp16 plane (input of a PREC_F16 GEMM), out_lo unused
    int rms;   // 1: x-transformers RMSNorm, y = x / max(||x||_2, 1e-12) * sqrt(D) * scale[c]  (no mean subtraction)
};

// The row routine in three phases -- load the row, load the scale / shift vectors, reduce + normalise + store -- so a caller with
// several rows per wave can put all its loads in flight before the first use (gemm5's fused prologue: the row and the modulation
// vectors come from beyond L2 right after a kernel boundary, ~2 us each if taken one after the other).  ln_row is the three phases
// in order; the arithmetic (and its order) is the same whichever way the phases are called, so the results are bit-identical.
template <int NV>
F5_DEVICE void ln_load(const LnArgs& p, const int row, const int lane, float4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.D && row < p.M) {
            if (p.dw_w) {
                const int s0 = p.row_seq_start[row], s1 = p.row_seq_end[row];
                float4 a = *reinterpret_cast<const float4*>(p.dw_b + c);
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int r = row + k - 3;
                    if (r >= s0 && r < s1) {
                        const float4 xv = *reinterpret_cast<const float4*>(p.x + (size_t)r * p.ldx + c);
                        a.x += p.dw_w[(c + 0) * 7 + k] * xv.x;
                        a.y += p.dw_w[(c + 1) * 7 + k] * xv.y;
                        a.z += p.dw_w[(c + 2) * 7 + k] * xv.z;
                        a.w += p.dw_w[(c + 3) * 7 + k] * xv.w;
                    }
                }
                v[i] = a;
            } else {
                v[i] = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + c);
            }
        }
    }
}

template <int NV>
F5_DEVICE void ln_load_mod(const LnArgs& p, const int lane, float4 (&sc)[NV], float4 (&sh)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        sc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        sh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < p.D) {
            sc[i] = *reinterpret_cast<const float4*>(p.scale + c);
            sh[i] = *reinterpret_cast<const float4*>(p.shift + c);
        }
    }
}

template <int NV>
F5_DEVICE void ln_finish(const LnArgs& p, const int row, const int lane, const float4 (&v)[NV], const float4 (&sc)[NV], const float4 (&sh)[NV]) {
    if (row >= p.M) return;   // (wave-uniform)
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.D) sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = p.rms ? 0.0f : wave_sum(sum) / (float)p.D;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.D) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += a * a + b * b + cc * cc + d * d;
        }
    }
    const float sqt = wave_sum(sq);
    const float rstd = p.rms ? sqrtf((float)p.D) / fmaxf(sqrtf(sqt), 1e-12f) : rsqrtf(sqt / (float)p.D + p.eps);
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.D) {
            float y[4];
            y[0] = (v[i].x - mean) * rstd * (p.gain_off + sc[i].x) + sh[i].x;
            y[1] = (v[i].y - mean) * rstd * (p.gain_off + sc[i].y) + sh[i].y;
            y[2] = (v[i].z - mean) * rstd * (p.gain_off + sc[i].z) + sh[i].z;
            y[3] = (v[i].w - mean) * rstd * (p.gain_off + sc[i].w) + sh[i].w;
            if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (size_t)row * p.ldof + c) = make_float4(y[0], y[1], y[2], y[3]);
            if (p.out_hi && p.f16_out) {
                store_f16x4(p.out_hi + (size_t)row * p.ldo + c, y);
            } else if (p.out_hi) {
                bf16x4 hi, lo;
                split_bf16x4(y, hi, lo);
                *reinterpret_cast<bf16x4*>(p.out_hi + (size_t)row * p.ldo + c) = hi;
                if (p.out_lo) *reinterpret_cast<bf16x4*>(p.out_lo + (size_t)row * p.ldo + c) = lo;
            }
        }
    }
}

// one row by one wave (64 lanes x NV float4)
template <int NV>
F5_DEVICE void ln_row(const LnArgs& p, const int row, const int lane) {
    if (row >= p.M) return;
    float4 v[NV], sc[NV], sh[NV];
    ln_load<NV>(p, row, lane, v);
    ln_load_mod<NV>(p, lane, sc, sh);
    ln_finish<NV>(p, row, lane, v, sc, sh);
}
