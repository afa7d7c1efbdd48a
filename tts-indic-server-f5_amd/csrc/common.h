// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libf5hip.
// Wave = 64 lanes everywhere; bf16 MFMA fragments are 8 x bf16 = one 16-byte register quad.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

#define F5_DEVICE __device__ __forceinline__

// q leaves the QKV epilogue scaled by the softmax scale 1/8 AND log2(e), so an attention score is a base-2 exponent (attn3.h)
#define F5_Q_SCALE (0.125f * 1.4426950408889634f)
// V^T [feature][token] keeps the tokens of every aligned group of 16 in the order 0-3, 8-11, 4-7, 12-15 (bits 2 and 3 of the token index
// swapped): the 8 tokens one lane feeds to a 32 x 32 x 16 PV MFMA -- rows {4 fh .. 4 fh + 3} and {8 + 4 fh ..} of the score block it holds --
// are then 16 contiguous bytes, one ds_read_b128 instead of two 8-byte reads and a register shuffle (attn3.h).  Writers store 4-token groups.
F5_DEVICE int vt_col(int tok) { return (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1); }

// fp32 -> (hi, lo) bf16 pair with hi = rn(x), lo = rn(x - hi): x ~= hi + lo to ~16 mantissa bits.
F5_DEVICE void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

F5_DEVICE void split_bf16x4(const float* y, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        __bf16 h, l;
        split_bf16(y[e], h, l);
        hi[e] = h;
        lo[e] = l;
    }
}

// fp32 -> fp16 with saturation at the largest finite fp16 (one v_med3_f32): the reference runs these activations in fp32
// (F/infer/utils_infer.py:176-184 -- the fp16 branch there is commented out), so an outlier beyond 65504 must degrade to a clamp,
// never to inf (inf x 0 = NaN would poison the whole output row of the next GEMM).  NaN stays NaN.
F5_DEVICE _Float16 sat_f16(float x) { return (_Float16)__builtin_fminf(__builtin_fmaxf(x, -65504.0f), 65504.0f); }

// fp32 x 4 -> fp16 x 4 (round to nearest even, saturating), stored through a 2-byte-element pointer shared with the bf16 planes
F5_DEVICE void store_f16x4(__bf16* dst, const float* y) {
    f16x4 h;
#pragma unroll
    for (int e = 0; e < 4; e++) h[e] = sat_f16(y[e]);
    *reinterpret_cast<f16x4*>(dst) = h;
}

// one 32 x 32 x 16 MFMA on 16-byte fragments: bf16 operands, or (F16) the same bits read as fp16
template <bool F16>
F5_DEVICE f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Activations on the hardware transcendental units (v_exp_f32 / v_rcp_f32 / v_log_f32, ~1 ulp each): the libm forms
// (tanhf, expf, log1pf) expand to 25-40 instructions per element and dominated the GEMM epilogues.
F5_DEVICE float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
F5_DEVICE float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

F5_DEVICE float gelu_tanh_f(float x) {
    // torch GELU(approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3);  0.5 (1 + tanh u) = 1 / (1 + e^{-2u})
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    return x * fast_rcp(1.0f + fast_exp(-2.0f * u));
}
F5_DEVICE float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
F5_DEVICE float silu_f(float x) { return x * fast_rcp(1.0f + fast_exp(-x)); }
F5_DEVICE float mish_f(float x) {
    // x * tanh(softplus(x));  with e = e^x: tanh(log(1 + e)) = (e^2 + 2e) / (e^2 + 2e + 2);  x > 20: tanh(softplus) == 1 in fp32
    const float e = fast_exp(fminf(x, 20.0f));
    const float n = e * (e + 2.0f);
    return x * n * fast_rcp(n + 2.0f);
}

F5_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
F5_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_MISH = 3, ACT_SILU = 4 };

F5_DEVICE float apply_act(float v, int act) {
    switch (act) {
        case ACT_GELU_TANH: return gelu_tanh_f(v);
        case ACT_GELU_ERF: return gelu_erf_f(v);
        case ACT_MISH: return mish_f(v);
        case ACT_SILU: return silu_f(v);
        default: return v;
    }
}
