// BigVGAN v2 generator (NVIDIA/BigVGAN `bigvgan_v2_24khz_100band_256x` geometry) on the split-bf16
// implicit-GEMM conv kernel (gemm.h) plus three HBM-streaming kernels.  Included at the end of f5hip.hip.
//
// Layout: channel-last rows, one row per time step, batch items are "sequences" of pitch P_i = ceil128(T) * prod(rates so far)
// rows (T_i = T * prod valid), so a ConvTranspose1d with stride r writes [rows_in][r * C_out] == [rows_in * r][C_out].
//   conv_pre / resblock convs : Conv1d as implicit GEMM, K = taps x C_pad32, dilation = row shift per tap
//   ups[i] (k = 2r, stride r, pad r/2): 3-tap implicit GEMM over inputs t-1, t, t+1 with N = r * C_out (phase-major
//       columns); phase p uses taps (t, t-1) if p < r/2 else (t, t+1) -- the third tap's weights are zero
//   Activation1d(SnakeBeta): fused [2x Kaiser-sinc upsample -> x + sin^2(x e^a)/(e^b + 1e-9) -> 2x low-pass downsample], one lane
//       per (channel, run of 16 time steps), everything in registers (aa_snake2_kernel)
// Operand precision (f5hip_bigvgan_config.gemm_planes): 2 = split bf16 (three MFMAs per product, the parity default), 3 = one fp16
//       plane (a third of the MFMA work and half the activation-plane traffic), 1 = plain bf16.
#pragma once

struct BvConv { PackedW w; int k = 0, dil = 1, c_in = 0, c_out = 0, c_in_pad = 0; };
struct BvRes { BvConv c1[3], c2[3]; float* alpha[6] = {}; float* beta[6] = {}; };

struct f5hip_bigvgan {
    f5hip_bigvgan_config cfg;
    int nsplit = 2;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    int n_up = 0, c0 = 0;
    BvConv pre;
    std::vector<BvConv> ups;
    std::vector<BvRes> res;
    float *post_alpha = nullptr, *post_beta = nullptr, *post_w = nullptr, *filt = nullptr;
    // workspace
    size_t cap = 0;
    void* ws = nullptr;
    float *X = nullptr, *Y[3] = {}, *S = nullptr, *Tm = nullptr;
    Plane2 act, melp;
    float filt_h[12] = {};
};

// ------------------------------------------------------------------------------------------------ kernels
// Anti-aliased SnakeBeta activation (alias_free_torch Activation1d, up = down = 2, 12 taps).  x fp32 [rows][ldx];
// output split bf16 [rows][ldo] (conv A operand) or fp32 [rows][ldo].  Uniform sequences: pitch P rows, T valid.
__global__ __launch_bounds__(256) void aa_snake_kernel(const float* x, int ldx, int C, int P, int T, const float* alpha_log,
                                                       const float* beta_log, const float* filt, __bf16* out_hi, __bf16* out_lo,
                                                       float* out_f32, int ldo) {
    __shared__ float xs[76][64];
    __shared__ float as[138][64];
    __shared__ float f[12];
    const int tid = threadIdx.x, cl = tid & 63, tg = tid >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int row0 = blockIdx.y * 64;
    const int seq0 = (row0 / P) * P, t0 = row0 - seq0;
    if (t0 >= T) return;
    if (tid < 12) f[tid] = filt[tid];
    const bool cok = c < C;
    for (int r = tg; r < 76; r += 4) {
        int ti = t0 - 6 + r;
        ti = ti < 0 ? 0 : (ti > T - 1 ? T - 1 : ti);   // replicate padding of the up-sampler
        xs[r][cl] = cok ? x[(size_t)(seq0 + ti) * ldx + c] : 0.0f;
    }
    __syncthreads();
    const float ea = cok ? expf(alpha_log[c]) : 1.0f;
    const float ib = cok ? 1.0f / (expf(beta_log[c]) + 1e-9f) : 0.0f;
    for (int r = tg; r < 138; r += 4) {
        int j = 2 * t0 - 5 + r;
        j = j < 0 ? 0 : (j > 2 * T - 1 ? 2 * T - 1 : j);   // replicate padding of the down-sampler
        const int t = j >> 1, odd = j & 1;
        // up[j] = 2 * sum_q x[t - 3 + odd + q] * f[11 - odd - 2 q]
        const int base = t - 3 + odd - (t0 - 6);
        float u = 0.0f;
#pragma unroll
        for (int q = 0; q < 6; q++) u += xs[base + q][cl] * f[11 - odd - 2 * q];
        u *= 2.0f;
        // sin on the hardware unit (v_sin_f32 takes revolutions): fp32 range reduction r = a / 2pi - rint(a / 2pi) keeps |a| < ~1e3 rad
        // within ~1e-5 rad, far inside the 1e-4 waveform bound; libm sinf was ~half of this kernel's time
        const float rev = u * ea * 0.15915494309189535f;
        const float sn = __builtin_amdgcn_sinf(rev - rintf(rev));
        as[r][cl] = u + ib * sn * sn;
    }
    __syncthreads();
    for (int tt = tg; tt < 64; tt += 4) {
        const int t = t0 + tt;
        if (t >= T || !cok) continue;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 12; k++) v += as[2 * tt + k][cl] * f[k];
        const size_t o = (size_t)(seq0 + t) * ldo + c;
        if (out_f32) out_f32[o] = v;
        if (out_hi) {
            __bf16 h, l;
            split_bf16(v, h, l);
            out_hi[o] = h;
            out_lo[o] = l;
        }
    }
}

struct AaFilt { float f[12]; };

// Same operator as aa_snake_kernel, register-resident: a lane owns one channel and R consecutive time steps.  It loads the R + 10
// inputs its outputs depend on (row index clamped = the replicate padding of the up-sampler; for a fixed register the lanes of a
// segment read consecutive channels of one row, so the loads coalesce), forms the 2R + 10 up-sampled snake values (index clamped
// to [0, 2T) = the replicate padding of the down-sampler: values past the end repeat the last one, values before 0 repeat value 0)
// and the R low-passed outputs.  No LDS, no barrier: ~44 VALU operations and 2.6 v_sin per output instead of ~25 LDS reads.
//   grid (C / cw, ceil(T / (nseg R)), sequences), 256 lanes = nseg segments x cw channels (cw | C, cw <= 64)
//   OUT: 0 = fp32, 1 = split bf16 planes, 2 = one fp16 plane
template <int R, int OUT>
__global__ __launch_bounds__(256) void aa_snake2_kernel(const float* __restrict__ x, int ldx, int C, int cw, int nseg, int P, int T,
                                                        const float* __restrict__ alpha_log, const float* __restrict__ beta_log, AaFilt flt,
                                                        __bf16* __restrict__ out_hi, __bf16* __restrict__ out_lo, float* __restrict__ out_f32, int ldo) {
    const int tid = threadIdx.x;
    const int seg = tid / cw, c = blockIdx.x * cw + (tid - seg * cw);
    const int t0 = (blockIdx.y * nseg + seg) * R;
    if (seg >= nseg || t0 >= T || c >= C) return;
    const size_t seq0 = (size_t)blockIdx.z * P;
    const float* xb = x + seq0 * ldx + c;
    float xv[R + 10];
#pragma unroll
    for (int i = 0; i < R + 10; i++) {
        int ti = t0 - 5 + i;
        ti = ti < 0 ? 0 : (ti > T - 1 ? T - 1 : ti);
        xv[i] = xb[(size_t)ti * ldx];
    }
    const float ea = expf(alpha_log[c]) * 0.15915494309189535f;   // radians -> revolutions for v_sin_f32
    const float ib = 1.0f / (expf(beta_log[c]) + 1e-9f);
    float f2[12];
#pragma unroll
    for (int k = 0; k < 12; k++) f2[k] = 2.0f * flt.f[k];   // ratio * conv_transpose1d
    const int lim = 2 * (T - t0) + 4;   // up-sampled index j = 2 t0 - 5 + i is inside [0, 2T) for 5 - 2 t0 <= i <= lim
    float a[2 * R + 10];
    float prev = 0.0f;
#pragma unroll
    for (int i = 0; i < 2 * R + 10; i++) {
        const int i0 = i >> 1, odd = (i & 1) ^ 1;   // j odd <=> i even
        float u = 0.0f;
#pragma unroll
        for (int q = 0; q < 6; q++) u += xv[i0 + q] * f2[11 - odd - 2 * q];
        const float rev = u * ea;
        const float sn = __builtin_amdgcn_sinf(rev - rintf(rev));
        float av = u + ib * sn * sn;
        av = i > lim ? prev : av;
        prev = av;
        a[i] = av;
    }
    if (t0 == 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] = a[5];
    }
#pragma unroll
    for (int tt = 0; tt < R; tt++) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 12; k++) v += a[2 * tt + k] * flt.f[k];
        if (t0 + tt < T) {
            const size_t o = (seq0 + t0 + tt) * ldo + c;
            if constexpr (OUT == 0) out_f32[o] = v;
            if constexpr (OUT == 1) {
                __bf16 h, l;
                split_bf16(v, h, l);
                out_hi[o] = h;
                out_lo[o] = l;
            }
            if constexpr (OUT == 2) reinterpret_cast<_Float16*>(out_hi)[o] = sat_f16(v);
        }
    }
}

// mean of the three AMP blocks of a stage (n_in == 3; n_in == 1 is a plain conversion of y0), 4 channels per lane: (y0 + y1 + y2) / 3 ->
// fp32 rows and / or the operand planes of the next up-sampler (mode 1 split bf16, 2 fp16, 3 plain bf16; ldo >= C, padding channels
// are left as they are)
__global__ __launch_bounds__(256) void bv_mean3_kernel(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2, int n_in, size_t rows,
                                                       int C, float* __restrict__ out_f32, __bf16* __restrict__ out_hi, __bf16* __restrict__ out_lo, int ldo, int mode) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C >> 2;
    if (i >= rows * c4) return;
    const size_t row = i / c4;
    const int c = (int)(i - row * c4) * 4;
    f32x4 s = reinterpret_cast<const f32x4*>(y0)[i];
    if (n_in == 3) s = (s + reinterpret_cast<const f32x4*>(y1)[i] + reinterpret_cast<const f32x4*>(y2)[i]) * (1.0f / 3.0f);
    if (out_f32) reinterpret_cast<f32x4*>(out_f32)[i] = s;
    if (mode == 0) return;
    const float y[4] = {s[0], s[1], s[2], s[3]};
    __bf16* dh = out_hi + row * ldo + c;
    if (mode == 2) { store_f16x4(dh, y); return; }
    bf16x4 h, l;
    split_bf16x4(y, h, l);
    *reinterpret_cast<bf16x4*>(dh) = h;
    if (mode == 1) *reinterpret_cast<bf16x4*>(out_lo + row * ldo + c) = l;
}

// mel [B][C][T] fp32 -> rows (b * P + t) of 128 split-bf16 channels (rows >= T and channels >= C are zero)
__global__ __launch_bounds__(128) void bv_mel_rows_kernel(const float* mel, int C, int T, int P, __bf16* hi, __bf16* lo, int f16) {
    const int row = blockIdx.x, c = threadIdx.x;
    const int b = row / P, t = row - b * P;
    float v = 0.0f;
    if (t < T && c < C) v = mel[((size_t)b * C + c) * T + t];
    if (f16) { reinterpret_cast<_Float16*>(hi)[(size_t)row * 128 + c] = sat_f16(v); return; }
    __bf16 h, l;
    split_bf16(v, h, l);
    hi[(size_t)row * 128 + c] = h;
    if (lo) lo[(size_t)row * 128 + c] = l;
}

// conv_post: Conv1d(C -> 1, k = 7, pad 3, no bias) + clamp(-1, 1);  a fp32 [rows][lda] -> wave [B][T].  256 outputs per workgroup: the
// 262 input rows go through LDS (row pitch C + 1 floats: the lanes of a wave read consecutive rows, an odd pitch is conflict-free),
// the weights are read as broadcasts.  Dynamic LDS = (262 (C + 1) + 7 C) floats.
__global__ __launch_bounds__(256) void bv_conv_post_kernel(const float* __restrict__ a, int lda, int C, int P, int T, const float* __restrict__ w /*[C][7]*/,
                                                           float* __restrict__ wave) {
    extern __shared__ float bvp_sm[];
    float* tile = bvp_sm;
    float* ws = bvp_sm + 262 * (C + 1);
    const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    for (int i = tid; i < 262 * C; i += 256) {
        const int r = i / C, c = i - r * C, ti = t0 - 3 + r;
        tile[r * (C + 1) + c] = (ti >= 0 && ti < T) ? a[(size_t)(b * P + ti) * lda + c] : 0.0f;
    }
    for (int i = tid; i < 7 * C; i += 256) {
        const int k = i / C, c = i - k * C;
        ws[i] = w[c * 7 + k];
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= T) return;
    float acc = 0.0f;
    for (int k = 0; k < 7; k++) {
        const float* row = tile + (tid + k) * (C + 1);
        const float* wk = ws + k * C;
        for (int c = 0; c < C; c++) acc += wk[c] * row[c];
    }
    wave[(size_t)b * T + t] = fminf(fmaxf(acc, -1.0f), 1.0f);
}

// the same operator without the LDS tile (channel counts whose tile would not fit)
__global__ __launch_bounds__(256) void bv_conv_post_naive_kernel(const float* a, int lda, int C, int P, int T, const float* w /*[C][7]*/, float* wave) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float acc = 0.0f;
    for (int k = 0; k < 7; k++) {
        const int ti = t + k - 3;
        if (ti < 0 || ti >= T) continue;
        const float* row = a + (size_t)(b * P + ti) * lda;
        for (int c = 0; c < C; c++) acc += w[c * 7 + k] * row[c];
    }
    wave[(size_t)b * T + t] = fminf(fmaxf(acc, -1.0f), 1.0f);
}

// ------------------------------------------------------------------------------------------------ host side
f5hip_bigvgan* f5hip_bigvgan_create(const f5hip_bigvgan_config* cfg) {
    if (!cfg) { set_error("null config"); return nullptr; }
    if (cfg->num_upsamples < 1 || cfg->num_upsamples > 8 || cfg->num_mels > 128 || cfg->upsample_initial_channel % (1 << cfg->num_upsamples) ||
        (cfg->gemm_planes < 1 || cfg->gemm_planes > 3)) { set_error("unsupported BigVGAN geometry"); return nullptr; }
    for (int i = 0; i < cfg->num_upsamples; i++)
        if (cfg->upsample_kernel_sizes[i] != 2 * cfg->upsample_rates[i] || cfg->upsample_rates[i] % 2) {
            set_error("BigVGAN: only kernel = 2 * stride, even stride up-samplers are supported"); return nullptr;
        }
    if ((cfg->upsample_initial_channel >> cfg->num_upsamples) % 4) { set_error("BigVGAN: final channel count must be a multiple of 4"); return nullptr; }
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) { set_error("no HIP device: libf5hip has no CPU fallback"); return nullptr; }
    f5hip_bigvgan* v = new f5hip_bigvgan();
    v->cfg = *cfg; v->nsplit = cfg->gemm_planes; v->n_up = cfg->num_upsamples; v->c0 = cfg->upsample_initial_channel;
    return v;
}

static void bv_free_conv(BvConv& c) { free_packed(c.w); }

void f5hip_bigvgan_destroy(f5hip_bigvgan* v) {
    if (!v) return;
    bv_free_conv(v->pre);
    for (auto& u : v->ups) bv_free_conv(u);
    for (auto& r : v->res) {
        for (int j = 0; j < 3; j++) { bv_free_conv(r.c1[j]); bv_free_conv(r.c2[j]); }
        for (int a = 0; a < 6; a++) { dev_free(r.alpha[a]); dev_free(r.beta[a]); }
    }
    for (float* p : {v->post_alpha, v->post_beta, v->post_w, v->filt}) dev_free(p);
    dev_free(v->ws);
    delete v;
}

int f5hip_bigvgan_load_param(f5hip_bigvgan* v, const char* name, const float* data, int64_t numel) {
    if (!v || !name || !data || numel <= 0) return fail(-1, "load_param: bad argument");
    if (v->finalized) return fail(-2, "load_param after finalize");
    v->host[name].assign(data, data + numel);
    return 0;
}

#define BGETP(var, name, numel)                                                                                    \
    const std::vector<float>* var = nullptr;                                                                        \
    {                                                                                                               \
        auto it = v->host.find(name);                                                                               \
        if (it == v->host.end()) return fail(-3, "missing parameter %s", std::string(name).c_str());                \
        if ((int64_t)it->second.size() != (int64_t)(numel)) return fail(-3, "parameter %s: wrong size", std::string(name).c_str()); \
        var = &it->second;                                                                                          \
    }

// Conv1d weight [co][ci][k] -> [co][tap][ci_pad]
static int bv_pack_conv(BvConv& c, const std::vector<float>& w, const float* bias, int co, int ci, int k, int dil, bool f16) {
    c.k = k; c.dil = dil; c.c_in = ci; c.c_out = co; c.c_in_pad = ceil_to(ci, 32);
    const int K = k * c.c_in_pad;
    std::vector<float> wp((size_t)co * K, 0.0f);
    for (int o = 0; o < co; o++)
        for (int i = 0; i < ci; i++)
            for (int t = 0; t < k; t++) wp[(size_t)o * K + t * c.c_in_pad + i] = w[((size_t)o * ci + i) * k + t];
    return pack_linear(c.w, wp.data(), co, K, K, bias, co <= 64 ? 64 : 128, f16);
}

int f5hip_bigvgan_finalize(f5hip_bigvgan* v) {
    if (!v) return fail(-1, "null vocoder");
    if (v->finalized) return 0;
    const f5hip_bigvgan_config& c = v->cfg;
    const bool f16 = v->nsplit == 3;
    {
        BGETP(w, "conv_pre.weight", (int64_t)v->c0 * c.num_mels * 7); BGETP(b, "conv_pre.bias", v->c0);
        // input rows are mel frames padded to 128 channels
        BvConv& p = v->pre; p.k = 7; p.dil = 1; p.c_in = c.num_mels; p.c_out = v->c0; p.c_in_pad = 128;
        std::vector<float> wp((size_t)v->c0 * 7 * 128, 0.0f);
        for (int o = 0; o < v->c0; o++)
            for (int i = 0; i < c.num_mels; i++)
                for (int t = 0; t < 7; t++) wp[(size_t)o * 896 + t * 128 + i] = (*w)[((size_t)o * c.num_mels + i) * 7 + t];
        if (pack_linear(p.w, wp.data(), v->c0, 896, 896, b->data(), 128, f16)) return -4;
    }
    v->ups.resize(v->n_up);
    v->res.resize(v->n_up * 3);
    for (int i = 0; i < v->n_up; i++) {
        const int ci = v->c0 >> i, co = v->c0 >> (i + 1), r = c.upsample_rates[i], k = 2 * r;
        BGETP(w, "ups." + std::to_string(i) + ".0.weight", (int64_t)ci * co * k);
        BGETP(b, "ups." + std::to_string(i) + ".0.bias", co);
        BvConv& u = v->ups[i]; u.k = 3; u.dil = 1; u.c_in = ci; u.c_out = r * co; u.c_in_pad = ceil_to(ci, 32);
        const int K = 3 * u.c_in_pad;
        std::vector<float> wp((size_t)r * co * K, 0.0f), bp((size_t)r * co);
        for (int p = 0; p < r; p++)
            for (int o = 0; o < co; o++) {
                const size_t n = (size_t)p * co + o;
                bp[n] = (*b)[o];
                for (int i2 = 0; i2 < ci; i2++) {
                    const float* wr = &(*w)[((size_t)i2 * co + o) * k];   // ConvTranspose1d weight [c_in][c_out][k]
                    wp[n * K + 1 * u.c_in_pad + i2] = wr[p + r / 2];                       // input t
                    if (p < r / 2) wp[n * K + 0 * u.c_in_pad + i2] = wr[p + 3 * r / 2];   // input t - 1
                    else wp[n * K + 2 * u.c_in_pad + i2] = wr[p - r / 2];                  // input t + 1
                }
            }
        if (pack_linear(u.w, wp.data(), r * co, K, K, bp.data(), r * co <= 64 ? 64 : 128, f16)) return -4;
        for (int j = 0; j < 3; j++) {
            BvRes& rb = v->res[i * 3 + j];
            const int kk = c.resblock_kernel_sizes[j];
            const std::string q = "resblocks." + std::to_string(i * 3 + j) + ".";
            for (int d = 0; d < 3; d++) {
                BGETP(w1, q + "convs1." + std::to_string(d) + ".weight", (int64_t)co * co * kk); BGETP(b1, q + "convs1." + std::to_string(d) + ".bias", co);
                BGETP(w2, q + "convs2." + std::to_string(d) + ".weight", (int64_t)co * co * kk); BGETP(b2, q + "convs2." + std::to_string(d) + ".bias", co);
                if (bv_pack_conv(rb.c1[d], *w1, b1->data(), co, co, kk, c.resblock_dilations[j * 3 + d], f16)) return -4;
                if (bv_pack_conv(rb.c2[d], *w2, b2->data(), co, co, kk, 1, f16)) return -4;
            }
            for (int a = 0; a < 6; a++) {
                BGETP(al, q + "activations." + std::to_string(a) + ".act.alpha", co); BGETP(be, q + "activations." + std::to_string(a) + ".act.beta", co);
                if (upload_f32(&rb.alpha[a], al->data(), co) || upload_f32(&rb.beta[a], be->data(), co)) return -4;
            }
        }
    }
    {
        const int ch = v->c0 >> v->n_up;
        BGETP(al, "activation_post.act.alpha", ch); BGETP(be, "activation_post.act.beta", ch); BGETP(w, "conv_post.weight", (int64_t)ch * 7);
        if (upload_f32(&v->post_alpha, al->data(), ch) || upload_f32(&v->post_beta, be->data(), ch) || upload_f32(&v->post_w, w->data(), ch * 7)) return -4;
    }
    {   // kaiser_sinc_filter1d(cutoff 0.25, half_width 0.3, 12): alias_free_torch/filter.py
        const int ks = 12, half = 6;
        const double cutoff = 0.25, hw = 0.3, delta_f = 4 * hw, A = 2.285 * (half - 1) * M_PI * delta_f + 7.95;
        const double beta = A > 50.0 ? 0.1102 * (A - 8.7) : (A >= 21.0 ? 0.5842 * pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0) : 0.0);
        auto i0 = [](double x) { double s = 1.0, t = 1.0; for (int k = 1; k < 50; k++) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; } return s; };
        double f[12], sum = 0.0;
        for (int n = 0; n < ks; n++) {
            const double r = 2.0 * n / (ks - 1) - 1.0;                       // torch.kaiser_window(periodic=False)
            const double win = i0(beta * sqrt(1.0 - r * r)) / i0(beta);
            const double tm = (n - half) + 0.5, xx = 2 * cutoff * tm;
            const double sinc = fabs(xx) < 1e-12 ? 1.0 : sin(M_PI * xx) / (M_PI * xx);
            f[n] = 2 * cutoff * win * sinc;
            sum += f[n];
        }
        float ff[12];
        for (int n = 0; n < ks; n++) ff[n] = (float)(f[n] / sum);
        if (upload_f32(&v->filt, ff, 12)) return -4;
        memcpy(v->filt_h, ff, sizeof(ff));
    }
    v->host.clear();
    v->finalized = true;
    return 0;
}

static int bv_conv(f5hip_bigvgan* v, const BvConv& c, const Plane2& A, int M, int P, int T, int act, const float* res, float* out, int ldo,
                   hipStream_t st) {
    GemmArgs g = gemm_base(A, c.c_in_pad, c.w, M);
    g.conv_kpt = c.c_in_pad / 32; g.conv_center = (c.k - 1) / 2; g.conv_dil = c.dil; g.conv_group_cols = 0;
    g.row_seq_start = nullptr; g.row_seq_end = nullptr; g.seq_pitch = P; g.seq_valid = T;
    g.act = act; g.res = res; g.ldres = ldo; g.out_f32 = out; g.ldo = ldo;
    // conv5.h (window of the tile once in LDS, taps served from it) where it covers the shape; F5HIP_CONV5=0 keeps everything on gemm.h (A/B)
    static const int use_conv5 = getenv("F5HIP_CONV5") ? atoi(getenv("F5HIP_CONV5")) : 1;
    if (use_conv5 && v->nsplit >= 2) {
        prof_begin(PROF_GEMM, st);
        const hipError_t e = f5_launch_conv5(v->nsplit, g, c.w.n_pad, st);
        prof_end(PROF_GEMM, st);
        if (e == hipSuccess) { g_counters[4]++; return 0; }
        if (e != hipErrorInvalidValue) return fail(-7, "conv5 launch: %s", hipGetErrorString(e));
    }
    return run_gemm_n(v->nsplit, M, g, c.w, EPI_GENERIC, true, c.w.n_pad % 128 ? 64 : 128, st);
}

// Activation1d over fp32 rows x [M][ch] -> the conv operand planes (out == nullptr) or fp32 rows out [M][ch]
static int bv_snake(f5hip_bigvgan* v, const float* x, int ch, int cpad, int M, int P, int T, const float* alpha, const float* beta, float* out,
                    hipStream_t st) {
    static const int old_kernel = getenv("F5HIP_BV_SNAKE") ? atoi(getenv("F5HIP_BV_SNAKE")) : 0;   // 1 = the round-1 LDS-tiled kernel (A/B; split-bf16 / fp32 outputs only)
    if (old_kernel == 1 && (out || v->nsplit == 2)) {
        hipLaunchKernelGGL(aa_snake_kernel, dim3((ch + 63) / 64, M / 64), dim3(256), 0, st, x, ch, ch, P, T, alpha, beta, v->filt, out ? (__bf16*)nullptr : v->act.hi,
                           out ? (__bf16*)nullptr : v->act.lo, out, out ? ch : cpad);
        CKL("aa_snake");
        return 0;
    }
    constexpr int R = 16;
    int cw = ch < 64 ? ch : 64;
    while (ch % cw) cw--;
    if (ch == 96) cw = 32;   // 8 segments of 32 channels fill the 256 lanes; 64 would leave a half-empty second column block
    const int nseg = 256 / cw;
    AaFilt f;
    memcpy(f.f, v->filt_h, sizeof(f.f));
    const dim3 grid(ch / cw, (T + nseg * R - 1) / (nseg * R), M / P);
    if (out) hipLaunchKernelGGL((aa_snake2_kernel<R, 0>), grid, dim3(256), 0, st, x, ch, ch, cw, nseg, P, T, alpha, beta, f, (__bf16*)nullptr, (__bf16*)nullptr, out, ch);
    else if (v->nsplit == 3) hipLaunchKernelGGL((aa_snake2_kernel<R, 2>), grid, dim3(256), 0, st, x, ch, ch, cw, nseg, P, T, alpha, beta, f, v->act.hi, (__bf16*)nullptr, (float*)nullptr, cpad);
    else hipLaunchKernelGGL((aa_snake2_kernel<R, 1>), grid, dim3(256), 0, st, x, ch, ch, cw, nseg, P, T, alpha, beta, f, v->act.hi, v->act.lo, (float*)nullptr, cpad);
    CKL("aa_snake2");
    return 0;
}

int f5hip_bigvgan_forward(f5hip_bigvgan* v, int32_t batch, int32_t frames, const float* mel_dev, float* wave_dev, void* stream) {
    if (!v || !v->finalized) return fail(-1, "vocoder not finalized");
    if (batch <= 0 || frames <= 0 || !mel_dev || !wave_dev) return fail(-1, "bigvgan_forward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const f5hip_bigvgan_config& c = v->cfg;
    const int T0 = frames, P0 = ceil_to(T0, 128);
    const int plane_mode = v->nsplit == 3 ? 2 : (v->nsplit == 2 ? 1 : 3);   // bv_mean3_kernel's output mode
    // largest stage: rows_i * C_i with rows_i = batch * P0 * prod(rates), C_i = c0 >> (i+1)
    size_t max_f32 = (size_t)batch * P0 * v->c0, max_act = (size_t)batch * P0 * ceil_to(v->c0, 32);
    {
        size_t rows = (size_t)batch * P0;
        for (int i = 0; i < v->n_up; i++) {
            rows *= c.upsample_rates[i];
            const int ch = v->c0 >> (i + 1);
            max_f32 = std::max(max_f32, rows * ch);
            max_act = std::max(max_act, rows * ceil_to(ch, 32));
        }
    }
    if (max_f32 > v->cap) {
        dev_free(v->ws);
        Arena a;
        for (int pass = 0; pass < 2; pass++) {
            a.reset(pass ? (char*)v->ws : nullptr);
            v->X = a.f32(max_f32); v->S = a.f32(max_f32); v->Tm = a.f32(max_f32);
            for (int j = 0; j < 3; j++) v->Y[j] = a.f32(max_f32);
            v->act = a.plane2(max_act + 4096); v->melp = a.plane2((size_t)batch * P0 * 128 + 4096);
            if (!pass) {
                if (hipMalloc(&v->ws, a.used()) != hipSuccess) { v->ws = nullptr; v->cap = 0; return fail(-5, "hipMalloc BigVGAN workspace %zu bytes", a.used()); }
                if (hipMemset(v->ws, 0, a.used()) != hipSuccess) return fail(-5, "hipMemset BigVGAN workspace");
            }
        }
        v->cap = max_f32;
    }
    prof_begin(PROF_VOCOS, st);
    // mel [B][num_mels][T] -> rows [B*P0][128] operand planes (uniform sequences: row = b * P0 + t)
    hipLaunchKernelGGL(bv_mel_rows_kernel, dim3(batch * P0), dim3(128), 0, st, mel_dev, c.num_mels, T0, P0, v->melp.hi, v->melp.lo, v->nsplit == 3 ? 1 : 0);
    CKL("bv_mel_rows");
    int M = batch * P0, P = P0, T = T0;
    int ch = v->c0;
    CK(bv_conv(v, v->pre, v->melp, M, P, T, ACT_NONE, nullptr, v->S, ch, st));   // S = conv_pre(mel)
    {   // operand planes of ups[0]
        const size_t n4 = (size_t)M * ch / 4;
        hipLaunchKernelGGL(bv_mean3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, v->S, v->S, v->S, 1, (size_t)M, ch, (float*)nullptr, v->act.hi, v->act.lo,
                           ceil_to(ch, 32), plane_mode);
        CKL("bv planes");
    }
    for (int i = 0; i < v->n_up; i++) {
        const int r = c.upsample_rates[i], co = ch / 2, cpad = ceil_to(co, 32);
        // ups[i]: 3-tap implicit GEMM over the planes of the previous stage -> X viewed as [M][r*co] == [M*r][co]
        CK(bv_conv(v, v->ups[i], v->act, M, P, T, ACT_NONE, nullptr, v->X, r * co, st));
        M *= r; P *= r; T *= r; ch = co;
        if (cpad != ch) {   // padded channels of the A operand must read as zero
            if (hipMemsetAsync(v->act.hi, 0, (size_t)M * cpad * 2, st) != hipSuccess || (v->nsplit == 2 && hipMemsetAsync(v->act.lo, 0, (size_t)M * cpad * 2, st) != hipSuccess))
                return fail(-6, "bigvgan memset");
        }
        for (int j = 0; j < 3; j++) {
            const BvRes& rb = v->res[i * 3 + j];
            float* y = v->Y[j];
            for (int d = 0; d < 3; d++) {
                const float* in = d == 0 ? v->X : y;   // AMPBlock1: x = x + convs2[d](act(convs1[d](act(x))))
                CK(bv_snake(v, in, ch, cpad, M, P, T, rb.alpha[2 * d], rb.beta[2 * d], nullptr, st));
                CK(bv_conv(v, rb.c1[d], v->act, M, P, T, ACT_NONE, nullptr, v->Tm, ch, st));
                CK(bv_snake(v, v->Tm, ch, cpad, M, P, T, rb.alpha[2 * d + 1], rb.beta[2 * d + 1], nullptr, st));
                CK(bv_conv(v, rb.c2[d], v->act, M, P, T, ACT_NONE, in, y, ch, st));
            }
        }
        // mean of the three blocks: the last stage keeps fp32 rows for activation_post, the others only feed the next up-sampler
        const bool last = i == v->n_up - 1;
        const size_t n4 = (size_t)M * ch / 4;
        hipLaunchKernelGGL(bv_mean3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, v->Y[0], v->Y[1], v->Y[2], 3, (size_t)M, ch, last ? v->S : (float*)nullptr,
                           v->act.hi, v->act.lo, cpad, last ? 0 : plane_mode);
        CKL("bv_mean3");
    }
    // activation_post -> fp32 (Tm), conv_post + clamp -> wave [B][T]
    CK(bv_snake(v, v->S, ch, ch, M, P, T, v->post_alpha, v->post_beta, v->Tm, st));
    const size_t post_lds = (size_t)(262 * (ch + 1) + 7 * ch) * sizeof(float);
    if (post_lds <= 48 * 1024) hipLaunchKernelGGL(bv_conv_post_kernel, dim3((T + 255) / 256, batch), dim3(256), post_lds, st, v->Tm, ch, ch, P, T, v->post_w, wave_dev);
    else hipLaunchKernelGGL(bv_conv_post_naive_kernel, dim3((T + 255) / 256, batch), dim3(256), 0, st, v->Tm, ch, ch, P, T, v->post_w, wave_dev);
    CKL("conv_post");
    prof_end(PROF_VOCOS, st);
    return 0;
}
