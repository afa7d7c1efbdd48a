// Per-kernel unit operators of the C ABI (include/f5hip.h, "unit ops"): each runs ONE production kernel through the same dispatcher the
// sampler uses, on fp32 device tensors, so that `pytest -m gpu` can pin every hot kernel against a plain fp32 reference and the
// tools can time it in isolation.  Included at the end of f5hip.hip (same translation unit as the dispatcher).
#pragma once

struct OpBufs {
    std::vector<void*> ptrs;
    ~OpBufs() { for (void* p : ptrs) if (p) (void)hipFree(p); }
    template <typename T> T* get(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T*)p;
    }
};

// device fp32 [R][C] -> operand planes [R_pad][C] (prec 3: one fp16 plane; 2: split bf16; 1: bf16 hi only is read)
static void op_pack_planes(const float* src, int R, int C, int R_pad, __bf16* hi, __bf16* lo, bool f16, hipStream_t st) {
    hipLaunchKernelGGL(pack_weight_kernel, dim3(R_pad), dim3(256), 0, st, src, R, C, C, hi, f16 ? (__bf16*)nullptr : lo, C);
}

// diagnostics (-DF5HIP_GEMM5_ABL builds, F5HIP_GEMM5_ABL=5): s_memrealtime stamps (100 MHz) of wave 0 (consumer) and wave 4 (loader) of every workgroup
template <typename Launch>
static int gemm5_stamp_report(OpBufs& b, int M, int N, int K, hipStream_t st, Launch launch) {
    const int maxg = 4096;
    unsigned long long* d = b.get<unsigned long long>((size_t)maxg * 16);
    if (!d) return 0;
    (void)hipMemsetAsync(d, 0, sizeof(unsigned long long) * maxg * 16, st);
    for (int rep = 0; rep < 3; rep++) CK(launch(d, rep));
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)maxg * 16);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    int ng = 0;
    for (int g2 = 0; g2 < maxg; g2++) if (h[(size_t)g2 * 16]) { ng = g2 + 1; tmin = std::min(tmin, h[(size_t)g2 * 16]); tmax = std::max(tmax, std::max(h[(size_t)g2 * 16 + 6], h[(size_t)g2 * 16 + 14])); }
    const char* names[7] = {"start", "tile0 landed (loader)", "k-loop end", "E1 passed", "E2 passed (slab done)", "row phase issued", "stores drained"};
    fprintf(stderr, "[gemm5 stamps] M %d N %d K %d: %d workgroups, first start -> last end %.2f us\n", M, N, K, ng, (tmax - tmin) * 0.01);
    for (int w = 0; w < 2; w++)
        for (int i = 0; i < 7; i++) {
            std::vector<double> v;
            for (int g2 = 0; g2 < ng; g2++) { const unsigned long long t = h[(size_t)g2 * 16 + w * 8 + i]; if (t) v.push_back((t - tmin) * 0.01); }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[gemm5 stamps]   wave %d  %-26s min %6.2f  median %6.2f  max %6.2f us after the first workgroup started\n", w * 4, names[i], v.front(), v[v.size() / 2], v.back());
        }
    // epilogue duration (E1 -> end of the row phase) by column slab of the tile (XCD-blocked order of gemm5_tile_of_block, 16 column slabs)
    if (ng == 256) {
        for (int tn = 0; tn < 16; tn++) {
            std::vector<double> v;
            for (int g2 = 0; g2 < ng; g2++) {
                const int tile = (g2 & 7) * (ng >> 3) + (g2 >> 3);
                if (tile % 16 == tn) v.push_back((h[(size_t)g2 * 16 + 5] - h[(size_t)g2 * 16 + 3]) * 0.01);
            }
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[gemm5 stamps]   column slab %2d: epilogue min %5.2f median %5.2f max %5.2f us\n", tn, v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}

// diagnostics of gemm6 (F5HIP_GEMM6_STAMPS=1, tools/gemm6_stamps.py): per-workgroup time line from the kernel's run-time stamps
template <typename Launch>
static int gemm6_stamp_report(OpBufs& b, int M, int N, int K, hipStream_t st, Launch launch) {
    const int maxg = 4096;
    unsigned long long* d = b.get<unsigned long long>((size_t)maxg * 16);
    if (!d) return 0;
    (void)hipMemsetAsync(d, 0, sizeof(unsigned long long) * maxg * 16, st);
    for (int rep = 0; rep < 3; rep++) CK(launch(d, rep));
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)maxg * 16);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    int ng = 0;
    for (int g2 = 0; g2 < maxg; g2++) if (h[(size_t)g2 * 16]) { ng = g2 + 1; tmin = std::min(tmin, h[(size_t)g2 * 16]); tmax = std::max(tmax, std::max(h[(size_t)g2 * 16 + 6], h[(size_t)g2 * 16 + 14])); }
    fprintf(stderr, "[gemm6 stamps] M %d N %d K %d: %d workgroups, first start -> last end %.2f us\n", M, N, K, ng, (tmax - tmin) * 0.01);
    const char* names[6] = {"k-loop", "quarter 0", "quarter 1", "quarter 2", "quarter 3", "store drain"};
    for (int w = 0; w < 2; w++)
        for (int i = 0; i < 6; i++) {
            std::vector<double> v;
            for (int g2 = 0; g2 < ng; g2++) {
                const unsigned long long t0 = h[(size_t)g2 * 16 + w * 8 + i], t1 = h[(size_t)g2 * 16 + w * 8 + i + 1];
                if (t0 && t1) v.push_back((t1 - t0) * 0.01);
            }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[gemm6 stamps]   wave %d  %-12s duration min %6.2f  median %6.2f  max %6.2f us\n", w * 4, names[i], v.front(), v[v.size() / 2], v.back());
        }
    {   // the shader clock the k-loops ran at: s_memtime cycles over s_memrealtime (100 MHz) ticks
        std::vector<double> v;
        for (int g2 = 0; g2 < ng; g2++) {
            const unsigned long long cyc = h[(size_t)g2 * 16 + 7], t0 = h[(size_t)g2 * 16], t1 = h[(size_t)g2 * 16 + 1];
            if (cyc && t1 > t0) v.push_back((double)cyc / ((double)(t1 - t0) * 10.0));   // cycles per ns = GHz
        }
        if (!v.empty()) {
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[gemm6 stamps]   shader clock during the k-loop: min %.2f  median %.2f  max %.2f GHz  (the 2.5 PFLOP/s roof is 2.4 GHz x 4096 FLOP / clk / CU x 256 CUs)\n", v.front(), v[v.size() / 2], v.back());
        }
    }
    {   // start times: how the rounds lay out
        std::vector<double> v;
        for (int g2 = 0; g2 < ng; g2++) v.push_back((h[(size_t)g2 * 16] - tmin) * 0.01);
        std::sort(v.begin(), v.end());
        fprintf(stderr, "[gemm6 stamps]   workgroup start: 25 %% %.2f  50 %% %.2f  75 %% %.2f  90 %% %.2f  last %.2f us\n", v[ng / 4], v[ng / 2], v[3 * ng / 4], v[9 * ng / 10], v.back());
    }
    return 0;
}

extern "C" int f5hip_op_gemm(int32_t M, int32_t N, int32_t K, const float* a_dev, const float* w_dev, const float* bias_dev, int32_t prec,
                             int32_t act, const float* mul_dev, const float* res_dev, const uint8_t* row_keep_host, float* out_dev,
                             uint16_t* out16_dev, int32_t w_copies, int32_t iters, double* avg_us, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4 || !a_dev || !w_dev || prec < 1 || prec > 3 || (!out_dev && !out16_dev))
        return fail(-1, "op_gemm: bad argument (need K %% 32 == 0, N %% 4 == 0, prec 1..3)");
    if (out16_dev && (res_dev || out_dev)) return fail(-1, "op_gemm: the 16-bit output is the (no residual, no fp32 output) epilogue");
    hipStream_t st = (hipStream_t)stream;
    const int M_pad = (M + 127) / 128 * 128, N_pad = (N + 127) / 128 * 128;
    const bool f16 = prec == 3;
    if (w_copies < 1) w_copies = 1;
    OpBufs b;
    Plane2 A;
    A.hi = b.get<__bf16>((size_t)M_pad * K); A.lo = b.get<__bf16>((size_t)M_pad * K);
    float* bias = b.get<float>(N_pad);
    int* keep = nullptr;
    if (!A.hi || !A.lo || !bias) return fail(-5, "op_gemm: hipMalloc");
    std::vector<PackedW> Ws(w_copies);
    for (auto& W : Ws) {
        W.hi = b.get<__bf16>((size_t)N_pad * K); W.lo = f16 ? nullptr : b.get<__bf16>((size_t)N_pad * K);
        if (!W.hi || (!f16 && !W.lo)) return fail(-5, "op_gemm: hipMalloc weights");
        W.n = N; W.k = K; W.n_pad = N_pad; W.k_pad = K; W.ld = K; W.bias = bias; W.f16 = f16;
        op_pack_planes(w_dev, N, K, N_pad, W.hi, W.lo, f16, st);
        if (f16 && N >= 2048 && pack_frag(W, st)) return -5;   // like the model's FF1 weights: fragment order for the W-direct kernels (freed with b)
        if (W.frag) b.ptrs.push_back(W.frag);
    }
    op_pack_planes(a_dev, M, K, M_pad, A.hi, A.lo, f16, st);
    (void)hipMemsetAsync(bias, 0, sizeof(float) * N_pad, st);
    if (bias_dev) (void)hipMemcpyAsync(bias, bias_dev, sizeof(float) * N, hipMemcpyDeviceToDevice, st);
    if (row_keep_host) {
        keep = b.get<int>(M_pad);
        std::vector<int> hk(M_pad, 0);
        for (int i = 0; i < M; i++) hk[i] = row_keep_host[i];
        if (!keep || hipMemcpyAsync(keep, hk.data(), sizeof(int) * M_pad, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(-6, "op_gemm: row_keep upload");
    }
    auto args_for = [&](const PackedW& W) {
        GemmArgs g = gemm_base(A, K, W, M);
        g.act = act; g.mul = mul_dev; g.res = res_dev; g.ldres = N; g.row_keep = keep;
        if (out16_dev) { g.out_hi = (__bf16*)out16_dev; g.ldob = N; g.f16_out = 1; }
        else { g.out_f32 = out_dev; g.ldo = N; }
        return g;
    };
    {
        GemmArgs g = args_for(Ws[0]);
        CK(run_gemm_n(prec, M_pad, g, Ws[0], EPI_GENERIC, false, 128, st));
    }
    if (getenv("F5HIP_GEMM5_STAMPS"))
        CK(gemm5_stamp_report(b, M, N, K, st, [&](unsigned long long* d, int rep) {
            GemmArgs g = args_for(Ws[rep % w_copies]);
            g.stamps = d;
            return run_gemm_n(prec, M_pad, g, Ws[rep % w_copies], EPI_GENERIC, false, 128, st);
        }));
    if (getenv("F5HIP_GEMM6_STAMPS"))
        CK(gemm6_stamp_report(b, M, N, K, st, [&](unsigned long long* d, int rep) {
            GemmArgs g = args_for(Ws[rep % w_copies]);
            g.stamps = d;
            return run_gemm_n(prec, M_pad, g, Ws[rep % w_copies], EPI_GENERIC, false, 128, st);
        }));
    if (iters > 0 && avg_us) {
        // timing: the residual epilogue accumulates in place, so time into a scratch output
        float* scratch = b.get<float>((size_t)M * N);
        if (!scratch) return fail(-5, "op_gemm: hipMalloc scratch");
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int it = -3; it < iters; it++) {
            if (it == 0) (void)hipEventRecord(e0, st);
            const PackedW& W = Ws[(it + 3) % w_copies];
            GemmArgs g = args_for(W);
            if (!out16_dev) { g.out_f32 = scratch; if (res_dev) g.res = scratch; }
            CK(run_gemm_n(prec, M_pad, g, W, EPI_GENERIC, false, 128, st));
        }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_us = (double)ms * 1e3 / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(-7, "op_gemm: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

// Fused QKV projection with its epilogue (bias, rotary on head 0, q / 8, V transposed): F/model/modules.py:409-426.
//   a_dev fp32 [M][D], w_dev fp32 [3 D][D] (to_q | to_k | to_v rows), bias_dev [3 D], row_pos host int32 [M]
//   qk_dev  bf16 [M_pad][2 D] (q pre-scaled by 1/8 | k), vt_dev bf16 [D][M_pad], M_pad = ceil128(M)
extern "C" int f5hip_op_qkv(int32_t M, int32_t D, const float* a_dev, const float* w_dev, const float* bias_dev, const int32_t* row_pos,
                            int32_t prec, uint16_t* qk_dev, uint16_t* vt_dev, int32_t iters, double* avg_us, void* stream) {
    if (M <= 0 || D <= 0 || D % 128 || !a_dev || !w_dev || !bias_dev || !row_pos || !qk_dev || !vt_dev || prec < 1 || prec > 3)
        return fail(-1, "op_qkv: bad argument (need D %% 128 == 0)");
    hipStream_t st = (hipStream_t)stream;
    const int M_pad = (M + 127) / 128 * 128, N = 3 * D, N_pad = (N + 127) / 128 * 128;
    const bool f16 = prec == 3;
    OpBufs b;
    Plane2 A; PackedW W;
    A.hi = b.get<__bf16>((size_t)M_pad * D); A.lo = b.get<__bf16>((size_t)M_pad * D);
    W.hi = b.get<__bf16>((size_t)N_pad * D); W.lo = f16 ? nullptr : b.get<__bf16>((size_t)N_pad * D);
    float* bias = b.get<float>(N_pad);
    float* rc = b.get<float>((size_t)4097 * 32); float* rs = b.get<float>((size_t)4097 * 32);
    int* pos = b.get<int>(M_pad);
    if (!A.hi || !A.lo || !W.hi || (!f16 && !W.lo) || !bias || !rc || !rs || !pos) return fail(-5, "op_qkv: hipMalloc");
    W.n = N; W.k = D; W.n_pad = N_pad; W.k_pad = D; W.ld = D; W.bias = bias; W.f16 = f16;
    std::vector<float> hc((size_t)4097 * 32), hs((size_t)4097 * 32);
    for (int p = 0; p < 4097; p++)
        for (int i = 0; i < 32; i++) {
            const float ang = (float)p * (1.0f / powf(10000.0f, (float)(2 * i) / 64.0f));
            hc[(size_t)p * 32 + i] = (float)cos((double)ang); hs[(size_t)p * 32 + i] = (float)sin((double)ang);
        }
    std::vector<int> hp(M_pad, 0);
    for (int i = 0; i < M; i++) { if (row_pos[i] < 0 || row_pos[i] > 4096) return fail(-1, "op_qkv: row_pos out of range"); hp[i] = row_pos[i]; }
    if (hipMemcpyAsync(rc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(rs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(-6, "op_qkv: table upload");
    op_pack_planes(w_dev, N, D, N_pad, W.hi, W.lo, f16, st);
    if (f16 && pack_frag(W, st)) return -5;
    if (W.frag) b.ptrs.push_back(W.frag);
    op_pack_planes(a_dev, M, D, M_pad, A.hi, A.lo, f16, st);
    (void)hipMemsetAsync(bias, 0, sizeof(float) * N_pad, st);
    (void)hipMemcpyAsync(bias, bias_dev, sizeof(float) * N, hipMemcpyDeviceToDevice, st);
    GemmArgs g = gemm_base(A, D, W, M);
    g.D = D; g.row_pos = pos; g.rope_cos = rc; g.rope_sin = rs; g.qk = (__bf16*)qk_dev; g.vt = (__bf16*)vt_dev; g.ldvt = M_pad;
    if (getenv("F5HIP_GEMM5_STAMPS"))
        CK(gemm5_stamp_report(b, M, N, D, st, [&](unsigned long long* d, int) {
            GemmArgs g2 = g;
            g2.stamps = d;
            return run_gemm_n(prec, M_pad, g2, W, EPI_QKV, false, 128, st);
        }));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = -1; it < iters; it++) {
        if (it == 0) (void)hipEventRecord(e0, st);
        CK(run_gemm_n(prec, M_pad, g, W, EPI_QKV, false, 128, st));
    }
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    if (iters > 0 && avg_us) { float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1); *avg_us = (double)ms * 1e3 / iters; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(-7, "op_qkv: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

// LayerNorm + modulation y = LN(x) (gain_off + scale) + shift (F/model/modules.py:285-290), or x-transformers RMSNorm (rms = 1): fp32 in, fp32 out
extern "C" int f5hip_op_layernorm(int32_t M, int32_t D, const float* x_dev, const float* scale_dev, const float* shift_dev, float gain_off, float eps,
                                  int32_t rms, float* out_dev, void* stream) {
    if (M <= 0 || D <= 0 || D % 4 || !x_dev || !scale_dev || !shift_dev || !out_dev) return fail(-1, "op_layernorm: bad argument");
    hipStream_t st = (hipStream_t)stream;
    LnArgs ln; memset(&ln, 0, sizeof(ln));
    ln.x = x_dev; ln.ldx = D; ln.M = M; ln.D = D; ln.scale = scale_dev; ln.shift = shift_dev; ln.gain_off = gain_off; ln.eps = eps; ln.rms = rms;
    ln.out_f32 = out_dev; ln.ldof = D;
    CK(run_ln(ln, st));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- attention unit op
__global__ __launch_bounds__(256) void op_pack_qkv_kernel(const float* q, const float* k, const float* v, int D, const int* row_src, int M_pad,
                                                          __bf16* qk, __bf16* vt) {
    const int row = blockIdx.x, src = row_src[row];
    for (int c = threadIdx.x; c < D; c += 256) {
        const float qv = src >= 0 ? q[(size_t)src * D + c] * F5_Q_SCALE : 0.0f;   // q is pre-scaled by log2(e) / 8 like the QKV epilogue's
        const float kv = src >= 0 ? k[(size_t)src * D + c] : 0.0f;
        const float vv = src >= 0 ? v[(size_t)src * D + c] : 0.0f;
        // fp16 bits, saturated, like the QKV epilogue's outputs
        reinterpret_cast<_Float16*>(qk)[(size_t)row * 2 * D + c] = sat_f16(qv);
        reinterpret_cast<_Float16*>(qk)[(size_t)row * 2 * D + D + c] = sat_f16(kv);
        reinterpret_cast<_Float16*>(vt)[(size_t)c * M_pad + vt_col(row)] = sat_f16(vv);
    }
}
__global__ __launch_bounds__(256) void op_unpack_planes_kernel(const __bf16* hi, const __bf16* lo, int D, const int* frame_row, float* out) {
    const int f = blockIdx.x, row = frame_row[f];
    for (int c = threadIdx.x; c < D; c += 256) out[(size_t)f * D + c] = (float)hi[(size_t)row * D + c] + (float)lo[(size_t)row * D + c];
}

// softmax(q k^T / 8 + key-padding mask) v per (sequence, head), head dim 64 (F/model/modules.py:424-436): q / k / v fp32 [sum(seq_len)][64 heads]
// are rounded to fp16 like the QKV epilogue's outputs (q after the log2(e) / 8 scale); out fp32 [sum(seq_len)][64 heads] = split-bf16 planes summed.
// impl 3 = attn3 (production), 4 = experiments/attn4.h (attn3 unless built with -DF5HIP_EXPERIMENTS).
extern "C" int f5hip_op_attention(int32_t n_seq, const int32_t* seq_len, const int32_t* kv_len, int32_t heads, const float* q_dev,
                                  const float* k_dev, const float* v_dev, float* out_dev, int32_t impl, int32_t iters, double* avg_us, void* stream) {
    if (n_seq <= 0 || !seq_len || heads <= 0 || !q_dev || !k_dev || !v_dev || !out_dev || (impl != 3 && impl != 4 && impl != 5)) return fail(-1, "op_attention: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int D = heads * 64;
    int M_pad = 0, F = 0, max_len = 0;
    for (int i = 0; i < n_seq; i++) {
        if (seq_len[i] <= 0 || seq_len[i] > 4096 || (kv_len && (kv_len[i] <= 0 || kv_len[i] > seq_len[i]))) return fail(-1, "op_attention: bad lengths");
        M_pad += (seq_len[i] + 127) / 128 * 128; F += seq_len[i]; max_len = std::max(max_len, (int)seq_len[i]);
    }
    std::vector<int> row_src(M_pad, -1), frame_row(F), meta(3 * n_seq);
    for (int i = 0, r0 = 0, f0 = 0; i < n_seq; i++) {
        meta[i] = r0; meta[n_seq + i] = seq_len[i]; meta[2 * n_seq + i] = kv_len ? kv_len[i] : seq_len[i];
        for (int j = 0; j < seq_len[i]; j++) { row_src[r0 + j] = f0 + j; frame_row[f0 + j] = r0 + j; }
        r0 += (seq_len[i] + 127) / 128 * 128; f0 += seq_len[i];
    }
    OpBufs b;
    __bf16* qk = b.get<__bf16>((size_t)(M_pad + 256) * 2 * D); __bf16* vt = b.get<__bf16>((size_t)D * M_pad);
    __bf16* ohi = b.get<__bf16>((size_t)M_pad * D); __bf16* olo = b.get<__bf16>((size_t)M_pad * D);
    int* d_rs = b.get<int>(M_pad); int* d_fr = b.get<int>(F); int* d_meta = b.get<int>(3 * n_seq);
    if (!qk || !vt || !ohi || !olo || !d_rs || !d_fr || !d_meta) return fail(-5, "op_attention: hipMalloc");
    if (hipMemcpyAsync(d_rs, row_src.data(), sizeof(int) * M_pad, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_fr, frame_row.data(), sizeof(int) * F, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_meta, meta.data(), sizeof(int) * 3 * n_seq, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(qk, 0, sizeof(__bf16) * (size_t)(M_pad + 256) * 2 * D, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(-6, "op_attention: upload");
    hipLaunchKernelGGL(op_pack_qkv_kernel, dim3(M_pad), dim3(256), 0, st, q_dev, k_dev, v_dev, D, d_rs, M_pad, qk, vt);
    AttnArgs at; memset(&at, 0, sizeof(at));
    at.qk = qk; at.vt = vt; at.D = D; at.ldvt = M_pad; at.seq_row0 = d_meta; at.seq_len = d_meta + n_seq; at.seq_kvlen = d_meta + 2 * n_seq;
    at.out_hi = ohi; at.out_lo = olo; at.shape_invariant = -1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipError_t e = hipSuccess;
    for (int it = -1; it < iters && e == hipSuccess; it++) {
        if (it == 0) (void)hipEventRecord(e0, st);
        e = impl == 3 ? f5_launch_attn3(at, max_len, heads, n_seq, st) : impl == 5 ? f5_launch_attn5(at, max_len, heads, n_seq, st) : f5_launch_attn4(at, max_len, heads, n_seq, st);
    }
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    if (iters > 0 && avg_us) { float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1); *avg_us = (double)ms * 1e3 / iters; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(-7, "op_attention launch: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(op_unpack_planes_kernel, dim3(F), dim3(256), 0, st, ohi, olo, D, d_fr, out_dev);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(-7, "op_attention: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}


// Joint attention of the MMDiT blocks (JointAttnProcessor, F/model/modules.py:496-522): per sequence the queries and the keys are the
// audio rows followed by the text rows; padding is masked on the audio keys only (x_kvlen <= x_len valid audio keys, every text key).
// q / k / v / out fp32 [sum(x_len) + sum(c_len)][64 heads]: all audio frames sequence by sequence, then all text tokens sequence by
// sequence.  Runs the two-range attn3 kernels over 2 n_seq pseudo-sequences (audio queries, text queries) that share the key ranges.
extern "C" int f5hip_op_joint_attention(int32_t n_seq, const int32_t* x_len, const int32_t* x_kvlen, const int32_t* c_len, int32_t heads,
                                        const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, void* stream) {
    if (n_seq <= 0 || !x_len || !c_len || heads <= 0 || !q_dev || !k_dev || !v_dev || !out_dev) return fail(-1, "op_joint_attention: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int D = heads * 64;
    int Mx = 0, Mc = 0, Fx = 0, Fc = 0, max_len = 0;
    for (int i = 0; i < n_seq; i++) {
        if (x_len[i] <= 0 || x_len[i] > 4096 || c_len[i] <= 0 || c_len[i] > 4096 || (x_kvlen && (x_kvlen[i] <= 0 || x_kvlen[i] > x_len[i])))
            return fail(-1, "op_joint_attention: bad lengths");
        Mx += (x_len[i] + 127) / 128 * 128; Mc += (c_len[i] + 127) / 128 * 128; Fx += x_len[i]; Fc += c_len[i];
        max_len = std::max(max_len, std::max((int)x_len[i], (int)c_len[i]));
    }
    const int M_pad = Mx + Mc, F = Fx + Fc, NS = 2 * n_seq;
    std::vector<int> row_src(M_pad, -1), frame_row(F), meta(6 * NS);
    for (int i = 0, rx = 0, rc = Mx, fx = 0, fc = Fx; i < n_seq; i++) {
        for (int j = 0; j < x_len[i]; j++) { row_src[rx + j] = fx + j; frame_row[fx + j] = rx + j; }
        for (int j = 0; j < c_len[i]; j++) { row_src[rc + j] = fc + j; frame_row[fc + j] = rc + j; }
        for (int q = 0; q < 2; q++) {              // pseudo-sequence 2 i: audio queries, 2 i + 1: text queries
            const int s = 2 * i + q;
            meta[s] = q ? rc : rx;                                  // seq_row0
            meta[NS + s] = q ? c_len[i] : x_len[i];                 // seq_len
            meta[2 * NS + s] = x_kvlen ? x_kvlen[i] : x_len[i];     // seq_kvlen (first key range: audio)
            meta[3 * NS + s] = rx;                                  // seq_kv_row0
            meta[4 * NS + s] = rc;                                  // seq_kv2_row0
            meta[5 * NS + s] = c_len[i];                            // seq_kv2_len
        }
        rx += (x_len[i] + 127) / 128 * 128; rc += (c_len[i] + 127) / 128 * 128; fx += x_len[i]; fc += c_len[i];
    }
    OpBufs b;
    __bf16* qk = b.get<__bf16>((size_t)(M_pad + 256) * 2 * D); __bf16* vt = b.get<__bf16>((size_t)D * (M_pad + 256));
    __bf16* ohi = b.get<__bf16>((size_t)(M_pad + 256) * D); __bf16* olo = b.get<__bf16>((size_t)(M_pad + 256) * D);
    int* d_rs = b.get<int>(M_pad); int* d_fr = b.get<int>(F); int* d_meta = b.get<int>(6 * NS);
    if (!qk || !vt || !ohi || !olo || !d_rs || !d_fr || !d_meta) return fail(-5, "op_joint_attention: hipMalloc");
    if (hipMemcpyAsync(d_rs, row_src.data(), sizeof(int) * M_pad, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_fr, frame_row.data(), sizeof(int) * F, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_meta, meta.data(), sizeof(int) * 6 * NS, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(qk, 0, sizeof(__bf16) * (size_t)(M_pad + 256) * 2 * D, st) != hipSuccess ||
        hipMemsetAsync(vt, 0, sizeof(__bf16) * (size_t)D * (M_pad + 256), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(-6, "op_joint_attention: upload");
    hipLaunchKernelGGL(op_pack_qkv_kernel, dim3(M_pad), dim3(256), 0, st, q_dev, k_dev, v_dev, D, d_rs, M_pad + 256, qk, vt);
    AttnArgs at; memset(&at, 0, sizeof(at));
    at.qk = qk; at.vt = vt; at.D = D; at.ldvt = M_pad + 256; at.seq_row0 = d_meta; at.seq_len = d_meta + NS; at.seq_kvlen = d_meta + 2 * NS;
    at.seq_kv_row0 = d_meta + 3 * NS; at.seq_kv2_row0 = d_meta + 4 * NS; at.seq_kv2_len = d_meta + 5 * NS;
    at.out_hi = ohi; at.out_lo = olo; at.shape_invariant = -1;
    const hipError_t e = f5_launch_attn3(at, max_len, heads, NS, st);
    if (e != hipSuccess) return fail(-7, "op_joint_attention launch: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(op_unpack_planes_kernel, dim3(F), dim3(256), 0, st, ohi, olo, D, d_fr, out_dev);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(-7, "op_joint_attention: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

// One Conv1d of the BigVGAN kind over channel-last rows -- batch sequences of pitch P rows, T valid -- through the library's two paths:
// impl 0 = gemm.h implicit GEMM (A window re-read per tap), 5 = conv5.h (window once in LDS).  x_dev fp32 [batch * P][c_in],
// w_host [c_out][c_in][k], out_dev fp32 [batch * P][c_out] (= conv + bias + res).  prec 2 = split bf16, 3 = fp16.
// stamps_host (optional, impl 5, wide shapes): [blocks][16] cycle stamps of the diagnostics kernel (conv5.h).
extern "C" int f5hip_op_conv1d(int32_t batch, int32_t P, int32_t T, int32_t c_in, int32_t c_out, int32_t k, int32_t dil, const float* x_dev,
                               const float* w_host, const float* bias_host, const float* res_dev, float* out_dev, int32_t prec, int32_t impl,
                               int32_t iters, double* avg_us, uint64_t* stamps_host, int32_t stamp_blocks, void* stream) {
    if (batch <= 0 || P <= 0 || T <= 0 || T > P || P % 128 || c_in <= 0 || c_in % 4 || c_out <= 0 || k < 1 || !(k & 1) || dil < 1 || !x_dev || !w_host || !out_dev ||
        (prec != 2 && prec != 3) || (impl != 0 && impl != 5))
        return fail(-1, "op_conv1d: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int M = batch * P, cpad = ceil_to(c_in, 32);
    BvConv c;
    const std::vector<float> w(w_host, w_host + (size_t)c_out * c_in * k);
    std::vector<float> bias(c_out, 0.0f);
    if (bias_host) bias.assign(bias_host, bias_host + c_out);
    if (bv_pack_conv(c, w, bias.data(), c_out, c_in, k, dil, prec == 3)) return -4;
    OpBufs b;
    Plane2 A;
    A.hi = b.get<__bf16>((size_t)M * cpad + 4096); A.lo = b.get<__bf16>((size_t)M * cpad + 4096);
    unsigned long long* stamps = stamps_host ? b.get<unsigned long long>((size_t)stamp_blocks * 16) : nullptr;
    if (!A.hi || !A.lo || (stamps_host && !stamps)) { bv_free_conv(c); return fail(-5, "op_conv1d: hipMalloc"); }
    (void)hipMemsetAsync(A.hi, 0, ((size_t)M * cpad + 4096) * 2, st);
    (void)hipMemsetAsync(A.lo, 0, ((size_t)M * cpad + 4096) * 2, st);
    const size_t n4 = (size_t)M * c_in / 4;
    hipLaunchKernelGGL(bv_mean3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x_dev, x_dev, x_dev, 1, (size_t)M, c_in, (float*)nullptr, A.hi, A.lo, cpad,
                       prec == 3 ? 2 : 1);
    auto run = [&](float* out, const float* res, unsigned long long* stm) -> int {
        GemmArgs g = gemm_base(A, c.c_in_pad, c.w, M);
        g.conv_kpt = c.c_in_pad / 32; g.conv_center = (k - 1) / 2; g.conv_dil = dil; g.seq_pitch = P; g.seq_valid = T;
        g.res = res; g.ldres = c_out; g.out_f32 = out; g.ldo = c_out; g.stamps = stm;
        if (impl == 5) {
            const hipError_t e = f5_launch_conv5(prec, g, c.w.n_pad, st);
            if (e != hipSuccess) return fail(-7, "op_conv1d: conv5 %s", e == hipErrorInvalidValue ? "does not cover this shape" : hipGetErrorString(e));
            return 0;
        }
        return run_gemm_n(prec, M, g, c.w, EPI_GENERIC, true, c.w.n_pad % 128 ? 64 : 128, st);
    };
    int rc = run(out_dev, res_dev, nullptr);
    if (!rc && iters > 0 && avg_us) {
        float* scratch = b.get<float>((size_t)M * c_out);
        if (!scratch) rc = fail(-5, "op_conv1d: hipMalloc scratch");
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int it = -3; it < iters && !rc; it++) {
            if (it == 0) (void)hipEventRecord(e0, st);
            rc = run(scratch, nullptr, nullptr);
        }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_us = 1e3 * ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (!rc && stamps) {
            (void)hipMemsetAsync(stamps, 0, (size_t)stamp_blocks * 16 * 8, st);
            rc = run(scratch, nullptr, stamps);
            if (!rc && (hipMemcpyAsync(stamps_host, stamps, (size_t)stamp_blocks * 16 * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
                rc = fail(-6, "op_conv1d: stamp download");
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(-6, "op_conv1d: sync");
    bv_free_conv(c);
    return rc;
}
