// Diagnostics: kernel micro-benchmarks timed with HIP events (not part of the product path).
#pragma once

__global__ void fill_pattern_kernel(float* x, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    x[i] = ((float)(h & 0xFFFF) / 32768.0f - 1.0f);
}

// per-phase s_memtime stamps (100 MHz ticks) of one workgroup of the register-staged kernel: out[0..5] = prologue, load issue,
// LDS read + MFMA, load wait + LDS write, barrier, epilogue; out[6] = total
extern "C" int f5hip_debug_gemm_stamps(int32_t M, int32_t N, int32_t K, int32_t bn, int32_t bx, int32_t by, unsigned long long* out) {
    float *fa = nullptr, *fw = nullptr; Plane2 A, O; PackedW W;
    size_t na = (size_t)M * K, nw = (size_t)N * K, no = (size_t)M * N;
    unsigned long long* dst = nullptr;
    if (hipMalloc(&fa, na * 4) || hipMalloc(&fw, nw * 4) || hipMalloc(&A.hi, na * 2) || hipMalloc(&A.lo, na * 2) || hipMalloc(&O.hi, no * 2) ||
        hipMalloc(&O.lo, no * 2) || hipMalloc(&W.hi, nw * 2) || hipMalloc(&W.lo, nw * 2) || hipMalloc(&dst, 128)) return fail(-5, "stamps: hipMalloc");
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((na + 255) / 256), dim3(256), 0, 0, fa, na, 1u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((nw + 255) / 256), dim3(256), 0, 0, fw, nw, 2u);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(M), dim3(256), 0, 0, fa, M, K, K, A.hi, A.lo, K);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(N), dim3(256), 0, 0, fw, N, K, K, W.hi, W.lo, K);
    W.n = N; W.k = K; W.n_pad = N; W.k_pad = K; W.ld = K; W.bias = nullptr;
    GemmArgs a = gemm_base(A, K, W, M);
    a.act = ACT_GELU_TANH; a.out_hi = O.hi; a.out_lo = O.lo; a.ldob = N; a.stamps = dst;
    if (bx < 0) { bx = -bx - 1; a.act = ACT_NONE; a.out_hi = nullptr; a.out_lo = nullptr; hipMalloc(&a.out_f32, no * 4); a.ldo = N; a.res = a.out_f32; a.ldres = N; } a.stamp_bx = bx; a.stamp_by = by;
    hipError_t e = hipSuccess;
    for (int it = 0; it < 3; it++) e = bn == 3 ? launch_gemm3_t<2, EPI_GENERIC, 3>(a, M, N, 0) : bn == 128 ? launch_gemm_t<2, 128, false, EPI_GENERIC, 3>(a, M, N, 0) : launch_gemm_t<2, 64, false, EPI_GENERIC, 3>(a, M, N, 0);
    hipDeviceSynchronize();
    hipMemcpy(out, dst, 88, hipMemcpyDeviceToHost);
    for (void* p : {(void*)fa, (void*)fw, (void*)A.hi, (void*)A.lo, (void*)O.hi, (void*)O.lo, (void*)W.hi, (void*)W.lo, (void*)dst}) hipFree(p);
    if (e != hipSuccess) return fail(-7, "stamps launch: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int f5hip_debug_gemm_bench(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t bn, int32_t variant, int32_t iters,
                                      double* avg_us) {
    if (M % 256 || N % 128 || K % 32 || !avg_us) return fail(-1, "debug_gemm_bench: bad shape");
    float *fa = nullptr, *fw = nullptr, *outf = nullptr;
    Plane2 A, O; PackedW W;
    const int pad = variant >= 100 ? 64 : 0;   // +128 B per row: breaks the power-of-two row stride
    if (variant >= 100) variant -= 100;
    const int lda = K + pad, ldw = K + pad;
    size_t na = (size_t)M * lda, nw = (size_t)N * ldw, no = (size_t)M * N;
    if (hipMalloc(&fa, na * 4) || hipMalloc(&fw, nw * 4) || hipMalloc(&outf, no * 4) || hipMalloc(&A.hi, na * 2) || hipMalloc(&A.lo, na * 2) ||
        hipMalloc(&O.hi, no * 2) || hipMalloc(&O.lo, no * 2) || hipMalloc(&W.hi, nw * 2) || hipMalloc(&W.lo, nw * 2))
        return fail(-5, "debug_gemm_bench: hipMalloc");
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((na + 255) / 256), dim3(256), 0, 0, fa, na, 1u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((nw + 255) / 256), dim3(256), 0, 0, fw, nw, 2u);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(M), dim3(256), 0, 0, fa, M, K, K, A.hi, A.lo, lda);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(N), dim3(256), 0, 0, fw, N, K, K, W.hi, W.lo, ldw);
    W.n = N; W.k = K; W.n_pad = N; W.k_pad = K; W.ld = ldw; W.bias = nullptr;
    GemmArgs a = gemm_base(A, lda, W, M);
    a.act = ACT_GELU_TANH; a.out_hi = O.hi; a.out_lo = O.lo; a.ldob = N;
    if (variant >= 10 && variant < 20) { a.act = ACT_NONE; a.out_hi = nullptr; a.out_lo = nullptr; a.res = outf; a.ldres = N; a.out_f32 = outf; a.ldo = N; variant -= 10; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipError_t e = hipSuccess;
    for (int it = -2; it < iters; it++) {
        if (it == 0) hipEventRecord(e0, 0);
        if (variant == 20) e = planes == 2 ? launch_gemm2_t<2, 128, 128, EPI_GENERIC>(a, M, N, 0) : launch_gemm2_t<1, 128, 128, EPI_GENERIC>(a, M, N, 0);
        else if (variant == 30) e = planes == 2 ? launch_gemm3_t<2, EPI_GENERIC>(a, M, N, 0) : launch_gemm3_t<1, EPI_GENERIC>(a, M, N, 0);
        else if (variant == 31) e = launch_gemm3_t<2, EPI_GENERIC, 1>(a, M, N, 0);
        else if (variant == 32) e = launch_gemm3_t<2, EPI_GENERIC, 2>(a, M, N, 0);
        else if (variant == 37) e = launch_gemm3_t<1, EPI_GENERIC, 0, 256>(a, M, N, 0);
        else if (variant == 33) e = launch_gemm3_t<1, EPI_GENERIC, 1>(a, M, N, 0);
        else if (variant == 34) e = launch_gemm3_t<1, EPI_GENERIC, 2>(a, M, N, 0);
        else if (variant == 35) e = launch_gemm3_t<1, EPI_GENERIC, 4>(a, M, N, 0);
        else if (variant == 36) e = launch_gemm3_t<2, EPI_GENERIC, 4>(a, M, N, 0);
        else if (variant == 21) e = planes == 2 ? launch_gemm2_t<2, 256, 128, EPI_GENERIC>(a, M, N, 0) : launch_gemm2_t<1, 256, 128, EPI_GENERIC>(a, M, N, 0);
        else if (planes == 2 && bn == 128) {
            if (variant == 0) e = launch_gemm_t<2, 128, false, EPI_GENERIC, 0>(a, M, N, 0);
            else if (variant == 1) e = launch_gemm_t<2, 128, false, EPI_GENERIC, 1>(a, M, N, 0);
            else e = launch_gemm_t<2, 128, false, EPI_GENERIC, 2>(a, M, N, 0);
        } else if (planes == 2) {
            if (variant == 0) e = launch_gemm_t<2, 64, false, EPI_GENERIC, 0>(a, M, N, 0);
            else if (variant == 1) e = launch_gemm_t<2, 64, false, EPI_GENERIC, 1>(a, M, N, 0);
            else e = launch_gemm_t<2, 64, false, EPI_GENERIC, 2>(a, M, N, 0);
        } else if (bn == 128) {
            e = launch_gemm_t<1, 128, false, EPI_GENERIC, 0>(a, M, N, 0);
        } else {
            e = launch_gemm_t<1, 64, false, EPI_GENERIC, 0>(a, M, N, 0);
        }
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *avg_us = (double)ms * 1e3 / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    for (void* p : {(void*)fa, (void*)fw, (void*)outf, (void*)A.hi, (void*)A.lo, (void*)O.hi, (void*)O.lo, (void*)W.hi, (void*)W.lo}) hipFree(p);
    if (e != hipSuccess) return fail(-7, "debug_gemm_bench launch: %s", hipGetErrorString(e));
    return 0;
}

// per-phase cycle totals of one wave of attn3 (ring wait + barrier | QK^T issue | softmax + PV) and the kernel's average duration.
// n_seq = 2 with 1152 < n <= 1536: the balanced 8-wave kernel of C2; n_seq > 2: the 4-wave kernel of the batch shapes; n <= 768: 64 queries per wave
extern "C" int f5hip_debug_attn_stamps(int32_t n, int32_t heads, int32_t n_seq, int32_t iters, unsigned long long* out, double* avg_us) {
    const int D = heads * 64, pitch = (n + 127) / 128 * 128, M = n_seq * pitch + 256;
    if (n_seq < 1 || n_seq > 64) return fail(-1, "attn stamps: n_seq");
    float* f = nullptr; __bf16 *qk = nullptr, *vt = nullptr, *oh = nullptr; int* meta = nullptr; unsigned long long* dbg = nullptr;
    const size_t nq = (size_t)M * 2 * D, nv = (size_t)D * M;
    if (hipMalloc(&f, nq * 4) || hipMalloc(&qk, nq * 2) || hipMalloc(&vt, nv * 2) || hipMalloc(&oh, (size_t)M * D * 2) ||
        hipMalloc(&meta, 3 * 64 * sizeof(int)) || hipMalloc(&dbg, 128)) return fail(-5, "attn stamps: hipMalloc");
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((nq + 255) / 256), dim3(256), 0, 0, f, nq, 3u);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(M), dim3(256), 0, 0, f, M, 2 * D, 2 * D, qk, (__bf16*)nullptr, 2 * D);   // one fp16 plane (q | k)
    hipLaunchKernelGGL(pack_weight_kernel, dim3(D), dim3(256), 0, 0, f, D, M, M, vt, (__bf16*)nullptr, M);
    int h_meta[3 * 64];
    for (int i = 0; i < n_seq; i++) { h_meta[i] = i * pitch; h_meta[64 + i] = n; h_meta[128 + i] = n; }
    hipMemcpy(meta, h_meta, sizeof(h_meta), hipMemcpyHostToDevice);
    hipMemset(dbg, 0, 128);
    AttnArgs at; memset(&at, 0, sizeof(at));
    at.qk = qk; at.vt = vt; at.D = D; at.ldvt = M; at.seq_row0 = meta; at.seq_len = meta + 64; at.seq_kvlen = meta + 128; at.out_hi = oh; at.f16_out = 1; at.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = -2; it < iters; it++) {
        if (it == 0) hipEventRecord(e0, 0);
        if (n <= 768 && n_seq == 2) hipLaunchKernelGGL((attn3_fwd_kernel<4, false, true, 9, 2>), dim3((n + 255) / 256, heads, n_seq), dim3(256), 0, 0, at);
        else if (n_seq == 2) hipLaunchKernelGGL((attn3_fwd_kernel<8, false, true, 9, 1, true>), dim3((n + 191) / 192, heads, n_seq), dim3(512), 0, 0, at);
        else hipLaunchKernelGGL((attn3_fwd_kernel<4, false, true, 5>), dim3((n + 127) / 128, heads, n_seq), dim3(256), 0, 0, at);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (avg_us) *avg_us = (double)ms * 1e3 / iters;
    hipMemcpy(out, dbg, 72, hipMemcpyDeviceToHost);
    hipError_t e = hipGetLastError();
    for (void* p : {(void*)f, (void*)qk, (void*)vt, (void*)oh, (void*)meta, (void*)dbg}) hipFree(p);
    if (e != hipSuccess) return fail(-7, "attn stamps: %s", hipGetErrorString(e));
    return 0;
}

// one-plane 128 x 128 vs 128 x 256 tile on the same operands: max |difference| and max |value| of the outputs.
// mode 0: bias + GELU -> fp32;  1: bias + GELU -> one fp16 plane;  2: QKV epilogue (N = 3 D: rotary q | k bf16 rows, V transposed)
extern "C" int f5hip_debug_gemm_wide_check(int32_t M, int32_t N, int32_t K, int32_t mode, double* max_diff, double* max_abs) {
    const bool f16 = mode >= 10;   // mode + 10: fp16 operands (PREC 3) instead of bf16
    if (f16) mode -= 10;
    if (M % 128 || N % 256 || K % 32 || (mode == 2 && N % 768)) return fail(-1, "wide_check: bad shape");
    const int D = N / 3;
    float *fa = nullptr, *fw = nullptr, *bias = nullptr, *rc = nullptr; Plane2 A; PackedW W; int* pos = nullptr;
    size_t na = (size_t)M * K, nw = (size_t)N * K, no = (size_t)M * N;
    char* o[2] = {nullptr, nullptr};
    if (hipMalloc(&fa, na * 4) || hipMalloc(&fw, nw * 4) || hipMalloc(&o[0], no * 4) || hipMalloc(&o[1], no * 4) || hipMalloc(&bias, N * 4) ||
        hipMalloc(&A.hi, na * 2) || hipMalloc(&A.lo, na * 2) || hipMalloc(&W.hi, nw * 2) || hipMalloc(&W.lo, nw * 2) || hipMalloc(&rc, 4096 * 32 * 4) ||
        hipMalloc(&pos, M * 4)) return fail(-5, "wide_check: hipMalloc");
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((na + 255) / 256), dim3(256), 0, 0, fa, na, 1u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((nw + 255) / 256), dim3(256), 0, 0, fw, nw, 2u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3((N + 255) / 256), dim3(256), 0, 0, bias, (size_t)N, 3u);
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(4096 * 32 / 256), dim3(256), 0, 0, rc, (size_t)4096 * 32, 4u);
    std::vector<int> hpos(M);
    for (int i = 0; i < M; i++) hpos[i] = i % 1404;
    hipMemcpy(pos, hpos.data(), M * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(M), dim3(256), 0, 0, fa, M, K, K, A.hi, f16 ? (__bf16*)nullptr : A.lo, K);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(N), dim3(256), 0, 0, fw, N, K, K, W.hi, f16 ? (__bf16*)nullptr : W.lo, K);
    W.n = N; W.k = K; W.n_pad = N; W.k_pad = K; W.ld = K; W.bias = bias;
    hipError_t e = hipSuccess;
    for (int v = 0; v < 2 && e == hipSuccess; v++) {
        hipMemset(o[v], 0, no * 4);
        GemmArgs a = gemm_base(A, K, W, M);
        if (mode == 2) {
            a.D = D; a.row_pos = pos; a.rope_cos = rc; a.rope_sin = rc + 64; a.qk = (__bf16*)o[v]; a.vt = (__bf16*)(o[v] + (size_t)M * 2 * D * 2); a.ldvt = M;
            if (f16) e = v == 0 ? launch_gemm3_t<3, EPI_QKV>(a, M, N, 0) : launch_gemm3_t<3, EPI_QKV, 0, 256>(a, M, N, 0);
            else e = v == 0 ? launch_gemm3_t<1, EPI_QKV>(a, M, N, 0) : launch_gemm3_t<1, EPI_QKV, 0, 256>(a, M, N, 0);
        } else {
            a.act = ACT_GELU_TANH;
            if (mode == 0) { a.out_f32 = (float*)o[v]; a.ldo = N; }
            else { a.out_hi = (__bf16*)o[v]; a.ldob = N; a.f16_out = 1; }
            if (f16) e = v == 0 ? launch_gemm3_t<3, EPI_GENERIC>(a, M, N, 0) : launch_gemm3_t<3, EPI_GENERIC, 0, 256>(a, M, N, 0);
            else e = v == 0 ? launch_gemm3_t<1, EPI_GENERIC>(a, M, N, 0) : launch_gemm3_t<1, EPI_GENERIC, 0, 256>(a, M, N, 0);
        }
    }
    hipDeviceSynchronize();
    const size_t nbytes = mode == 0 ? no * 4 : (mode == 1 ? no * 2 : no * 2);
    std::vector<unsigned char> h1(nbytes), h2(nbytes);
    hipMemcpy(h1.data(), o[0], nbytes, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), o[1], nbytes, hipMemcpyDeviceToHost);
    double md = 0, ma = 0; size_t worst = 0, ndiff = 0;
    if (mode == 0) {
        const float *x = (const float*)h1.data(), *y = (const float*)h2.data();
        for (size_t i = 0; i < no; i++) { double d = fabs((double)x[i] - y[i]); if (d > md) { md = d; worst = i; } ma = fmax(ma, fabs((double)x[i])); ndiff += d != 0; }
    } else {
        const unsigned short *x = (const unsigned short*)h1.data(), *y = (const unsigned short*)h2.data();
        for (size_t i = 0; i < no; i++) if (x[i] != y[i]) { if (!ndiff) worst = i; ndiff++; }
        md = (double)ndiff;
    }
    *max_diff = md; *max_abs = ma;
    fprintf(stderr, "[wide_check] mode %d: %zu differing elements, first / worst at flat index %zu (row %zu col %zu)\n", mode, ndiff, worst, worst / (mode == 2 ? 2 * D : N),
            worst % (mode == 2 ? 2 * D : N));
    for (void* q : {(void*)fa, (void*)fw, (void*)o[0], (void*)o[1], (void*)bias, (void*)A.hi, (void*)A.lo, (void*)W.hi, (void*)W.lo, (void*)rc, (void*)pos}) hipFree(q);
    if (e != hipSuccess) return fail(-7, "wide_check launch: %s", hipGetErrorString(e));
    return 0;
}
