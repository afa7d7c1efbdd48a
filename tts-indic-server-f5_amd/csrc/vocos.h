// Vocos (vocos 0.1.0, charactr/vocos-mel-24khz geometry) mel -> waveform decoder and the torchaudio-style
// mel front-end, on the same GEMM / LayerNorm kernels as the DiT plus LDS radix-2 FFT kernels for
// iSTFT / STFT.  Included at the end of f5hip.hip (same translation unit).
//
// decode(mel[B,100,T]):
//   x = Conv1d(100->512,k7,p3)(mel) -> LN -> 8 x [dwconv k7 -> LN -> 512->1536 -> GELU(erf) -> 1536->512 -> gamma* -> +res]
//   -> LN -> Linear(512->1026) -> (mag = min(exp(.),1e2), phase) -> irfft(1024) * hann -> overlap-add / envelope
#pragma once

struct VocosBlock {
    float *dw_w = nullptr, *dw_b = nullptr, *ln_w = nullptr, *ln_b = nullptr, *gamma = nullptr;
    PackedW pw1, pw2;
};

struct f5hip_vocos {
    f5hip_vocos_config cfg;
    int nsplit = 2;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    PackedW embed, head;
    float *norm_w = nullptr, *norm_b = nullptr, *fnorm_w = nullptr, *fnorm_b = nullptr;
    std::vector<VocosBlock> blk;
    float* window = nullptr;     // periodic hann [n_fft]
    float2* twiddle = nullptr;   // (cos, sin)(2 pi k / n_fft), k < n_fft/2
    // workspace
    int cap_rows = 0;
    void* ws = nullptr;
    float *x = nullptr, *y = nullptr, *fw = nullptr;
    Plane2 melp, tn, hid;
    int* meta = nullptr;
};

// ---------------------------------------------------------------------------------------- FFT in LDS
// In-place radix-2 decimation-in-time FFT of 1024 complex points held in LDS in bit-reversed order.
// sign = +1: sum_k X[k] e^{+2 pi i k n / N} (inverse, unnormalised); sign = -1: forward.  256 threads.
F5_DEVICE void fft1024_lds(float2* s, const float2* __restrict__ tw, int tid, float sign) {
#pragma unroll 1
    for (int stage = 0; stage < 10; stage++) {
        const int half = 1 << stage;
        __syncthreads();
#pragma unroll
        for (int b = tid; b < 512; b += 256) {
            const int grp = b >> stage, pos = b & (half - 1);
            const int i0 = (grp << (stage + 1)) + pos, i1 = i0 + half;
            float2 w = tw[pos << (9 - stage)];
            w.y *= sign;
            const float2 a = s[i0], c = s[i1];
            const float tx = c.x * w.x - c.y * w.y, ty = c.x * w.y + c.y * w.x;
            s[i0] = make_float2(a.x + tx, a.y + ty);
            s[i1] = make_float2(a.x - tx, a.y - ty);
        }
    }
    __syncthreads();
}

// mel [B][C][T] fp32 -> rows (frame-major) split bf16 [M_pad][128]
__global__ __launch_bounds__(128) void mel_to_rows_kernel(const float* mel, int C, int T, const int* row_seq, const int* row_pos,
                                                          int M, __bf16* hi, __bf16* lo) {
    const int row = blockIdx.x, c = threadIdx.x;
    if (row >= M) return;
    const int b = row_seq[row];
    float v = 0.0f;
    if (b >= 0 && c < C) v = mel[((size_t)b * C + c) * T + row_pos[row]];
    __bf16 h, l;
    split_bf16(v, h, l);
    hi[(size_t)row * 128 + c] = h;
    lo[(size_t)row * 128 + c] = l;
}

// ISTFT head, per frame: y[row] = [log-mag (513) | phase (513)] -> windowed irfft frame fw[row][1024]
__global__ __launch_bounds__(256) void istft_frame_kernel(const float* y, int ldy, const int* row_seq, int M, const float* window,
                                                          const float2* tw, float* fw) {
    __shared__ float2 s[1024];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (row >= M || row_seq[row] < 0) return;
    const float* yr = y + (size_t)row * ldy;
    for (int k = tid; k <= 512; k += 256) {
        float mag = fminf(expf(yr[k]), 100.0f);   // torch.clip(exp(mag), max=1e2)
        const float ph = yr[513 + k];
        float re = mag * cosf(ph), im = mag * sinf(ph);
        if (k == 0 || k == 512) im = 0.0f;        // irfft ignores the imaginary part of DC / Nyquist
        s[__brev((unsigned)k) >> 22] = make_float2(re, im);
        if (k > 0 && k < 512) s[__brev((unsigned)(1024 - k)) >> 22] = make_float2(re, -im);
    }
    fft1024_lds(s, tw, tid, 1.0f);
    for (int n = tid; n < 1024; n += 256) fw[(size_t)row * 1024 + n] = s[n].x * (1.0f / 1024.0f) * window[n];
}

// overlap-add + window-envelope normalisation + centre trim (torch.istft, center=True): out[b][hop*(T-1)]
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* fw, const int* seq_row0, int T, int hop, const float* window,
                                                        float* out) {
    const int b = blockIdx.y;
    const int L = hop * (T - 1);
    const int sidx = blockIdx.x * 256 + threadIdx.x;
    if (sidx >= L) return;
    const int p = sidx + 512;
    int t_hi = p / hop; if (t_hi > T - 1) t_hi = T - 1;
    int t_lo = (p - 1023 + hop - 1) / hop; if (t_lo < 0) t_lo = 0;
    float val = 0.0f, env = 0.0f;
    for (int t = t_lo; t <= t_hi; t++) {
        const int n = p - t * hop;
        val += fw[(size_t)(seq_row0[b] + t) * 1024 + n];
        env += window[n] * window[n];
    }
    out[(size_t)b * L + sidx] = val / env;
}

// STFT magnitude -> HTK mel filterbank -> log(clamp(., 1e-5)); one block per (frame, batch)
__global__ __launch_bounds__(256) void mel_frame_kernel(const float* wave, int nw, int T, int hop, int n_mels, const float* window,
                                                        const float2* tw, const float* fb /*[513][n_mels]*/, float* mel, int pad, float mag_eps) {
    __shared__ float2 s[1024];
    __shared__ float mag[520];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* w = wave + (size_t)b * nw;
    for (int n = tid; n < 1024; n += 256) {
        int idx = t * hop - pad + n;               // reflect padding by `pad` samples on both sides
        if (idx < 0) idx = -idx;
        if (idx >= nw) idx = 2 * (nw - 1) - idx;
        s[__brev((unsigned)n) >> 22] = make_float2(w[idx] * window[n], 0.0f);
    }
    fft1024_lds(s, tw, tid, -1.0f);
    for (int k = tid; k <= 512; k += 256) mag[k] = sqrtf(s[k].x * s[k].x + s[k].y * s[k].y + mag_eps);   // power = 1
    __syncthreads();
    if (tid < n_mels) {
        float acc = 0.0f;
        for (int k = 0; k <= 512; k++) acc += fb[k * n_mels + tid] * mag[k];
        mel[((size_t)b * n_mels + tid) * T + t] = logf(fmaxf(acc, 1e-5f));
    }
}

// ---------------------------------------------------------------------------------------- host side
static int make_fft_tables(float** window, float2** twiddle, int n_fft) {
    std::vector<float> w(n_fft);
    std::vector<float2> tw(n_fft / 2);
    for (int n = 0; n < n_fft; n++) w[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)n_fft));   // periodic hann
    for (int k = 0; k < n_fft / 2; k++) {
        tw[k].x = (float)cos(2.0 * M_PI * (double)k / (double)n_fft);
        tw[k].y = (float)sin(2.0 * M_PI * (double)k / (double)n_fft);
    }
    if (upload_f32(window, w.data(), w.size())) return -4;
    if (hipMalloc((void**)twiddle, sizeof(float2) * tw.size()) != hipSuccess) return fail(-4, "hipMalloc twiddle");
    if (hipMemcpy(*twiddle, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(-4, "H2D twiddle");
    return 0;
}

f5hip_vocos* f5hip_vocos_create(const f5hip_vocos_config* cfg) {
    if (!cfg) { set_error("null config"); return nullptr; }
    if (cfg->n_fft != 1024 || cfg->hop_length != 256 || cfg->in_channels > 128 || cfg->dim % 128 || cfg->intermediate_dim % 128 ||
        (cfg->gemm_planes != 1 && cfg->gemm_planes != 2)) {
        set_error("unsupported Vocos geometry (need n_fft 1024, hop 256, in_channels <= 128, dim %% 128 == 0)");
        return nullptr;
    }
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) { set_error("no HIP device: libf5hip has no CPU fallback"); return nullptr; }
    f5hip_vocos* v = new f5hip_vocos();
    v->cfg = *cfg;
    v->nsplit = cfg->gemm_planes;
    return v;
}

void f5hip_vocos_destroy(f5hip_vocos* v) {
    if (!v) return;
    free_packed(v->embed); free_packed(v->head);
    for (float* p : {v->norm_w, v->norm_b, v->fnorm_w, v->fnorm_b, v->window}) dev_free(p);
    dev_free(v->twiddle);
    for (auto& b : v->blk) {
        for (float* p : {b.dw_w, b.dw_b, b.ln_w, b.ln_b, b.gamma}) dev_free(p);
        free_packed(b.pw1); free_packed(b.pw2);
    }
    dev_free(v->ws); dev_free(v->meta);
    delete v;
}

int f5hip_vocos_load_param(f5hip_vocos* v, const char* name, const float* data, int64_t numel) {
    if (!v || !name || !data || numel <= 0) return fail(-1, "load_param: bad argument");
    if (v->finalized) return fail(-2, "load_param after finalize");
    v->host[name].assign(data, data + numel);
    return 0;
}

#define VGETP(var, name, numel)                                                                                     \
    const std::vector<float>* var = nullptr;                                                                        \
    {                                                                                                               \
        auto it = v->host.find(name);                                                                               \
        if (it == v->host.end()) return fail(-3, "missing parameter %s", std::string(name).c_str());                \
        if ((int64_t)it->second.size() != (int64_t)(numel)) return fail(-3, "parameter %s: wrong size", std::string(name).c_str()); \
        var = &it->second;                                                                                          \
    }

int f5hip_vocos_finalize(f5hip_vocos* v) {
    if (!v) return fail(-1, "null vocoder");
    if (v->finalized) return 0;
    const f5hip_vocos_config& c = v->cfg;
    const int C = c.in_channels, D = c.dim, I = c.intermediate_dim, NO = c.n_fft + 2;
    {   // embed Conv1d(C -> D, k=7) as a dense implicit GEMM: K = 7 taps x 128 (channels padded)
        VGETP(w, "backbone.embed.weight", (int64_t)D * C * 7); VGETP(b, "backbone.embed.bias", D);
        std::vector<float> wp((size_t)D * 7 * 128, 0.0f);
        for (int co = 0; co < D; co++)
            for (int ci = 0; ci < C; ci++)
                for (int tap = 0; tap < 7; tap++) wp[(size_t)co * 896 + tap * 128 + ci] = (*w)[((size_t)co * C + ci) * 7 + tap];
        if (pack_linear(v->embed, wp.data(), D, 896, 896, b->data())) return -4;
    }
    VGETP(nw, "backbone.norm.weight", D); VGETP(nb, "backbone.norm.bias", D);
    VGETP(fw_, "backbone.final_layer_norm.weight", D); VGETP(fb_, "backbone.final_layer_norm.bias", D);
    if (upload_f32(&v->norm_w, nw->data(), D) || upload_f32(&v->norm_b, nb->data(), D) || upload_f32(&v->fnorm_w, fw_->data(), D) ||
        upload_f32(&v->fnorm_b, fb_->data(), D)) return -4;
    v->blk.resize(c.num_layers);
    for (int i = 0; i < c.num_layers; i++) {
        std::string p = "backbone.convnext." + std::to_string(i) + ".";
        VocosBlock& b = v->blk[i];
        VGETP(dw, p + "dwconv.weight", (int64_t)D * 7); VGETP(db, p + "dwconv.bias", D);
        VGETP(lw, p + "norm.weight", D); VGETP(lb, p + "norm.bias", D);
        VGETP(w1, p + "pwconv1.weight", (int64_t)I * D); VGETP(b1, p + "pwconv1.bias", I);
        VGETP(w2, p + "pwconv2.weight", (int64_t)D * I); VGETP(b2, p + "pwconv2.bias", D);
        VGETP(gm, p + "gamma", D);
        if (upload_f32(&b.dw_w, dw->data(), dw->size()) || upload_f32(&b.dw_b, db->data(), D) || upload_f32(&b.ln_w, lw->data(), D) ||
            upload_f32(&b.ln_b, lb->data(), D) || upload_f32(&b.gamma, gm->data(), D)) return -4;
        if (pack_linear(b.pw1, w1->data(), I, D, D, b1->data())) return -4;
        if (pack_linear(b.pw2, w2->data(), D, I, I, b2->data())) return -4;
    }
    {
        VGETP(w, "head.out.weight", (int64_t)NO * D); VGETP(b, "head.out.bias", NO);
        if (pack_linear(v->head, w->data(), NO, D, D, b->data())) return -4;
    }
    if (make_fft_tables(&v->window, &v->twiddle, c.n_fft)) return -4;
    v->host.clear();
    v->finalized = true;
    return 0;
}

int f5hip_vocos_decode(f5hip_vocos* v, int32_t batch, int32_t frames, const float* mel_dev, float* wave_dev, void* stream) {
    if (!v || !v->finalized) return fail(-1, "vocoder not finalized");
    if (batch <= 0 || frames < 2 || !mel_dev || !wave_dev) return fail(-1, "vocos_decode: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const f5hip_vocos_config& c = v->cfg;
    const int D = c.dim, I = c.intermediate_dim, T = frames, Tp = ceil_to(T, 128), M = batch * Tp, LDY = 1152;
    if (M > v->cap_rows) {
        dev_free(v->ws); dev_free(v->meta);
        Arena a;
        for (int pass = 0; pass < 2; pass++) {
            a.reset(pass ? (char*)v->ws : nullptr);
            v->x = a.f32((size_t)M * D); v->y = a.f32((size_t)M * LDY); v->fw = a.f32((size_t)M * 1024);
            v->melp = a.plane2((size_t)M * 128 + 1024); v->tn = a.plane2((size_t)M * D); v->hid = a.plane2((size_t)M * I);
            if (!pass) {
                if (hipMalloc(&v->ws, a.used()) != hipSuccess) { v->ws = nullptr; v->cap_rows = 0; return fail(-5, "hipMalloc vocos workspace"); }
                if (hipMemset(v->ws, 0, a.used()) != hipSuccess) return fail(-5, "hipMemset vocos workspace");
            }
        }
        if (hipMalloc((void**)&v->meta, sizeof(int) * ((size_t)M * 4 + batch)) != hipSuccess) { v->meta = nullptr; return fail(-5, "hipMalloc vocos meta"); }
        v->cap_rows = M;
    }
    std::vector<int> h((size_t)M * 4 + batch, 0);
    int* row_seq = &h[0]; int* row_pos = row_seq + M; int* row_start = row_pos + M; int* row_end = row_start + M; int* seq_row0 = row_end + M;
    for (int r = 0; r < M; r++) row_seq[r] = -1;
    for (int b = 0; b < batch; b++) {
        seq_row0[b] = b * Tp;
        for (int t = 0; t < T; t++) { const int r = b * Tp + t; row_seq[r] = b; row_pos[r] = t; row_start[r] = b * Tp; row_end[r] = b * Tp + T; }
    }
    if (hipMemcpyAsync(v->meta, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(-6, "vocos metadata upload");
    const int *d_row_seq = v->meta, *d_row_pos = v->meta + M, *d_row_start = v->meta + 2 * M, *d_row_end = v->meta + 3 * M, *d_seq_row0 = v->meta + 4 * M;

    prof_begin(PROF_VOCOS, st);
    hipLaunchKernelGGL(mel_to_rows_kernel, dim3(M), dim3(128), 0, st, mel_dev, c.in_channels, T, d_row_seq, d_row_pos, M, v->melp.hi, v->melp.lo);
    CKL("mel_to_rows");
    // embed conv -> x (fp32), then LayerNorm in place
    GemmArgs e = gemm_base(v->melp, 128, v->embed, M);
    e.conv_kpt = 4; e.conv_center = 3; e.conv_group_cols = 0; e.row_seq_start = d_row_start; e.row_seq_end = d_row_end;
    e.out_f32 = v->x; e.ldo = D;
    CK(run_gemm_n(v->nsplit, M, e, v->embed, EPI_GENERIC, true, 128, st));
    LnArgs ln; memset(&ln, 0, sizeof(ln));
    ln.x = v->x; ln.ldx = D; ln.M = M; ln.D = D; ln.scale = v->norm_w; ln.shift = v->norm_b; ln.gain_off = 0.0f; ln.eps = 1e-6f;
    ln.out_f32 = v->x; ln.ldof = D;
    CK(run_ln(ln, st));
    for (int i = 0; i < c.num_layers; i++) {
        VocosBlock& b = v->blk[i];
        LnArgs l2; memset(&l2, 0, sizeof(l2));
        l2.x = v->x; l2.ldx = D; l2.M = M; l2.D = D; l2.scale = b.ln_w; l2.shift = b.ln_b; l2.gain_off = 0.0f; l2.eps = 1e-6f;
        l2.dw_w = b.dw_w; l2.dw_b = b.dw_b; l2.row_seq_start = d_row_start; l2.row_seq_end = d_row_end;
        l2.out_hi = v->tn.hi; l2.out_lo = v->tn.lo; l2.ldo = D;
        CK(run_ln(l2, st));
        GemmArgs g1 = gemm_base(v->tn, D, b.pw1, M);
        g1.act = ACT_GELU_ERF; g1.out_hi = v->hid.hi; g1.out_lo = v->hid.lo; g1.ldob = I;
        CK(run_gemm_n(v->nsplit, M, g1, b.pw1, EPI_GENERIC, false, 128, st));
        GemmArgs g2 = gemm_base(v->hid, I, b.pw2, M);
        g2.mul = b.gamma; g2.res = v->x; g2.ldres = D; g2.out_f32 = v->x; g2.ldo = D;
        CK(run_gemm_n(v->nsplit, M, g2, b.pw2, EPI_GENERIC, false, 64, st));
    }
    LnArgs lf; memset(&lf, 0, sizeof(lf));
    lf.x = v->x; lf.ldx = D; lf.M = M; lf.D = D; lf.scale = v->fnorm_w; lf.shift = v->fnorm_b; lf.gain_off = 0.0f; lf.eps = 1e-6f;
    lf.out_hi = v->tn.hi; lf.out_lo = v->tn.lo; lf.ldo = D;
    CK(run_ln(lf, st));
    GemmArgs hd = gemm_base(v->tn, D, v->head, M);
    hd.out_f32 = v->y; hd.ldo = LDY;
    CK(run_gemm_n(v->nsplit, M, hd, v->head, EPI_GENERIC, false, 128, st));
    hipLaunchKernelGGL(istft_frame_kernel, dim3(M), dim3(256), 0, st, v->y, LDY, d_row_seq, M, v->window, v->twiddle, v->fw);
    CKL("istft_frame");
    const int L = c.hop_length * (T - 1);
    hipLaunchKernelGGL(istft_ola_kernel, dim3((L + 255) / 256, batch), dim3(256), 0, st, v->fw, d_seq_row0, T, c.hop_length, v->window, wave_dev);
    CKL("istft_ola");
    prof_end(PROF_VOCOS, st);
    return 0;
}

// ---------------------------------------------------------------------------------------- mel front-end
struct MelTables { int n_fft = 0, n_mels = 0, sr = 0; float* window = nullptr; float2* twiddle = nullptr; float* fb = nullptr; };
static MelTables g_mel;

static double hz_to_mel_htk(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel_to_hz_htk(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

int f5hip_mel_spectrogram(int32_t batch, int32_t n_samples, const float* wave_dev, float* mel_dev, int32_t n_fft, int32_t hop_length,
                          int32_t n_mels, int32_t sample_rate, void* stream) {
    if (batch <= 0 || n_samples <= n_fft / 2 || !wave_dev || !mel_dev) return fail(-1, "mel_spectrogram: bad argument");
    if (n_fft != 1024 || n_mels > 256) return fail(-1, "mel_spectrogram: only n_fft = 1024, n_mels <= 256");
    hipStream_t st = (hipStream_t)stream;
    if (g_mel.n_fft != n_fft || g_mel.n_mels != n_mels || g_mel.sr != sample_rate) {
        dev_free(g_mel.window); dev_free(g_mel.twiddle); dev_free(g_mel.fb);
        g_mel = MelTables();
        if (make_fft_tables(&g_mel.window, &g_mel.twiddle, n_fft)) return -4;
        // torchaudio.functional.melscale_fbanks(n_freqs = 513, f_min = 0, f_max = sr/2, n_mels, sr, norm=None, "htk")
        const int nf = n_fft / 2 + 1;
        const double fmax = sample_rate / 2;
        std::vector<float> fb((size_t)nf * n_mels, 0.0f);
        std::vector<double> fpts(n_mels + 2);
        const double m0 = hz_to_mel_htk(0.0), m1 = hz_to_mel_htk(fmax);
        for (int i = 0; i < n_mels + 2; i++) fpts[i] = mel_to_hz_htk(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
        for (int k = 0; k < nf; k++) {
            const double f = fmax * (double)k / (double)(nf - 1);
            for (int j = 0; j < n_mels; j++) {
                const double down = (f - fpts[j]) / (fpts[j + 1] - fpts[j]);
                const double up = (fpts[j + 2] - f) / (fpts[j + 2] - fpts[j + 1]);
                const double val = std::max(0.0, std::min(down, up));
                fb[(size_t)k * n_mels + j] = (float)val;
            }
        }
        if (upload_f32(&g_mel.fb, fb.data(), fb.size())) return -4;
        g_mel.n_fft = n_fft; g_mel.n_mels = n_mels; g_mel.sr = sample_rate;
    }
    const int T = 1 + n_samples / hop_length;
    prof_begin(PROF_OTHER, st);
    hipLaunchKernelGGL(mel_frame_kernel, dim3(T, batch), dim3(256), 0, st, wave_dev, n_samples, T, hop_length, n_mels, g_mel.window,
                       g_mel.twiddle, g_mel.fb, mel_dev, n_fft / 2, 0.0f);
    prof_end(PROF_OTHER, st);
    CKL("mel_frame");
    return 0;
}

// ---- BigVGAN-style mel (F/model/modules.py:30-72): librosa Slaney filterbank, center=False after a reflect pad of (n_fft - hop) / 2
static MelTables g_mel_bv;

int f5hip_mel_spectrogram_bigvgan(int32_t batch, int32_t n_samples, const float* wave_dev, float* mel_dev, int32_t n_fft, int32_t hop_length,
                                  int32_t n_mels, int32_t sample_rate, void* stream) {
    if (batch <= 0 || n_samples < n_fft || !wave_dev || !mel_dev) return fail(-1, "mel_spectrogram_bigvgan: bad argument");
    if (n_fft != 1024 || n_mels > 256) return fail(-1, "mel_spectrogram_bigvgan: only n_fft = 1024, n_mels <= 256");
    hipStream_t st = (hipStream_t)stream;
    if (g_mel_bv.n_fft != n_fft || g_mel_bv.n_mels != n_mels || g_mel_bv.sr != sample_rate) {
        dev_free(g_mel_bv.window); dev_free(g_mel_bv.twiddle); dev_free(g_mel_bv.fb);
        g_mel_bv = MelTables();
        if (make_fft_tables(&g_mel_bv.window, &g_mel_bv.twiddle, n_fft)) return -4;
        // librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm="slaney")
        const int nf = n_fft / 2 + 1;
        const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
        auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp; };
        auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m; };
        const double fmax = sample_rate / 2.0, m0 = hz_to_mel(0.0), m1 = hz_to_mel(fmax);
        std::vector<double> mf(n_mels + 2);
        for (int i = 0; i < n_mels + 2; i++) mf[i] = mel_to_hz(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
        std::vector<float> fb((size_t)nf * n_mels, 0.0f);
        for (int k = 0; k < nf; k++) {
            const double f = fmax * (double)k / (double)(nf - 1);
            for (int j = 0; j < n_mels; j++) {
                const double lower = (f - mf[j]) / (mf[j + 1] - mf[j]), upper = (mf[j + 2] - f) / (mf[j + 2] - mf[j + 1]);
                const double w = std::max(0.0, std::min(lower, upper)) * (2.0 / (mf[j + 2] - mf[j]));
                fb[(size_t)k * n_mels + j] = (float)w;
            }
        }
        if (upload_f32(&g_mel_bv.fb, fb.data(), fb.size())) return -4;
        g_mel_bv.n_fft = n_fft; g_mel_bv.n_mels = n_mels; g_mel_bv.sr = sample_rate;
    }
    const int pad = (n_fft - hop_length) / 2;
    const int T = (n_samples + 2 * pad - n_fft) / hop_length + 1;
    prof_begin(PROF_OTHER, st);
    hipLaunchKernelGGL(mel_frame_kernel, dim3(T, batch), dim3(256), 0, st, wave_dev, n_samples, T, hop_length, n_mels, g_mel_bv.window,
                       g_mel_bv.twiddle, g_mel_bv.fb, mel_dev, pad, 1e-9f);
    prof_end(PROF_OTHER, st);
    CKL("mel_frame (bigvgan)");
    return 0;
}
