// libf5hip: C ABI (include/f5hip.h) + host orchestration of the DiT / CFM path on one MI355X.
// One process per GPU; every call enqueues its kernels on the caller's HIP stream.
#include "../../include/f5hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <algorithm>
#include <vector>

#include "common.h"
#include "elementwise.h"
#include "gemm_launch.h"
#include "host_util.h"
#ifdef F5HIP_EXPERIMENTS   // measured-and-rejected kernels kept for A/B (stream-K GEMM, 8-wave all-consume GEMM, earlier attention kernels)
#include "experiments/attn2.h"
#include "experiments/gemm2.h"
#include "experiments/gemm4.h"
#include "gemm.h"
#include "attn3.h"
#endif

// =================================================================================================
// DiT model
// =================================================================================================

struct TextBlock {
    float *dw_w = nullptr, *dw_b = nullptr, *ln_w = nullptr, *ln_b = nullptr, *gamma = nullptr, *beta = nullptr;
    PackedW pw1, pw2;
};

struct f5hip_dit {
    f5hip_dit_config cfg;
    int nsplit = 2;       // operand planes of the state-touching GEMMs (1 bf16, 2 split bf16)
#ifdef F5HIP_EXPERIMENTS
    StreamKWs sk;         // stream-K partial-tile slots + flags (experiments/gemm4.h), owned by the handle: launches of one handle are stream-ordered
#endif
    // per-handle settings (the process-wide setters are only their defaults)
    int attn_invariant = -1;          // f5hip_dit_set_attention_shape_invariant: -1 = follow f5hip_set_attention_shape_invariant
    ProfState* prof = nullptr;        // f5hip_dit_set_profiling: this handle's own HIP-event spans and totals (null: the process-wide state)
    HostStage up_meta, up_time[2];    // pinned staging of the per-call uploads (row metadata; the two planes of the sinusoid table)
    bool blk_f16 = false; // gemm_planes == 3: transformer-block GEMMs (QKV, out, FF1, FF2) take one fp16 plane per operand
    bool skip_f16 = false; // UNetT (experiment, F5HIP_UNETT_SKIP_F16=1): the U-skip projections too
    // LayerNorm fused behind the residual GEMMs (gemm5 LNE kernels: experiments builds only, measured slower): per-handle arrival
    // counters and a host-visible time-out flag
    unsigned* ln_sync = nullptr;   // [16] row slabs, monotonic
    unsigned ln_epoch = 0;         // fused launches since the counters were zeroed (every one adds 16 arrivals to each of its slabs)
    int ln_slabs = 0;              // row slabs of those launches: a launch with another count zeroes the counters first (stream-ordered)
    int* ln_err = nullptr;         // host-mapped: a kernel sets it when its slab barrier timed out
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    // packed weights
    PackedW time1, time2, adaln, wx, wct, conv1, conv2, proj_out;
    std::vector<PackedW> wqkv, wout, wff1, wff2, wskip;   // wskip: UNetT skip projections (later half of the layers)
    std::vector<PackedW> wqkv_c, wout_c, wff1_c, wff2_c;   // MMDiT text-stream weights (wout_c / wff*_c: all but the last, context-pre-only block)
    std::vector<int> mod_c, mod_x;                         // MMDiT: offsets of the blocks' text / audio modulation vectors inside one row of `mod`
    int mod_final = 0;
    std::vector<float*> g_attn, g_ff;                     // UNetT RMSNorm gains
    float *g_out = nullptr, *zeros = nullptr;
    std::vector<TextBlock> tblk;
    float *text_emb = nullptr, *text_pos = nullptr, *rope_cos = nullptr, *rope_sin = nullptr;
    float *rope_row_cos = nullptr, *rope_row_sin = nullptr;   // [rows][32]: the rotary factors of every row of the current layout (workspace; rope_rows_kernel)
    int rope_max_pos = 0;
    int arch = 0;     // 0 = DiT (F5-TTS), 1 = UNetT (E2-TTS): one extra row per sequence carries the time token, 2 = MMDiT: the text tokens of
                      // every sequence are rows of their own stream, laid out behind all audio rows (rows [M, M + Mc))
    int td_pad = 0;   // text_dim rounded up to 32 (K padding of the step-invariant input-projection operand)
    int gw = 0;       // conv_pos_embed channels per group
    int n_adaln = 0;  // depth * 6 D + 2 D
    // workspace
    int cap_rows = 0, cap_frames = 0, cap_seq = 0;
    DevBuf ws;   // one arena, carved below
    float *h = nullptr, *h0 = nullptr, *ce = nullptr, *pred = nullptr, *te = nullptr, *ty = nullptr, *gx = nullptr,
          *mod = nullptr, *xstate = nullptr, *xmid = nullptr, *temb = nullptr;
    int ode_method = 0;   // 0 = Euler, 1 = explicit midpoint (f5hip_dit_set_ode_method)
    std::vector<Plane2> skipbuf;
    Plane2 hn, c1, ao, ff, xs, tn, tg, act, sinp, t1, st;
    __bf16 *qk = nullptr, *vt = nullptr;
    int *meta = nullptr;   // device int arena
    int meta_cap = 0;
    // per-call metadata (device pointers into `meta`)
    int *d_row_pos, *d_row_start, *d_row_end, *d_row_seq, *d_row_token, *d_row_frame, *d_row_condframe, *d_row_keep,
        *d_seq_row0, *d_seq_len, *d_seq_kvlen, *d_urow_c, *d_urow_u, *d_frame_is_cond;
    int M = 0, M_pad = 0, n_seq = 0, n_frames = 0, max_len = 0;
    int Mc = 0, Rtot = 0;   // MMDiT: text-stream rows and all rows (= row pitch of the V^T buffer); Rtot == M otherwise
    int *d_j_row0 = nullptr, *d_j_len = nullptr, *d_j_kvlen = nullptr, *d_j_kv_row0 = nullptr, *d_j_kv2_row0 = nullptr, *d_j_kv2_len = nullptr;   // MMDiT joint attention: 2 n_seq pseudo-sequences
    bool any_masked = false;
    std::vector<int> h_seq_len;
};

static int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

f5hip_dit* f5hip_dit_create(const f5hip_dit_config* cfg) {
    if (!cfg) { set_error("null config"); return nullptr; }
    if (cfg->dim % 128 || cfg->dim != cfg->heads * 64 || cfg->dim % 16 || cfg->text_dim % 4 || cfg->mel_dim > 128 || cfg->mel_dim % 4 ||
        cfg->dim / 16 > 64 || cfg->gemm_planes < 1 || cfg->gemm_planes > 3 || cfg->arch < 0 || cfg->arch > 2 ||
        (cfg->arch == 1 && (cfg->conv_layers != 0 || cfg->depth % 2)) || (cfg->conv_layers > 0 && cfg->text_dim % 32) ||
        (cfg->arch == 2 && (cfg->conv_layers != 0 || cfg->text_dim != cfg->dim || cfg->depth < 1))) {
        set_error("unsupported backbone geometry (need dim %% 128 == 0, dim == 64*heads, dim/16 <= 64, mel_dim <= 128, text conv needs text_dim %% 32 == 0; UNetT: even depth, no text conv)");
        return nullptr;
    }
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) {
        set_error("no HIP device: libf5hip has no CPU fallback");
        return nullptr;
    }
    f5hip_dit* m = new f5hip_dit();
    m->cfg = *cfg;
    m->nsplit = cfg->gemm_planes == 3 ? 2 : cfg->gemm_planes;
    // both backbones: against the reference's own digests mixed mode measures 3.1e-4 rms (F5-Base, 32 NFE) and 4.9e-4 (E2-Base, N = 2340,
    // 64 NFE) of the 1e-3 bound; the U-skip projections, the final norm + proj_out and the input embedding stay split bf16
    m->blk_f16 = cfg->gemm_planes == 3;
    m->skip_f16 = m->blk_f16 && cfg->arch == 1 && getenv("F5HIP_UNETT_SKIP_F16") && atoi(getenv("F5HIP_UNETT_SKIP_F16")) == 1;
    m->arch = cfg->arch;
    if (hipMalloc((void**)&m->ln_sync, 16 * sizeof(unsigned)) != hipSuccess || hipMemset(m->ln_sync, 0, 16 * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void**)&m->ln_err, sizeof(int), hipHostMallocMapped) != hipSuccess) {
        set_error("hipMalloc LayerNorm-fusion state");
        dev_free(m->ln_sync);
        delete m;
        return nullptr;
    }
    *m->ln_err = 0;
    m->td_pad = cfg->arch == 2 ? 0 : ceil_to(cfg->text_dim, 32);   // MMDiT: the text never enters the input projection (mmdit.py:64-70)
    m->gw = cfg->dim / 16;
    m->n_adaln = cfg->arch == 0 ? cfg->depth * 6 * cfg->dim + 2 * cfg->dim : 0;
    if (cfg->arch == 2) {   // per block [text: 6 D, or 2 D in the last (context-pre-only) block][audio: 6 D], then the final 2 D
        int off = 0;
        for (int l = 0; l < cfg->depth; l++) {
            m->mod_c.push_back(off); off += (l == cfg->depth - 1 ? 2 : 6) * cfg->dim;
            m->mod_x.push_back(off); off += 6 * cfg->dim;
        }
        m->mod_final = off;
        m->n_adaln = off + 2 * cfg->dim;
    }
    return m;
}

static void free_packed(PackedW& w) { dev_free(w.hi); dev_free(w.lo); dev_free(w.bias); dev_free(w.frag); w = PackedW(); }

void f5hip_dit_destroy(f5hip_dit* m) {
    if (!m) return;
    for (PackedW* w : {&m->time1, &m->time2, &m->adaln, &m->wx, &m->wct, &m->conv1, &m->conv2, &m->proj_out}) free_packed(*w);
    for (auto* v : {&m->wqkv, &m->wout, &m->wff1, &m->wff2, &m->wskip, &m->wqkv_c, &m->wout_c, &m->wff1_c, &m->wff2_c}) for (auto& w : *v) free_packed(w);
    for (auto* v : {&m->g_attn, &m->g_ff}) for (float* g : *v) dev_free(g);
    dev_free(m->g_out); dev_free(m->zeros);
    for (auto& b : m->tblk) {
        for (float* p : {b.dw_w, b.dw_b, b.ln_w, b.ln_b, b.gamma, b.beta}) dev_free(p);
        free_packed(b.pw1); free_packed(b.pw2);
    }
    for (float* p : {m->text_emb, m->text_pos, m->rope_cos, m->rope_sin}) dev_free(p);
    dev_free(m->ln_sync);
    if (m->ln_err) (void)hipHostFree(m->ln_err);
    dev_free(m->ws.ptr);
#ifdef F5HIP_EXPERIMENTS
    streamk_ws_free(m->sk);
#endif
    dev_free(m->meta);
    delete m->prof;
    m->up_meta.release(); m->up_time[0].release(); m->up_time[1].release();
    delete m;
}

int f5hip_dit_load_param(f5hip_dit* m, const char* name, const float* data, int64_t numel) {
    if (!m || !name || !data || numel <= 0) return fail(-1, "load_param: bad argument");
    if (m->finalized) return fail(-2, "load_param after finalize");
    m->host[name].assign(data, data + numel);
    return 0;
}

static const std::vector<float>* get_param(f5hip_dit* m, const std::string& name, int64_t numel) {
    auto it = m->host.find(name);
    if (it == m->host.end()) { set_error("missing parameter %s", name.c_str()); return nullptr; }
    if ((int64_t)it->second.size() != numel) {
        set_error("parameter %s has %lld elements, expected %lld", name.c_str(), (long long)it->second.size(), (long long)numel);
        return nullptr;
    }
    return &it->second;
}

#define GETP(var, name, numel) const std::vector<float>* var = get_param(m, name, numel); if (!var) return -3;

int f5hip_dit_finalize(f5hip_dit* m) {
    if (!m) return fail(-1, "null model");
    if (m->finalized) return 0;
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim, Td = c.text_dim, mel = c.mel_dim, F = c.ff_mult * D;
    const std::string T = "transformer.";
    // --- time MLP + all AdaLN linears (one [depth*6D + 2D, D] matrix: modulation depends on t only) ---
    {
        GETP(w0, T + "time_embed.time_mlp.0.weight", (int64_t)D * 256);
        GETP(b0, T + "time_embed.time_mlp.0.bias", D);
        GETP(w2, T + "time_embed.time_mlp.2.weight", (int64_t)D * D);
        GETP(b2, T + "time_embed.time_mlp.2.bias", D);
        if (pack_linear(m->time1, w0->data(), D, 256, 256, b0->data())) return -4;
        if (pack_linear(m->time2, w2->data(), D, D, D, b2->data())) return -4;
    }
    if (m->arch == 0) {
        std::vector<float> wa((size_t)m->n_adaln * D), ba(m->n_adaln);
        for (int l = 0; l < c.depth; l++) {
            std::string p = T + "transformer_blocks." + std::to_string(l) + ".attn_norm.linear.";
            GETP(w, p + "weight", (int64_t)6 * D * D);
            GETP(b, p + "bias", 6 * D);
            memcpy(&wa[(size_t)l * 6 * D * D], w->data(), sizeof(float) * 6 * D * D);
            memcpy(&ba[(size_t)l * 6 * D], b->data(), sizeof(float) * 6 * D);
        }
        GETP(wf, T + "norm_out.linear.weight", (int64_t)2 * D * D);
        GETP(bfin, T + "norm_out.linear.bias", 2 * D);
        memcpy(&wa[(size_t)c.depth * 6 * D * D], wf->data(), sizeof(float) * 2 * D * D);
        memcpy(&ba[(size_t)c.depth * 6 * D], bfin->data(), sizeof(float) * 2 * D);
        if (pack_linear(m->adaln, wa.data(), m->n_adaln, D, D, ba.data())) return -4;
    }
    if (m->arch == 2) {   // MMDiTBlock: attn_norm_c (AdaLayerNormZero, or _Final in the last block) and attn_norm_x (F/model/modules.py:593-594)
        std::vector<float> wa((size_t)m->n_adaln * D), ba(m->n_adaln);
        for (int l = 0; l < c.depth; l++) {
            const std::string p = T + "transformer_blocks." + std::to_string(l) + ".";
            const int nc = (l == c.depth - 1 ? 2 : 6) * D;
            GETP(wc, p + "attn_norm_c.linear.weight", (int64_t)nc * D); GETP(bc, p + "attn_norm_c.linear.bias", nc);
            GETP(wxm, p + "attn_norm_x.linear.weight", (int64_t)6 * D * D); GETP(bxm, p + "attn_norm_x.linear.bias", 6 * D);
            memcpy(&wa[(size_t)m->mod_c[l] * D], wc->data(), sizeof(float) * (size_t)nc * D);
            memcpy(&ba[m->mod_c[l]], bc->data(), sizeof(float) * nc);
            memcpy(&wa[(size_t)m->mod_x[l] * D], wxm->data(), sizeof(float) * (size_t)6 * D * D);
            memcpy(&ba[m->mod_x[l]], bxm->data(), sizeof(float) * 6 * D);
        }
        GETP(wf, T + "norm_out.linear.weight", (int64_t)2 * D * D);
        GETP(bfin, T + "norm_out.linear.bias", 2 * D);
        memcpy(&wa[(size_t)m->mod_final * D], wf->data(), sizeof(float) * 2 * D * D);
        memcpy(&ba[m->mod_final], bfin->data(), sizeof(float) * 2 * D);
        if (pack_linear(m->adaln, wa.data(), m->n_adaln, D, D, ba.data())) return -4;
    }
    // --- text embedding ---
    {
        GETP(e, T + "text_embed.text_embed.weight", (int64_t)(c.text_num_embeds + 1) * Td);
        if (upload_f32(&m->text_emb, e->data(), e->size())) return -4;
        // precompute_freqs_cis (F/model/modules.py:196-207): [cos(pos w_j) || sin(pos w_j)], fp32 angle
        std::vector<float> tab((size_t)4096 * Td);
        for (int pos = 0; pos < 4096; pos++)
            for (int j = 0; j < Td / 2; j++) {
                float w = 1.0f / powf(10000.0f, (float)(2 * j) / (float)Td);
                float ang = (float)pos * w;
                tab[(size_t)pos * Td + j] = (float)cos((double)ang);
                tab[(size_t)pos * Td + Td / 2 + j] = (float)sin((double)ang);
            }
        if (upload_f32(&m->text_pos, tab.data(), tab.size())) return -4;
        m->tblk.resize(c.conv_layers);
        for (int i = 0; i < c.conv_layers; i++) {
            std::string p = T + "text_embed.text_blocks." + std::to_string(i) + ".";
            TextBlock& b = m->tblk[i];
            GETP(dw, p + "dwconv.weight", (int64_t)Td * 7); GETP(db, p + "dwconv.bias", Td);
            GETP(lw, p + "norm.weight", Td); GETP(lb, p + "norm.bias", Td);
            GETP(w1, p + "pwconv1.weight", (int64_t)2 * Td * Td); GETP(b1, p + "pwconv1.bias", 2 * Td);
            GETP(gg, p + "grn.gamma", 2 * Td); GETP(gb, p + "grn.beta", 2 * Td);
            GETP(w2, p + "pwconv2.weight", (int64_t)2 * Td * Td); GETP(b2, p + "pwconv2.bias", Td);
            if (upload_f32(&b.dw_w, dw->data(), dw->size()) || upload_f32(&b.dw_b, db->data(), db->size()) ||
                upload_f32(&b.ln_w, lw->data(), lw->size()) || upload_f32(&b.ln_b, lb->data(), lb->size()) ||
                upload_f32(&b.gamma, gg->data(), gg->size()) || upload_f32(&b.beta, gb->data(), gb->size())) return -4;
            if (pack_linear(b.pw1, w1->data(), 2 * Td, Td, Td, b1->data())) return -4;
            if (pack_linear(b.pw2, w2->data(), Td, 2 * Td, 2 * Td, b2->data())) return -4;
        }
    }
    // --- input projection split by source: x part (changes every step) | cond + text part (step invariant) ---
    {
        const bool mm = m->arch == 2;   // MMDiT: AudioEmbedding.linear over cat(x, cond) only (mmdit.py:60-70)
        const int Kin = 2 * mel + (mm ? 0 : Td);
        GETP(w, T + (mm ? "audio_embed.linear.weight" : "input_embed.proj.weight"), (int64_t)D * Kin);
        GETP(b, T + (mm ? "audio_embed.linear.bias" : "input_embed.proj.bias"), D);
        const int Kct = 128 + m->td_pad;
        std::vector<float> wx((size_t)D * 128, 0.0f), wct((size_t)D * Kct, 0.0f);
        for (int n = 0; n < D; n++) {
            for (int k = 0; k < mel; k++) {
                wx[(size_t)n * 128 + k] = (*w)[(size_t)n * Kin + k];
                wct[(size_t)n * Kct + k] = (*w)[(size_t)n * Kin + mel + k];
            }
            for (int k = 0; k < (mm ? 0 : Td); k++) wct[(size_t)n * Kct + 128 + k] = (*w)[(size_t)n * Kin + 2 * mel + k];
        }
        if (pack_linear(m->wx, wx.data(), D, 128, 128, nullptr)) return -4;
        if (pack_linear(m->wct, wct.data(), D, Kct, Kct, b->data())) return -4;
    }
    // --- conv_pos_embed: grouped Conv1d(D, D, 31, groups 16) as 16 implicit GEMMs, each padded to 64 x (31 x 64) ---
    for (int which = 0; which < 2; which++) {
        const int gw = m->gw;
        std::string p = T + (m->arch == 2 ? "audio_embed" : "input_embed") + ".conv_pos_embed.conv1d." + std::to_string(which * 2) + ".";
        GETP(w, p + "weight", (int64_t)D * gw * 31);
        GETP(b, p + "bias", D);
        const int K = 31 * 64;
        std::vector<float> wp((size_t)16 * 64 * K, 0.0f), bp(16 * 64, 0.0f);
        for (int g = 0; g < 16; g++)
            for (int co = 0; co < gw; co++) {
                bp[g * 64 + co] = (*b)[g * gw + co];
                for (int ci = 0; ci < gw; ci++)
                    for (int tap = 0; tap < 31; tap++)
                        wp[((size_t)(g * 64 + co)) * K + tap * 64 + ci] = (*w)[((size_t)(g * gw + co) * gw + ci) * 31 + tap];
            }
        if (pack_linear(which ? m->conv2 : m->conv1, wp.data(), 16 * 64, K, K, bp.data())) return -4;
    }
    // --- transformer blocks ---
    m->wqkv.resize(c.depth); m->wout.resize(c.depth); m->wff1.resize(c.depth); m->wff2.resize(c.depth);
    if (m->arch == 1) { m->wskip.resize(c.depth); m->g_attn.assign(c.depth, nullptr); m->g_ff.assign(c.depth, nullptr); }
    if (m->arch == 2) { m->wqkv_c.resize(c.depth); m->wout_c.resize(c.depth); m->wff1_c.resize(c.depth); m->wff2_c.resize(c.depth); }
    for (int l = 0; l < c.depth; l++) {
        // DiT: transformer_blocks.{l}.attn.* / .ff.*  (F/model/modules.py:542-556);  UNetT: layers.{l}.{0 skip_proj, 1 attn_norm, 2 attn, 3 ff_norm, 4 ff}
        const std::string p = T + (m->arch != 1 ? "transformer_blocks." : "layers.") + std::to_string(l) + ".";
        const std::string pa = p + (m->arch != 1 ? "attn." : "2."), pf = p + (m->arch == 0 ? "ff." : (m->arch == 2 ? "ff_x." : "4."));
        std::vector<float> wq((size_t)3 * D * D), bq(3 * D);
        const char* nm[3] = {"to_q", "to_k", "to_v"};
        for (int i = 0; i < 3; i++) {
            GETP(w, pa + nm[i] + ".weight", (int64_t)D * D);
            GETP(b, pa + nm[i] + ".bias", D);
            memcpy(&wq[(size_t)i * D * D], w->data(), sizeof(float) * D * D);
            memcpy(&bq[(size_t)i * D], b->data(), sizeof(float) * D);
        }
        if (pack_linear(m->wqkv[l], wq.data(), 3 * D, D, D, bq.data(), 128, m->blk_f16) || pack_frag(m->wqkv[l])) return -4;   // (+ fragment order: W-direct gemm5)
        GETP(wo, pa + "to_out.0.weight", (int64_t)D * D); GETP(bo, pa + "to_out.0.bias", D);
        if (pack_linear(m->wout[l], wo->data(), D, D, D, bo->data(), 128, m->blk_f16)) return -4;
        GETP(w1, pf + "ff.0.0.weight", (int64_t)F * D); GETP(b1, pf + "ff.0.0.bias", F);
        if (pack_linear(m->wff1[l], w1->data(), F, D, D, b1->data(), 128, m->blk_f16) || pack_frag(m->wff1[l])) return -4;
        GETP(w2, pf + "ff.2.weight", (int64_t)D * F); GETP(b2, pf + "ff.2.bias", D);
        if (pack_linear(m->wff2[l], w2->data(), D, F, F, b2->data(), 128, m->blk_f16)) return -4;
        if (m->arch == 2) {   // the text stream's own projections (Attention(context_dim=...), F/model/modules.py:365-374) and feed-forward
            std::vector<float> wqc((size_t)3 * D * D), bqc(3 * D);
            const char* nmc[3] = {"to_q_c", "to_k_c", "to_v_c"};
            for (int i = 0; i < 3; i++) {
                GETP(w, pa + nmc[i] + ".weight", (int64_t)D * D);
                GETP(b, pa + nmc[i] + ".bias", D);
                memcpy(&wqc[(size_t)i * D * D], w->data(), sizeof(float) * D * D);
                memcpy(&bqc[(size_t)i * D], b->data(), sizeof(float) * D);
            }
            if (pack_linear(m->wqkv_c[l], wqc.data(), 3 * D, D, D, bqc.data(), 128, m->blk_f16) || pack_frag(m->wqkv_c[l])) return -4;
            if (l < c.depth - 1) {
                GETP(woc, pa + "to_out_c.weight", (int64_t)D * D); GETP(boc, pa + "to_out_c.bias", D);
                if (pack_linear(m->wout_c[l], woc->data(), D, D, D, boc->data(), 128, m->blk_f16)) return -4;
                GETP(w1c, p + "ff_c.ff.0.0.weight", (int64_t)F * D); GETP(b1c, p + "ff_c.ff.0.0.bias", F);
                if (pack_linear(m->wff1_c[l], w1c->data(), F, D, D, b1c->data(), 128, m->blk_f16) || pack_frag(m->wff1_c[l])) return -4;
                GETP(w2c, p + "ff_c.ff.2.weight", (int64_t)D * F); GETP(b2c, p + "ff_c.ff.2.bias", D);
                if (pack_linear(m->wff2_c[l], w2c->data(), D, F, F, b2c->data(), 128, m->blk_f16)) return -4;
            }
        }
        if (m->arch == 1) {
            GETP(ga, p + "1.g", D); GETP(gf, p + "3.g", D);
            if (upload_f32(&m->g_attn[l], ga->data(), D) || upload_f32(&m->g_ff[l], gf->data(), D)) return -4;
            if (l >= c.depth / 2) {
                GETP(ws, p + "0.weight", (int64_t)D * 2 * D);
                if (pack_linear(m->wskip[l], ws->data(), D, 2 * D, 2 * D, nullptr, 128, m->skip_f16)) return -4;
            }
        }
    }
    if (m->arch == 1) {
        GETP(go, T + "norm_out.g", D);
        if (upload_f32(&m->g_out, go->data(), D)) return -4;
    }
    {
        std::vector<float> z(std::max(D, 4096), 0.0f);
        if (upload_f32(&m->zeros, z.data(), z.size())) return -4;
    }
    {
        GETP(w, T + "proj_out.weight", (int64_t)mel * D); GETP(b, T + "proj_out.bias", mel);
        if (pack_linear(m->proj_out, w->data(), mel, D, D, b->data())) return -4;
    }
    // --- rotary tables (x-transformers 2.2.8 RotaryEmbedding): angle = pos * 10000^(-2i/64) in fp32 ---
    {
        // 4097 rows: UNetT puts the time token at position 0, so a 4096-frame sequence reaches position 4096 (unett.py:184-188)
        std::vector<float> rc((size_t)4097 * 32), rs((size_t)4097 * 32);
        for (int pos = 0; pos < 4097; pos++)
            for (int i = 0; i < 32; i++) {
                float inv = 1.0f / powf(10000.0f, (float)(2 * i) / 64.0f);
                float ang = (float)pos * inv;
                rc[(size_t)pos * 32 + i] = (float)cos((double)ang);
                rs[(size_t)pos * 32 + i] = (float)sin((double)ang);
            }
        if (upload_f32(&m->rope_cos, rc.data(), rc.size()) || upload_f32(&m->rope_sin, rs.data(), rs.size())) return -4;
    }
    m->host.clear();
    m->finalized = true;
    return 0;
}

// -------------------------------------------------------------------------------------------------
// workspace
// -------------------------------------------------------------------------------------------------
static int ensure_workspace(f5hip_dit* m, int rows_pad, int frames, int n_seq) {
    if (rows_pad <= m->cap_rows && frames <= m->cap_frames && n_seq <= m->cap_seq) return 0;
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim, Td = c.text_dim, F = c.ff_mult * D;
    const size_t R = (size_t)std::max(rows_pad, m->cap_rows), U = (size_t)std::max(frames, m->cap_frames),
                 S = (size_t)std::max(n_seq, m->cap_seq);
    dev_free(m->ws.ptr);
    Arena a;
    // pass 1 sizes, pass 2 pointers
    for (int pass = 0; pass < 2; pass++) {
        a.reset(pass ? (char*)m->ws.ptr : nullptr);
        m->h = a.f32(R * D); m->h0 = a.f32(R * D); m->ce = a.f32(R * D); m->pred = a.f32(R * 128);
        m->te = a.f32(R * Td); m->ty = a.f32(R * 2 * Td); m->gx = a.f32(S * 2 * Td);
        m->rope_row_cos = a.f32(R * 32); m->rope_row_sin = a.f32(R * 32);
        m->mod = a.f32((size_t)128 * m->n_adaln + 64); m->xstate = a.f32(U * c.mel_dim); m->xmid = a.f32(U * c.mel_dim); m->temb = a.f32((size_t)128 * D);
        m->hn = a.plane2(R * D + 256); m->c1 = a.plane2(R * D + 256); m->ao = a.plane2(R * D); m->ff = a.plane2(R * F);
        m->xs = a.plane2(R * 128); m->tn = a.plane2(R * Td); m->tg = a.plane2(R * 2 * Td); m->act = a.plane2(R * (128 + m->td_pad));
        m->sinp = a.plane2(128 * 256); m->t1 = a.plane2((size_t)128 * D); m->st = a.plane2((size_t)128 * D);
        m->skipbuf.resize(m->arch == 1 ? c.depth / 2 : 0);
        for (auto& sb : m->skipbuf) sb = a.plane2(R * 2 * D);   // [R][2 D]: the concatenated operand [x || skip] of the U-skip Linear, built in place
        // qk: +256 rows because the last 256-query tile of attn2 may read (never store) past the padded rows
        m->qk = a.bf16((R + 256) * 2 * D); m->vt = a.bf16((size_t)D * R);
        if (!pass) {
            if (hipMalloc(&m->ws.ptr, a.used()) != hipSuccess) { m->ws.ptr = nullptr; m->cap_rows = 0; return fail(-5, "hipMalloc workspace %zu bytes", a.used()); }
            if (hipMemset(m->ws.ptr, 0, a.used()) != hipSuccess) return fail(-5, "hipMemset workspace");
        }
    }
    m->cap_rows = (int)R; m->cap_frames = (int)U; m->cap_seq = (int)S;
    const int need = (int)(R * 8 + S * 15 + U * 3 + 64);
    if (need > m->meta_cap) {
        dev_free(m->meta);
        if (hipMalloc((void**)&m->meta, sizeof(int) * need) != hipSuccess) { m->meta = nullptr; m->meta_cap = 0; return fail(-5, "hipMalloc meta"); }
        m->meta_cap = need;
    }
    return 0;
}

struct SeqDesc { int len, kvlen, frame0 /* first frame in caller's packed arrays */, text_row, drop_audio, drop_text, branch /* 0 cond, 1 uncond */;
                 int c_len = 0 /* MMDiT: rows of the text stream (tokens incl. filler positions, as the reference's [b, nt] text tensor has them) */; };

// Lays the sequences out (each padded to a multiple of 128 rows), builds the per-row metadata and uploads it.
// UNetT: row 0 of every sequence is the time token (F/model/backbones/unett.py:184); frames follow at rows 1..len.
static int setup_sequences(f5hip_dit* m, const std::vector<SeqDesc>& seqs, int n_frames, const int32_t* text, int nt_max,
                           const uint8_t* frame_is_cond, hipStream_t st) {
    const int extra = m->arch == 1 ? 1 : 0;
    const bool mm = m->arch == 2;
    int rows = 0, rows_x = 0;
    for (auto& s : seqs) rows += ceil_to(s.len + extra, 128);
    rows_x = rows;
    if (mm) for (auto& s : seqs) {
        if (s.c_len <= 0 || s.c_len > 4096) return fail(-1, "MMDiT: a sequence needs 1..4096 text positions (got %d)", s.c_len);
        rows += ceil_to(s.c_len, 128);
    }
    if (ensure_workspace(m, rows, n_frames, (int)seqs.size())) return -5;
    const int R = rows, S = (int)seqs.size(), U = n_frames;
    std::vector<int> hbuf((size_t)R * 8 + S * 3 + U * 3 + (mm ? 12 * S : 0), 0);
    int* row_pos = &hbuf[0]; int* row_start = row_pos + R; int* row_end = row_start + R; int* row_seq = row_end + R;
    int* row_token = row_seq + R; int* row_frame = row_token + R; int* row_condframe = row_frame + R; int* row_keep = row_condframe + R;
    int* seq_row0 = row_keep + R; int* seq_len = seq_row0 + S; int* seq_kvlen = seq_len + S;
    int* urow_c = seq_kvlen + S; int* urow_u = urow_c + U; int* fic = urow_u + U;
    int r0 = 0;
    m->any_masked = false;
    for (int r = 0; r < R; r++) { row_seq[r] = -1; row_token[r] = -1; row_frame[r] = -1; row_condframe[r] = -1; }
    for (int u = 0; u < U; u++) { urow_c[u] = -1; urow_u[u] = -1; fic[u] = frame_is_cond ? frame_is_cond[u] : 0; }
    m->max_len = 0;
    for (int s = 0; s < S; s++) {
        const SeqDesc& q = seqs[s];
        seq_row0[s] = r0; seq_len[s] = q.len + extra; seq_kvlen[s] = q.kvlen + extra;
        m->max_len = std::max(m->max_len, q.len + extra);
        if (q.kvlen < q.len) m->any_masked = true;
        if (extra) { row_pos[r0] = 0; row_seq[r0] = s; row_keep[r0] = 1; }   // time token: start = end = 0 keeps it out of the convs
        for (int i = 0; i < q.len; i++) {
            const int r = r0 + extra + i;
            row_pos[r] = i + extra;                                   // rotary position (time token = 0)
            row_start[r] = r0 + extra; row_end[r] = r0 + extra + q.len; row_seq[r] = s;
            int tok = 0;
            if (!q.drop_text && i < nt_max) tok = text[(size_t)q.text_row * nt_max + i] + 1;   // -1 pad -> filler 0
            // nn.Embedding raises IndexError on an id outside the table; here it would be an out-of-bounds read on the GPU
            if (tok < 0 || tok > m->cfg.text_num_embeds) return fail(-1, "text token %d of sequence %d is outside the vocabulary (0..%d)", tok - 1, s, m->cfg.text_num_embeds - 1);
            row_token[r] = tok;
            row_frame[r] = q.frame0 + i;
            const bool is_c = frame_is_cond ? frame_is_cond[q.frame0 + i] != 0 : true;
            row_condframe[r] = (!q.drop_audio && is_c) ? q.frame0 + i : -1;
            row_keep[r] = i < q.kvlen ? 1 : 0;
            (q.branch ? urow_u : urow_c)[q.frame0 + i] = r;
        }
        r0 += ceil_to(q.len + extra, 128);
    }
    int* jm = fic + U;   // MMDiT joint attention: 6 arrays of 2 S pseudo-sequences (2 s: audio queries of sequence s, 2 s + 1: its text queries)
    if (mm) {
        int rc0 = rows_x;
        for (int s = 0; s < S; s++) {
            const SeqDesc& q = seqs[s];
            for (int i = 0; i < q.c_len; i++) {
                const int r = rc0 + i;
                int tok = 0;   // filler (the reference feeds text + 1 with -1 padding -> 0; all ids 0 when the text is dropped: mmdit.py:38-40)
                if (!q.drop_text && i < nt_max) tok = text[(size_t)q.text_row * nt_max + i] + 1;
                if (tok < 0 || tok > m->cfg.text_num_embeds) return fail(-1, "text token %d of sequence %d is outside the vocabulary (0..%d)", tok - 1, s, m->cfg.text_num_embeds - 1);
                row_pos[r] = i; row_seq[r] = s; row_token[r] = tok; row_keep[r] = 1; row_start[r] = rc0; row_end[r] = rc0 + q.c_len;
            }
            for (int qd = 0; qd < 2; qd++) {
                const int j = 2 * s + qd;
                jm[j] = qd ? rc0 : seq_row0[s];                 // query rows
                jm[2 * S + j] = qd ? q.c_len : q.len;
                jm[4 * S + j] = q.kvlen;                        // first key range: the audio rows, padding masked
                jm[6 * S + j] = seq_row0[s];
                jm[8 * S + j] = rc0;                            // second key range: the text rows, never masked (F/model/modules.py:508)
                jm[10 * S + j] = q.c_len;
            }
            m->max_len = std::max(m->max_len, q.c_len);
            rc0 += ceil_to(q.c_len, 128);
        }
    }
    if (const int r_ = m->up_meta.upload(m->meta, hbuf.data(), sizeof(int) * hbuf.size(), st)) return r_;   // (pinned staging: no host sync)
    int* d = m->meta;
    m->d_row_pos = d; m->d_row_start = d + R; m->d_row_end = d + 2 * R; m->d_row_seq = d + 3 * R; m->d_row_token = d + 4 * R;
    m->d_row_frame = d + 5 * R; m->d_row_condframe = d + 6 * R; m->d_row_keep = d + 7 * R;
    d += 8 * R;
    m->d_seq_row0 = d; m->d_seq_len = d + S; m->d_seq_kvlen = d + 2 * S; d += 3 * S;
    m->d_urow_c = d; m->d_urow_u = d + U; m->d_frame_is_cond = d + 2 * U;
    d += 3 * U;
    m->d_j_row0 = d; m->d_j_len = d + 2 * S; m->d_j_kvlen = d + 4 * S; m->d_j_kv_row0 = d + 6 * S; m->d_j_kv2_row0 = d + 8 * S; m->d_j_kv2_len = d + 10 * S;
    // rotary factors per row of this layout (one load in the QKV epilogues instead of row_pos -> table)
    hipLaunchKernelGGL(rope_rows_kernel, dim3((R * 32 + 255) / 256), dim3(256), 0, st, m->d_row_pos, m->rope_cos, m->rope_sin, R, 4097, m->rope_row_cos, m->rope_row_sin);
    if (hipGetLastError() != hipSuccess) return fail(-7, "rope_rows_kernel launch");
    m->M = rows_x; m->M_pad = rows_x; m->Mc = R - rows_x; m->Rtot = R; m->n_seq = S; m->n_frames = U;
    return 0;
}

// -------------------------------------------------------------------------------------------------
// launch helpers
// -------------------------------------------------------------------------------------------------
static GemmArgs gemm_base(const Plane2& A, int lda, const PackedW& W, int M) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A[0] = A.hi; a.A[1] = A.lo; a.lda = lda;
    a.W[0] = W.hi; a.W[1] = W.lo; a.Wf = W.frag;
    a.M = M; a.N = W.n; a.K = W.k_pad; a.ldw = W.ld;
    a.bias = W.bias;
    return a;
}

static int g_gemm_impl = -1;   // F5HIP_GEMM_IMPL: 0 = automatic; 1 = register-staged kernel only (gemm.h); 3 = gemm3 instead of gemm5 (A/B); 2 / 4 = experiments build only
static long long g_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // f5hip_get_counter: gemm5 launches with RB 11 / RB 8 / 1 x 4 consumer layout / gemm3 wide-tile launches / ... / gemm6 launches
static int g_gemm6_mode = -1;   // F5HIP_GEMM6: 0 = never, 1 = whenever the shape is legal, 256 / 176 = that tile height whenever legal, unset = automatic (batch-mode shapes)

// Kernel choice per GEMM (measured: profiles/r02_fillrate_microbench.txt, profiles/r01_gemm_microbench.txt, tools/gemm_microbench.py):
//   fp16 one-plane operands with K % 64 == 0 (the four transformer-block GEMMs of the DiT in mixed mode): gemm5, exact-fit tiles;
//   one 128 x 128 tile per CU or fewer, generic epilogue, bf16 / split-bf16 operands: gemm3 (warp-specialised LDS-DMA ring);
//   everything else (implicit-GEMM convolutions, QKV in split-bf16, many-tile shapes): gemm.h, two 4-wave workgroups per CU.
static int run_gemm_n(int nsplit, int mp, GemmArgs& a, const PackedW& W, int epi, bool conv, int bn, hipStream_t st, void* sk = nullptr) {
    hipError_t e;
    const int np = W.n_pad;
    if (mp % 128 || np % bn || a.K % 32) return fail(-7, "gemm: bad padded shape %d x %d x %d", mp, np, a.K);
    if (g_gemm_impl < 0) {
        const char* env = getenv("F5HIP_GEMM_IMPL");
        g_gemm_impl = env ? atoi(env) : 0;
    }
    (void)sk;
    prof_begin(PROF_GEMM, st);
    const long long tiles128 = (long long)(mp / 128) * (np / 128);
    const bool use3 = !conv && (g_gemm_impl == 3 || (g_gemm_impl == 0 && tiles128 <= 256 && epi != EPI_QKV));
#ifdef F5HIP_EXPERIMENTS
    const int nk32 = a.K >> 5;
    StreamKWs* skw = (StreamKWs*)sk;
    if (g_gemm_impl == 4 && skw && !conv && nsplit != 2 && np % 128 == 0 && nk32 >= 16 && tiles128 >= 64) {   // stream-K: 5-25 % slower at every C2 shape
        if (streamk_ws_init(*skw)) { prof_end(PROF_GEMM, st); return fail(-5, "stream-K workspace"); }
        if (nsplit == 3) e = epi == EPI_QKV ? launch_gemm4_t<3, EPI_QKV>(a, mp, np, *skw, st) : launch_gemm4_t<3, EPI_GENERIC>(a, mp, np, *skw, st);
        else e = epi == EPI_QKV ? launch_gemm4_t<1, EPI_QKV>(a, mp, np, *skw, st) : launch_gemm4_t<1, EPI_GENERIC>(a, mp, np, *skw, st);
    } else if (g_gemm_impl == 2 && !conv && nsplit != 3) {
        if (nsplit == 2) e = epi == EPI_QKV ? launch_gemm2_t<2, 128, 128, EPI_QKV>(a, mp, np, st) : launch_gemm2_t<2, 128, 128, EPI_GENERIC>(a, mp, np, st);
        else e = epi == EPI_QKV ? launch_gemm2_t<1, 128, 128, EPI_QKV>(a, mp, np, st) : launch_gemm2_t<1, 128, 128, EPI_GENERIC>(a, mp, np, st);
    } else
#endif
    if (nsplit == 3) {   // fp16 operands, one plane each
        if (conv) {   // implicit-GEMM convolution (BigVGAN in fp16 mode): register-staged kernel
            e = f5_launch_gemm_reg(3, bn, true, epi, a, mp, np, st);
            prof_end(PROF_GEMM, st);
            if (e != hipSuccess) return fail(-7, "gemm launch: %s", hipGetErrorString(e));
            return 0;
        }
        const Gemm5Choice c5 = gemm5_choose(a.M, np);
        if (g_gemm6_mode < 0) {
            const char* env6 = getenv("F5HIP_GEMM6");
            g_gemm6_mode = env6 ? atoi(env6) : 2;
        }
        // gemm6 (256 x 256 ping-pong tiles): the batch-mode shapes -- enough tiles to occupy the chip in their first round
        const bool legal6 = g_gemm_impl == 0 && a.K % 64 == 0 && np % 256 == 0 && (epi != EPI_QKV || a.D % 256 == 0);
        int rows6 = legal6 ? gemm6_choose_rows(a.M, np) : 0;
        if (legal6 && (g_gemm6_mode == 1 || g_gemm6_mode == 256 || g_gemm6_mode == 176)) rows6 = g_gemm6_mode == 1 ? (rows6 ? rows6 : 256) : g_gemm6_mode;   // (forced: tools)
        if (rows6 && g_gemm6_mode != 0) {
            e = f5_launch_gemm6(epi, rows6, a, np, st);
            g_counters[7]++;
        } else if ((g_gemm_impl == 0 || g_gemm_impl == 5) && a.K % 64 == 0 && c5.rb) {
            e = epi == EPI_QKV ? f5_launch_gemm5_qkv(a, c5.rb, c5.cb, np, st) : f5_launch_gemm5_generic(a, c5.rb, c5.cb, np, st);
            g_counters[c5.rb == 11 ? 0 : 1]++;
            if (c5.cb >= 8) g_counters[2]++;
        } else if (g_gemm_impl == 1) {
            e = f5_launch_gemm_reg(3, 128, false, epi, a, mp, np, st);
        } else {
            // gemm3 (round 1): 128 x 256 tile in batch mode (>= 1024 tiles of 128 x 128), 128 x 128 otherwise
            const bool wide = tiles128 >= 1024 && np % 256 == 0;
            if (wide) g_counters[3]++;
            e = f5_launch_gemm3(3, epi, wide ? 256 : 128, a, mp, np, st);
        }
    } else if (use3) {
        e = f5_launch_gemm3(nsplit, epi, 128, a, mp, np, st);
    } else {
        e = f5_launch_gemm_reg(nsplit, bn, conv, epi, a, mp, np, st);
    }
    prof_end(PROF_GEMM, st);
    if (e != hipSuccess) return fail(-7, "gemm launch: %s", hipGetErrorString(e));
    return 0;
}
static int run_gemm(f5hip_dit* m, GemmArgs& a, const PackedW& W, int epi, bool conv, int bn, hipStream_t st, int m_pad = -1) {
#ifdef F5HIP_EXPERIMENTS
    void* sk = &m->sk;
#else
    void* sk = nullptr;
#endif
    return run_gemm_n(W.f16 ? 3 : m->nsplit, m_pad > 0 ? m_pad : m->M_pad, a, W, epi, conv, bn, st, sk);
}

static int run_ln(const LnArgs& a, hipStream_t st) {
    const int nv = (a.D + 255) / 256;
    dim3 grid((a.M + 3) / 4), blk(256);
    prof_begin(PROF_LN, st);
    static const bool ln_xcd = !(getenv("F5HIP_LN_XCD") && atoi(getenv("F5HIP_LN_XCD")) == 0);
    switch (nv) {
        case 1: hipLaunchKernelGGL(ln_kernel<1>, grid, blk, 0, st, a); break;
        case 2: hipLaunchKernelGGL(ln_kernel<2>, grid, blk, 0, st, a); break;
        case 3: hipLaunchKernelGGL(ln_kernel<3>, grid, blk, 0, st, a); break;
        case 4:
            if (ln_xcd) hipLaunchKernelGGL((ln_kernel<4, 1>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((ln_kernel<4, 0>), grid, blk, 0, st, a);
            break;
        case 5: case 6: hipLaunchKernelGGL(ln_kernel<6>, grid, blk, 0, st, a); break;
        default: return fail(-7, "ln: D=%d unsupported", a.D);
    }
    prof_end(PROF_LN, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-7, "ln launch: %s", hipGetErrorString(e));
    return 0;
}

// ---- LayerNorm fused behind a residual GEMM (EXPERIMENT, -DF5HIP_EXPERIMENTS builds with F5HIP_LN_FUSE=1) ------------------------------
// Measured and not shipped (profiles/r02_ln_fusion.txt): bit-identical to the separate kernel, but the fused residual GEMM takes 31.2 us
// against 18.8 us + 6.1 us for the GEMM and the stand-alone LayerNorm.  The in-kernel chain is five dependent L2 round trips of 1.5-2 us
// (store acknowledgement -> arrival atomic -> poll -> row loads -> stores); a kernel boundary resolves the same dependency in ~2 us.
#ifdef F5HIP_EXPERIMENTS
// The slab barrier of the LNE kernels needs (a) every workgroup of the launch resident at once and (b) the 16 workgroups of a row slab on
// ONE XCD (their only coherence point is that XCD's L2).  (b) holds under gemm5's tile order if the hardware deals workgroup b to XCD
// b % 8: probed once per process with s_getreg XCC_ID.  F5HIP_LN_FUSE=0 turns the fusion off (A/B, and the fallback after a time-out).
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = (int)(id & 0xF);
    }
}
static int g_ln_fuse = -1;   // -1 not probed, 0 off, 1 on
static int g_num_cus = 0;
static void ln_fuse_probe() {
    if (g_ln_fuse >= 0) return;
    g_ln_fuse = 0;
    if (!getenv("F5HIP_LN_FUSE") || atoi(getenv("F5HIP_LN_FUSE")) != 1) return;   // opt-in
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
    g_num_cus = prop.multiProcessorCount;
    const int nb = 512;
    int* d = nullptr;
    std::vector<int> h(nb, -1);
    if (hipMalloc((void**)&d, nb * sizeof(int)) != hipSuccess) return;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(nb), dim3(64), 0, 0, d);
    const bool ok = hipGetLastError() == hipSuccess && hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return;
    for (int b = 0; b < nb; b++)
        if (h[b] != h[b & 7]) return;               // workgroups b and b % 8 on different XCDs: no fusion
    for (int x = 1; x < 8; x++)
        for (int y = 0; y < x; y++)
            if (h[x] == h[y]) return;               // fewer than 8 XCDs in the round robin
    g_ln_fuse = 1;
}

#endif

// Residual GEMM g (out projection / FF2: updates the stream h in place) followed by the LayerNorm `ln` over h: one LNE kernel when the
// launch is one resident wave of exact-fit tiles with 16 column tiles per slab and 8 or 16 slabs (a slab then sits on one XCD) AND the
// library is an experiments build with F5HIP_LN_FUSE=1; otherwise -- the shipped path -- two launches.
static int run_gemm_ln(f5hip_dit* m, GemmArgs& g, const PackedW& W, const LnArgs& ln, hipStream_t st) {
#ifdef F5HIP_EXPERIMENTS
    ln_fuse_probe();
    if (g_ln_fuse == 1 && *m->ln_err) {             // a barrier of an earlier call timed out: its results were wrong; never again
        g_ln_fuse = 0;
        g_counters[6]++;
        return fail(-9, "fused LayerNorm: a slab barrier timed out in an earlier launch (its output is invalid); the fusion is now off");
    }
    if (g_ln_fuse == 1 && W.f16 && !ln.dw_w && !ln.out_f32 && ln.D == 1024 && ln.ldx == 1024 && ln.x == g.out_f32 && g.ldo == 1024 && W.n == 1024 && g.K % 64 == 0 &&
        g_gemm_impl == 0) {
        const Gemm5Choice c5 = gemm5_choose(g.M, W.n_pad);
        const int bm = c5.rb * 16, bnn = c5.cb * 16;
        const int tiles_m = c5.rb ? (g.M + bm - 1) / bm : 0, tiles_n = c5.rb ? W.n_pad / bnn : 0;
        if (c5.rb && c5.cb == 4 && tiles_n == 16 && (tiles_m == 8 || tiles_m == 16) && tiles_m * tiles_n <= g_num_cus) {
            if (tiles_m != m->ln_slabs) {           // counters of slabs the earlier launches did not have would lag behind the epoch
                if (hipMemsetAsync(m->ln_sync, 0, 16 * sizeof(unsigned), st) != hipSuccess) return fail(-6, "fused LayerNorm: counter reset");
                m->ln_epoch = 0;
                m->ln_slabs = tiles_m;
            }
            g.ln = ln;
            g.ln_sync = m->ln_sync;
            g.ln_target = ++m->ln_epoch * 16u;
            g.ln_err = m->ln_err;
            prof_begin(PROF_GEMM, st);
            const hipError_t e = f5_launch_gemm5_generic_lne(g, c5.rb, c5.cb, W.n_pad, st);
            prof_end(PROF_GEMM, st);
            if (e != hipSuccess) return fail(-7, "gemm5 LNE launch: %s", hipGetErrorString(e));
            g_counters[c5.rb == 11 ? 0 : 1]++;
            g_counters[5]++;
            return 0;
        }
    }
#endif
    if (const int r = run_gemm(m, g, W, EPI_GENERIC, false, 64, st)) return r;
    return run_ln(ln, st);
}

// Diagnostics for tests: which GEMM path the launches since the last reset took ("gemm5_rb11", "gemm5_rb8", "gemm5_wide", "gemm3_wide", "gemm6"); name "reset" zeroes them.
extern "C" int f5hip_get_counter(const char* name, int64_t* value) {
    static const char* names[8] = {"gemm5_rb11", "gemm5_rb8", "gemm5_wide", "gemm3_wide", "conv5", "ln_fused", "ln_fuse_timeouts", "gemm6"};
    if (!name) return fail(-1, "get_counter: null name");
    if (!strcmp(name, "reset")) { for (auto& c : g_counters) c = 0; return 0; }
    for (int i = 0; i < 8; i++)
        if (!strcmp(name, names[i])) { if (value) *value = g_counters[i]; return 0; }
    return fail(-1, "unknown counter %s", name);
}

#define CK(x) do { int _r = (x); if (_r) return _r; } while (0)
#define CKL(name) do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return fail(-7, "%s launch: %s", name, hipGetErrorString(_e)); } while (0)

// -------------------------------------------------------------------------------------------------
// step-invariant precompute: text embedding for every sequence, cond/text part of the input projection
// -------------------------------------------------------------------------------------------------
static int precompute_text_and_ce(f5hip_dit* m, const float* cond_dev, hipStream_t st) {
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim, Td = c.text_dim, M = m->M, Kct = 128 + m->td_pad;
    prof_begin(PROF_OTHER, st);
    if (m->arch == 2) {
        // MMDiT TextEmbedding (mmdit.py:37-52): embedding + absolute position table for the rows of the text stream (rows [M, M + Mc) of
        // every per-row buffer); step invariant, copied into the stream at the start of each forward.  Padding rows stay zero.
        hipLaunchKernelGGL(text_gather_kernel, dim3(m->Mc), dim3(256), 0, st, m->text_emb, m->text_pos, D, m->Mc, m->d_row_token + M,
                           m->d_row_pos + M, 1, m->te + (size_t)M * D, D, 1023);
    } else {
        hipLaunchKernelGGL(text_gather_kernel, dim3(M), dim3(256), 0, st, m->text_emb, m->text_pos, Td, M, m->d_row_token,
                           m->d_row_pos, c.conv_layers > 0 ? 1 : 0, m->te, Td, 4095);
    }
    CKL("text_gather");
    // audio-cond columns of the step-invariant operand (zero rows for dropped cond / non-cond frames / padding)
    hipLaunchKernelGGL(split_rows_kernel, dim3(M), dim3(256), 0, st, cond_dev, c.mel_dim, c.mel_dim, M, m->d_row_condframe,
                       m->act.hi, m->act.lo, Kct, 0);
    CKL("split cond");
    prof_end(PROF_OTHER, st);
    for (int i = 0; i < c.conv_layers; i++) {
        TextBlock& b = m->tblk[i];
        LnArgs ln; memset(&ln, 0, sizeof(ln));
        ln.x = m->te; ln.ldx = Td; ln.M = M; ln.D = Td; ln.scale = b.ln_w; ln.shift = b.ln_b; ln.gain_off = 0.0f; ln.eps = 1e-6f;
        ln.dw_w = b.dw_w; ln.dw_b = b.dw_b; ln.row_seq_start = m->d_row_start; ln.row_seq_end = m->d_row_end;
        ln.out_hi = m->tn.hi; ln.out_lo = m->tn.lo; ln.ldo = Td;
        CK(run_ln(ln, st));
        GemmArgs g1 = gemm_base(m->tn, Td, b.pw1, M);
        g1.act = ACT_GELU_ERF; g1.out_f32 = m->ty; g1.ldo = 2 * Td;
        CK(run_gemm(m, g1, b.pw1, EPI_GENERIC, false, 128, st));
        prof_begin(PROF_OTHER, st);
        hipLaunchKernelGGL(grn_stats_kernel, dim3((2 * Td + 255) / 256, m->n_seq), dim3(256), 0, st, m->ty, 2 * Td, 2 * Td,
                           m->d_seq_row0, m->d_seq_len, m->gx);
        CKL("grn_stats");
        hipLaunchKernelGGL(grn_apply_kernel, dim3((M + 3) / 4), dim3(256), 0, st, m->ty, 2 * Td, 2 * Td, M, m->d_row_seq, m->gx,
                           b.gamma, b.beta, m->tg.hi, m->tg.lo, 2 * Td);
        CKL("grn_apply");
        prof_end(PROF_OTHER, st);
        GemmArgs g2 = gemm_base(m->tg, 2 * Td, b.pw2, M);
        g2.res = m->te; g2.ldres = Td; g2.out_f32 = m->te; g2.ldo = Td;
        if (i == c.conv_layers - 1) { g2.out_hi = m->act.hi + 128; g2.out_lo = m->act.lo + 128; g2.ldob = Kct; }
        CK(run_gemm(m, g2, b.pw2, EPI_GENERIC, false, 128, st));
    }
    if (c.conv_layers == 0 && m->arch != 2) {   // (MMDiT: the text is not an input of the audio projection)
        prof_begin(PROF_OTHER, st);
        hipLaunchKernelGGL(split_rows_kernel, dim3(M), dim3(256), 0, st, m->te, Td, Td, M, (const int*)nullptr, m->act.hi,
                           m->act.lo, Kct, 128);
        CKL("split text");
        prof_end(PROF_OTHER, st);
    }
    GemmArgs g = gemm_base(m->act, Kct, m->wct, M);
    g.out_f32 = m->ce; g.ldo = D;
    CK(run_gemm(m, g, m->wct, EPI_GENERIC, false, 128, st));
    return 0;
}

// time embedding + every AdaLN modulation vector for all steps at once (they depend on t only)
static int precompute_time(f5hip_dit* m, const float* t_host, int n_t, hipStream_t st) {
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim;
    if (n_t > 128) return fail(-8, "at most 128 time points per call (got %d)", n_t);
    // SinusPositionEmbedding (F/model/modules.py:154-161) on the host: the table is n_t x 256
    std::vector<uint16_t> hi((size_t)128 * 256, 0), lo((size_t)128 * 256, 0);
    const float emb = logf(10000.0f) / (float)(128 - 1);
    for (int i = 0; i < n_t; i++)
        for (int k = 0; k < 128; k++) {
            const float f = expf((float)k * -emb);
            const float e = 1000.0f * t_host[i] * f;
            const float sv = (float)sin((double)e), cv = (float)cos((double)e);
            host_split_bf16(sv, hi[(size_t)i * 256 + k], lo[(size_t)i * 256 + k]);
            host_split_bf16(cv, hi[(size_t)i * 256 + 128 + k], lo[(size_t)i * 256 + 128 + k]);
        }
    CK(m->up_time[0].upload(m->sinp.hi, hi.data(), hi.size() * 2, st));
    CK(m->up_time[1].upload(m->sinp.lo, lo.data(), lo.size() * 2, st));
    GemmArgs g1 = gemm_base(m->sinp, 256, m->time1, n_t);
    g1.act = ACT_SILU; g1.out_hi = m->t1.hi; g1.out_lo = m->t1.lo; g1.ldob = D;
    int r = run_gemm(m, g1, m->time1, EPI_GENERIC, false, 128, st, 128);
    GemmArgs g2 = gemm_base(m->t1, D, m->time2, n_t);
    if (m->arch == 1) {
        g2.out_f32 = m->temb; g2.ldo = D;   // UNetT: the raw time embedding is prepended as a token (unett.py:184)
        if (!r) r = run_gemm(m, g2, m->time2, EPI_GENERIC, false, 128, st, 128);
        return r;
    }
    g2.act = ACT_SILU; g2.out_hi = m->st.hi; g2.out_lo = m->st.lo; g2.ldob = D;   // silu(t_emb): the only form AdaLN consumes
    if (!r) r = run_gemm(m, g2, m->time2, EPI_GENERIC, false, 128, st, 128);
    GemmArgs g3 = gemm_base(m->st, D, m->adaln, n_t);
    g3.out_f32 = m->mod; g3.ldo = m->n_adaln;
    if (!r) r = run_gemm(m, g3, m->adaln, EPI_GENERIC, false, 128, st, 128);
    return r;
}

// rotary operands of a QKV launch over rows row_off ..: the per-row tables of the current layout (one load per row in the epilogue);
// F5HIP_ROPE_ROWS=0: positions + the [pos][32] tables (two dependent loads; A/B)
static void set_rope(GemmArgs& q, const f5hip_dit* m, int row_off) {
    static const bool per_row = !(getenv("F5HIP_ROPE_ROWS") && atoi(getenv("F5HIP_ROPE_ROWS")) == 0);
    if (per_row) { q.row_pos = nullptr; q.rope_cos = m->rope_row_cos + (size_t)row_off * 32; q.rope_sin = m->rope_row_sin + (size_t)row_off * 32; }
    else { q.row_pos = m->d_row_pos + row_off; q.rope_cos = m->rope_cos; q.rope_sin = m->rope_sin; }
}

static int launch_attention(f5hip_dit* m, hipStream_t st) {
    const f5hip_dit_config& c = m->cfg;
    AttnArgs at; memset(&at, 0, sizeof(at));
    at.qk = m->qk; at.vt = m->vt; at.D = c.dim; at.ldvt = m->Rtot; at.seq_row0 = m->d_seq_row0; at.seq_len = m->d_seq_len;
    at.seq_kvlen = m->d_seq_kvlen; at.out_hi = m->ao.hi; at.out_lo = m->nsplit == 2 ? m->ao.lo : nullptr; at.f16_out = m->blk_f16 ? 1 : 0;
    at.shape_invariant = m->attn_invariant;
    int n_att = m->n_seq;
    if (m->arch == 2) {   // joint attention: audio and text queries of a sequence over its audio keys followed by its text keys
        at.seq_row0 = m->d_j_row0; at.seq_len = m->d_j_len; at.seq_kvlen = m->d_j_kvlen;
        at.seq_kv_row0 = m->d_j_kv_row0; at.seq_kv2_row0 = m->d_j_kv2_row0; at.seq_kv2_len = m->d_j_kv2_len;
        n_att = 2 * m->n_seq;
    }
    prof_begin(PROF_ATTN, st);
#ifdef F5HIP_EXPERIMENTS
    static int attn_impl = -1;
    if (attn_impl < 0) { const char* env = getenv("F5HIP_ATTN_IMPL"); attn_impl = env ? atoi(env) : 3; }
    if (attn_impl == 1) hipLaunchKernelGGL(attn_fwd_kernel, dim3((m->max_len + 127) / 128, c.heads, m->n_seq), dim3(256), 0, st, at);
    else if (attn_impl == 2) hipLaunchKernelGGL(attn2_fwd_kernel, dim3((m->max_len + 255) / 256, c.heads, m->n_seq), dim3(512), 0, st, at);
    else
#endif
    {
        static const int attn_sel = getenv("F5HIP_ATTN") ? atoi(getenv("F5HIP_ATTN")) : 3;   // 4 = experiments/attn4.h (A/B in -DF5HIP_EXPERIMENTS builds)
        const hipError_t e = attn_sel == 4 && m->arch != 2 ? f5_launch_attn4(at, m->max_len, c.heads, n_att, st) : f5_launch_attn3(at, m->max_len, c.heads, n_att, st);
        if (e != hipSuccess) { prof_end(PROF_ATTN, st); return fail(-7, "attention launch: %s", hipGetErrorString(e)); }
    }
    prof_end(PROF_ATTN, st);
    CKL("attention");
    return 0;
}

// UNetT (E2-TTS) layers, F/model/backbones/unett.py:184-219: time token at row 0 of every sequence, pre-norm blocks
//   x = attn(RMSNorm(x)) + x;  x = ff(RMSNorm(x)) + x,  U-skips: layer l >= depth/2 first does x = W_skip [x || skip(depth-1-l)].
static int forward_unett_layers(f5hip_dit* m, int ti, int n_blocks, hipStream_t st) {
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim, F = c.ff_mult * D, M = m->M;
    prof_begin(PROF_OTHER, st);
    hipLaunchKernelGGL(set_time_token_kernel, dim3(m->n_seq), dim3(256), 0, st, m->h, D, m->d_seq_row0, m->temb + (size_t)ti * D);
    prof_end(PROF_OTHER, st);
    CKL("set_time_token");
    const int nb = n_blocks < 0 ? c.depth : n_blocks;
    LnArgs ln; memset(&ln, 0, sizeof(ln));
    ln.x = m->h; ln.ldx = D; ln.M = M; ln.D = D; ln.shift = m->zeros; ln.gain_off = 0.0f; ln.eps = 0.0f; ln.rms = 1;
    ln.out_hi = m->hn.hi; ln.out_lo = m->hn.lo; ln.ldo = D;
    for (int l = 0; l < nb; l++) {
        if (l < c.depth / 2) {
            prof_begin(PROF_OTHER, st);
            // the skip is saved where its consumer wants it: columns D .. 2 D - 1 of that layer's [x || skip] operand (round 3: it used to go to a
            // [M][D] buffer and was copied behind x with hipMemcpy2DAsync at the consumer, ~55 % of the "other" kernel class at C5)
            if (m->skip_f16) hipLaunchKernelGGL(cast_rows_f16_kernel, dim3(M), dim3(256), 0, st, m->h, D, D, M, m->skipbuf[l].hi, 2 * D, D);
            else hipLaunchKernelGGL(split_rows_kernel, dim3(M), dim3(256), 0, st, m->h, D, D, M, (const int*)nullptr, m->skipbuf[l].hi, m->skipbuf[l].lo, 2 * D, D);
            prof_end(PROF_OTHER, st);
            CKL("skip save");
        } else {
            const Plane2& sk = m->skipbuf[c.depth - 1 - l];
            prof_begin(PROF_OTHER, st);
            if (m->skip_f16) hipLaunchKernelGGL(cast_rows_f16_kernel, dim3(M), dim3(256), 0, st, m->h, D, D, M, sk.hi, 2 * D, 0);
            else hipLaunchKernelGGL(split_rows_kernel, dim3(M), dim3(256), 0, st, m->h, D, D, M, (const int*)nullptr, sk.hi, sk.lo, 2 * D, 0);
            prof_end(PROF_OTHER, st);
            CKL("skip concat x");
            GemmArgs sp = gemm_base(sk, 2 * D, m->wskip[l], M);
            sp.bias = nullptr; sp.out_f32 = m->h; sp.ldo = D;
            CK(run_gemm(m, sp, m->wskip[l], EPI_GENERIC, false, 64, st));
        }
        ln.scale = m->g_attn[l];
        ln.f16_out = m->blk_f16 ? 1 : 0;   // block norms feed the fp16 block GEMMs in mixed mode; the final norm (proj_out) stays split bf16
        GemmArgs q = gemm_base(m->hn, D, m->wqkv[l], M);
        q.D = D; set_rope(q, m, 0); q.qk = m->qk; q.vt = m->vt; q.ldvt = m->Rtot;
        CK(run_ln(ln, st));
        CK(run_gemm(m, q, m->wqkv[l], EPI_QKV, false, 128, st));
        CK(launch_attention(m, st));
        GemmArgs o = gemm_base(m->ao, D, m->wout[l], M);
        o.res = m->h; o.ldres = D; o.out_f32 = m->h; o.ldo = D;
        o.row_keep = m->any_masked ? m->d_row_keep : nullptr;
        ln.scale = m->g_ff[l];
        CK(run_gemm_ln(m, o, m->wout[l], ln, st));     // out projection + the feed-forward norm behind it
        GemmArgs f1 = gemm_base(m->hn, D, m->wff1[l], M);
        f1.act = ACT_GELU_TANH; f1.out_hi = m->ff.hi; f1.out_lo = m->ff.lo; f1.ldob = F; f1.f16_out = m->blk_f16 ? 1 : 0;
        CK(run_gemm(m, f1, m->wff1[l], EPI_GENERIC, false, 128, st));
        GemmArgs f2 = gemm_base(m->ff, F, m->wff2[l], M);
        f2.res = m->h; f2.ldres = D; f2.out_f32 = m->h; f2.ldo = D;
        CK(run_gemm(m, f2, m->wff2[l], EPI_GENERIC, false, 64, st));
    }
    if (n_blocks >= 0) return 0;
    ln.scale = m->g_out;
    ln.f16_out = 0;
    CK(run_ln(ln, st));
    GemmArgs po = gemm_base(m->hn, D, m->proj_out, M);
    po.out_f32 = m->pred; po.ldo = 128;
    CK(run_gemm(m, po, m->proj_out, EPI_GENERIC, false, 128, st));
    return 0;
}

// Diagnostics (F5HIP_DUMP_QKV = 100 * layer + step): checksums of one QKV projection by column block and a hash of every other workspace
// buffer, printed to stderr.  This is how the 1-ulp rotary difference between tile widths was found.
static void debug_dump_qkv(f5hip_dit* m, hipStream_t st) {
    const int D = m->cfg.dim, F = m->cfg.ff_mult * D, M = m->M;
   // F5HIP_DUMP_QKV = 100 * layer + step   // diagnostics: checksums of the first QKV projection by column block
    (void)hipStreamSynchronize(st);
            std::vector<unsigned short> hq((size_t)M * 2 * D), hv((size_t)D * m->M_pad);
            (void)hipMemcpy(hq.data(), m->qk, hq.size() * 2, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hv.data(), m->vt, hv.size() * 2, hipMemcpyDeviceToHost);
            auto bf = [](unsigned short u) { unsigned v = (unsigned)u << 16; float f; memcpy(&f, &v, 4); return (double)f; };
            const int edges[6] = {0, 64, 128, 256, 512, D};
            for (int part = 0; part < 2; part++)
                for (int e = 0; e < 5; e++) {
                    double sum = 0, asum = 0;
                    for (int r = 0; r < M; r++)
                        for (int c = edges[e]; c < edges[e + 1]; c++) { const double x = bf(hq[(size_t)r * 2 * D + part * D + c]); sum += x; asum += fabs(x); }
                    fprintf(stderr, "[dump_qkv] %s cols [%d,%d): sum %.6f abs %.6f\n", part ? "K" : "Q", edges[e], edges[e + 1], sum, asum);
                }
            for (int e = 0; e < 5; e++) {
                double sum = 0, asum = 0;
                for (int c = edges[e]; c < edges[e + 1]; c++)
                    for (int r = 0; r < M; r++) { const double x = bf(hv[(size_t)c * m->M_pad + r]); sum += x; asum += fabs(x); }
                fprintf(stderr, "[dump_qkv] V rows [%d,%d): sum %.6f abs %.6f\n", edges[e], edges[e + 1], sum, asum);
            }
            // side effects: word checksums of every other workspace buffer (an out-of-bounds store of the projection would show here)
            struct { const char* name; const void* p; size_t bytes; } bufs[] = {
                {"h", m->h, (size_t)M * D * 4}, {"h0", m->h0, (size_t)M * D * 4}, {"ce", m->ce, (size_t)M * D * 4}, {"pred", m->pred, (size_t)M * 128 * 4},
                {"mod", m->mod, (size_t)128 * m->n_adaln * 4}, {"hn.hi", m->hn.hi, (size_t)M * D * 2}, {"hn.lo", m->hn.lo, (size_t)M * D * 2},
                {"c1.hi", m->c1.hi, (size_t)M * D * 2}, {"ao.hi", m->ao.hi, (size_t)M * D * 2}, {"ao.lo", m->ao.lo, (size_t)M * D * 2},
                {"ff.hi", m->ff.hi, (size_t)M * F * 2}, {"ff.lo", m->ff.lo, (size_t)M * F * 2}, {"xs.hi", m->xs.hi, (size_t)M * 128 * 2},
                {"qk slack rows", m->qk + (size_t)M * 2 * D, (size_t)256 * 2 * D * 2}};
            for (auto& b : bufs) {
                std::vector<unsigned> w(b.bytes / 4);
                (void)hipMemcpy(w.data(), b.p, b.bytes, hipMemcpyDeviceToHost);
                unsigned long long acc = 0;
                for (unsigned x : w) acc = acc * 1000003ull + x;
                fprintf(stderr, "[dump_qkv] buffer %-14s hash %016llx\n", b.name, acc);
            }
        }

// One DiT evaluation at time index ti for all laid-out sequences.  xs (split bf16 of x) must be current.
// n_blocks < 0: full network, result in m->pred [M][128];  else stops after n_blocks blocks, result in m->h.
// MMDiT blocks (F/model/modules.py:614-642, JointAttnProcessor :460-536): two residual streams -- the audio rows [0, M) in m->h and the
// text rows [M, M + Mc) behind them in the same buffers -- with their own modulation, QKV, output and feed-forward weights and ONE
// joint attention per block over [audio keys ; text keys] (rotary on head 0 of each stream with its own positions).  The last block is
// context-pre-only: the text stream is only normalised and projected to q / k / v, then dropped.
static int forward_mmdit_layers(f5hip_dit* m, int ti, int n_blocks, hipStream_t st) {
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim, F = c.ff_mult * D, M = m->M, Mc = m->Mc;
    const float* mod = m->mod + (size_t)ti * m->n_adaln;
    const size_t ro = (size_t)M;                                   // first text row
    // the text stream starts every forward from its step-invariant embedding
    if (hipMemcpyAsync(m->h + ro * D, m->te + ro * D, sizeof(float) * (size_t)Mc * D, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(-6, "MMDiT: text stream copy");
    const Plane2 hn_c{m->hn.hi + ro * D, m->hn.lo + ro * D}, ao_c{m->ao.hi + ro * D, m->ao.lo + ro * D}, ff_c{m->ff.hi + ro * F, m->ff.lo + ro * F};
    auto ln_for = [&](bool text, const float* shift, const float* scale, bool f16) {
        LnArgs ln; memset(&ln, 0, sizeof(ln));
        ln.x = m->h + (text ? ro * D : 0); ln.ldx = D; ln.M = text ? Mc : M; ln.D = D; ln.shift = shift; ln.scale = scale; ln.gain_off = 1.0f; ln.eps = 1e-6f;
        ln.out_hi = text ? hn_c.hi : m->hn.hi; ln.out_lo = text ? hn_c.lo : m->hn.lo; ln.ldo = D; ln.f16_out = f16 ? 1 : 0;
        return ln;
    };
    const int nb = n_blocks < 0 ? c.depth : n_blocks;
    for (int l = 0; l < nb; l++) {
        const bool last = l == c.depth - 1;
        const float* mc = mod + m->mod_c[l];                       // text: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp; last block: scale, shift
        const float* mx = mod + m->mod_x[l];
        CK(run_ln(last ? ln_for(true, mc + D, mc, m->blk_f16) : ln_for(true, mc, mc + D, m->blk_f16), st));
        CK(run_ln(ln_for(false, mx, mx + D, m->blk_f16), st));
        GemmArgs q = gemm_base(m->hn, D, m->wqkv[l], M);
        q.D = D; set_rope(q, m, 0); q.qk = m->qk; q.vt = m->vt; q.ldvt = m->Rtot;
        CK(run_gemm(m, q, m->wqkv[l], EPI_QKV, false, 128, st, M));
        GemmArgs qc = gemm_base(hn_c, D, m->wqkv_c[l], Mc);
        qc.D = D; set_rope(qc, m, ro);
        qc.qk = m->qk + ro * 2 * D; qc.vt = m->vt + ro; qc.ldvt = m->Rtot;
        CK(run_gemm(m, qc, m->wqkv_c[l], EPI_QKV, false, 128, st, Mc));
        CK(launch_attention(m, st));
        GemmArgs o = gemm_base(m->ao, D, m->wout[l], M);
        o.mul = mx + 2 * D; o.res = m->h; o.ldres = D; o.out_f32 = m->h; o.ldo = D;
        o.row_keep = m->any_masked ? m->d_row_keep : nullptr;
        CK(run_gemm(m, o, m->wout[l], EPI_GENERIC, false, 64, st, M));
        if (!last) {
            GemmArgs oc = gemm_base(ao_c, D, m->wout_c[l], Mc);
            oc.mul = mc + 2 * D; oc.res = m->h + ro * D; oc.ldres = D; oc.out_f32 = m->h + ro * D; oc.ldo = D;
            CK(run_gemm(m, oc, m->wout_c[l], EPI_GENERIC, false, 64, st, Mc));
            CK(run_ln(ln_for(true, mc + 3 * D, mc + 4 * D, m->blk_f16), st));
            GemmArgs f1c = gemm_base(hn_c, D, m->wff1_c[l], Mc);
            f1c.act = ACT_GELU_TANH; f1c.out_hi = ff_c.hi; f1c.out_lo = ff_c.lo; f1c.ldob = F; f1c.f16_out = m->blk_f16 ? 1 : 0;
            CK(run_gemm(m, f1c, m->wff1_c[l], EPI_GENERIC, false, 128, st, Mc));
            GemmArgs f2c = gemm_base(ff_c, F, m->wff2_c[l], Mc);
            f2c.mul = mc + 5 * D; f2c.res = m->h + ro * D; f2c.ldres = D; f2c.out_f32 = m->h + ro * D; f2c.ldo = D;
            CK(run_gemm(m, f2c, m->wff2_c[l], EPI_GENERIC, false, 64, st, Mc));
        }
        CK(run_ln(ln_for(false, mx + 3 * D, mx + 4 * D, m->blk_f16), st));
        GemmArgs f1 = gemm_base(m->hn, D, m->wff1[l], M);
        f1.act = ACT_GELU_TANH; f1.out_hi = m->ff.hi; f1.out_lo = m->ff.lo; f1.ldob = F; f1.f16_out = m->blk_f16 ? 1 : 0;
        CK(run_gemm(m, f1, m->wff1[l], EPI_GENERIC, false, 128, st, M));
        GemmArgs f2 = gemm_base(m->ff, F, m->wff2[l], M);
        f2.mul = mx + 5 * D; f2.res = m->h; f2.ldres = D; f2.out_f32 = m->h; f2.ldo = D;
        CK(run_gemm(m, f2, m->wff2[l], EPI_GENERIC, false, 64, st, M));
    }
    if (n_blocks >= 0) return 0;
    const float* mf = mod + m->mod_final;                          // (scale, shift): F/model/modules.py:308
    CK(run_ln(ln_for(false, mf + D, mf, false), st));
    GemmArgs po = gemm_base(m->hn, D, m->proj_out, M);
    po.out_f32 = m->pred; po.ldo = 128;
    CK(run_gemm(m, po, m->proj_out, EPI_GENERIC, false, 128, st, M));
    return 0;
}

// One grouped convolution of ConvPositionEmbedding: the sliding-window kernel (conv5.h, 128-row tiles, one column tile per group) for the
// DiT and MMDiT layouts, the implicit GEMM of gemm.h otherwise (UNetT: the time-token row at the head of every sequence has an empty
// window of its own, which a per-tile bound cannot express) or with F5HIP_CONV5=0.
static int run_pos_conv(f5hip_dit* m, GemmArgs& g, const PackedW& W, hipStream_t st) {
    static const int use_conv5 = getenv("F5HIP_CONV5") ? atoi(getenv("F5HIP_CONV5")) : 1;
    if (use_conv5 && m->arch != 1 && m->nsplit == 2 && !W.f16) {
        prof_begin(PROF_GEMM, st);
        const hipError_t e = f5_launch_conv5(2, g, W.n_pad, st);
        prof_end(PROF_GEMM, st);
        if (e == hipSuccess) { g_counters[4]++; return 0; }
        if (e != hipErrorInvalidValue) return fail(-7, "conv5 launch: %s", hipGetErrorString(e));
    }
    return run_gemm(m, g, W, EPI_GENERIC, true, 64, st);
}

static int forward_step(f5hip_dit* m, int ti, int n_blocks, hipStream_t st) {
    const f5hip_dit_config& c = m->cfg;
    const int D = c.dim, F = c.ff_mult * D, M = m->M;
    const float* mod = m->mod + (size_t)ti * m->n_adaln;
    // input projection: x part + precomputed cond/text part
    GemmArgs gi = gemm_base(m->xs, 128, m->wx, M);
    gi.bias = nullptr; gi.res = m->ce; gi.ldres = D; gi.out_f32 = m->h0; gi.ldo = D;
    gi.out_hi = m->hn.hi; gi.out_lo = m->hn.lo; gi.ldob = D;
    CK(run_gemm(m, gi, m->wx, EPI_GENERIC, false, 128, st));
    // conv_pos_embed: Mish(GConv(Mish(GConv(h0)))) + h0   (F/model/modules.py:171-176, F/model/backbones/dit.py:86)
    GemmArgs c1 = gemm_base(m->hn, D, m->conv1, M);
    c1.conv_kpt = 2; c1.conv_center = 15; c1.conv_group_cols = m->gw; c1.row_seq_start = m->d_row_start; c1.row_seq_end = m->d_row_end;
    c1.group_w = m->gw; c1.N = 16 * 64;
    c1.act = ACT_MISH; c1.out_hi = m->c1.hi; c1.out_lo = m->c1.lo; c1.ldob = D;
    CK(run_pos_conv(m, c1, m->conv1, st));
    GemmArgs c2 = gemm_base(m->c1, D, m->conv2, M);
    c2.conv_kpt = 2; c2.conv_center = 15; c2.conv_group_cols = m->gw; c2.row_seq_start = m->d_row_start; c2.row_seq_end = m->d_row_end;
    c2.group_w = m->gw; c2.N = 16 * 64;
    c2.act = ACT_MISH; c2.res = m->h0; c2.ldres = D; c2.out_f32 = m->h; c2.ldo = D;
    CK(run_pos_conv(m, c2, m->conv2, st));

    const int nb = n_blocks < 0 ? c.depth : n_blocks;
    if (m->arch == 1) return forward_unett_layers(m, ti, n_blocks, st);
    if (m->arch == 2) return forward_mmdit_layers(m, ti, n_blocks, st);
    // Every LayerNorm but the first is launched together with the residual GEMM in front of it (run_gemm_ln: one kernel where the launch
    // is exact-fit): the out projection carries norm 2 of its block, FF2 carries norm 1 of the next block or the final norm.
    const float* mf = mod + (size_t)c.depth * 6 * D;   // final (scale, shift): F/model/modules.py:308
    auto block_ln = [&](const float* shift, const float* scale) {
        LnArgs ln; memset(&ln, 0, sizeof(ln));
        ln.x = m->h; ln.ldx = D; ln.M = M; ln.D = D; ln.shift = shift; ln.scale = scale; ln.gain_off = 1.0f; ln.eps = 1e-6f;
        ln.out_hi = m->hn.hi; ln.out_lo = m->hn.lo; ln.ldo = D; ln.f16_out = m->blk_f16 ? 1 : 0;
        return ln;
    };
    if (nb > 0) CK(run_ln(block_ln(mod, mod + D), st));
    for (int l = 0; l < nb; l++) {
        const float* ml = mod + (size_t)l * 6 * D;   // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        GemmArgs q = gemm_base(m->hn, D, m->wqkv[l], M);
        q.D = D; set_rope(q, m, 0); q.qk = m->qk; q.vt = m->vt; q.ldvt = m->Rtot;
        CK(run_gemm(m, q, m->wqkv[l], EPI_QKV, false, 128, st));
        static const int dump_qkv = getenv("F5HIP_DUMP_QKV") ? atoi(getenv("F5HIP_DUMP_QKV")) : -1;   // diagnostics, read once
        if (dump_qkv >= 0 && l == dump_qkv / 100 && ti == dump_qkv % 100) debug_dump_qkv(m, st);
        CK(launch_attention(m, st));
        GemmArgs o = gemm_base(m->ao, D, m->wout[l], M);
        o.mul = ml + 2 * D; o.res = m->h; o.ldres = D; o.out_f32 = m->h; o.ldo = D;
        o.row_keep = m->any_masked ? m->d_row_keep : nullptr;
        CK(run_gemm_ln(m, o, m->wout[l], block_ln(ml + 3 * D, ml + 4 * D), st));
        GemmArgs f1 = gemm_base(m->hn, D, m->wff1[l], M);
        f1.act = ACT_GELU_TANH; f1.out_hi = m->ff.hi; f1.out_lo = m->ff.lo; f1.ldob = F; f1.f16_out = m->blk_f16 ? 1 : 0;
        CK(run_gemm(m, f1, m->wff1[l], EPI_GENERIC, false, 128, st));
        GemmArgs f2 = gemm_base(m->ff, F, m->wff2[l], M);
        f2.mul = ml + 5 * D; f2.res = m->h; f2.ldres = D; f2.out_f32 = m->h; f2.ldo = D;
        if (l + 1 < nb) {
            CK(run_gemm_ln(m, f2, m->wff2[l], block_ln(ml + 6 * D, ml + 7 * D), st));   // norm 1 of block l + 1
        } else if (n_blocks < 0) {
            LnArgs lf = block_ln(mf + D, mf);      // final norm: (scale, shift) order, split-bf16 planes for proj_out
            lf.f16_out = 0;
            CK(run_gemm_ln(m, f2, m->wff2[l], lf, st));
        } else {
            CK(run_gemm(m, f2, m->wff2[l], EPI_GENERIC, false, 64, st));
        }
    }
    if (n_blocks >= 0) return 0;
    if (nb == 0) {                                     // (a model without blocks: the final norm has no GEMM to ride on)
        LnArgs lf = block_ln(mf + D, mf);
        lf.f16_out = 0;
        CK(run_ln(lf, st));
    }
    GemmArgs po = gemm_base(m->hn, D, m->proj_out, M);
    po.out_f32 = m->pred; po.ldo = 128;
    CK(run_gemm(m, po, m->proj_out, EPI_GENERIC, false, 128, st));
    return 0;
}

// -------------------------------------------------------------------------------------------------
// public entry points
// -------------------------------------------------------------------------------------------------
int f5hip_dit_forward(f5hip_dit* m, int32_t n_seq, const int32_t* seq_len, const int32_t* kv_len, const float* x_dev,
                      const float* cond_dev, const int32_t* text, int32_t nt_max, float time, const uint8_t* drop_audio_cond,
                      const uint8_t* drop_text, int32_t n_blocks, float* out_dev, float* h_out_dev, void* stream) {
    if (!m || !m->finalized) return fail(-1, "model not finalized");
    if (n_seq <= 0 || !seq_len || !x_dev || !cond_dev || !text) return fail(-1, "dit_forward: bad argument");
    ProfScope prof_scope(m->prof);
    hipStream_t st = (hipStream_t)stream;
    std::vector<SeqDesc> seqs(n_seq);
    int f0 = 0;
    m->h_seq_len.assign(n_seq, 0);
    for (int i = 0; i < n_seq; i++) {
        if (seq_len[i] <= 0 || seq_len[i] > 4096) return fail(-1, "seq_len[%d] = %d out of range", i, seq_len[i]);
        seqs[i] = {seq_len[i], kv_len ? kv_len[i] : seq_len[i], f0, i, drop_audio_cond ? drop_audio_cond[i] : 0, drop_text ? drop_text[i] : 0, 0};
        seqs[i].c_len = nt_max;   // MMDiT.forward embeds every position of its [b, nt] text tensor, fillers included (mmdit.py:37-52, no text mask)
        if (seqs[i].kvlen <= 0 || seqs[i].kvlen > seqs[i].len) return fail(-1, "kv_len[%d] out of range", i);
        m->h_seq_len[i] = seq_len[i];
        f0 += seq_len[i];
    }
    CK(setup_sequences(m, seqs, f0, text, nt_max, nullptr, st));
    const int M = m->M, mel = m->cfg.mel_dim, D = m->cfg.dim;
    hipLaunchKernelGGL(split_rows_kernel, dim3(M), dim3(256), 0, st, x_dev, mel, mel, M, m->d_row_frame, m->xs.hi, m->xs.lo, 128, 0);
    CKL("split x");
    CK(precompute_text_and_ce(m, cond_dev, st));
    CK(precompute_time(m, &time, 1, st));
    CK(forward_step(m, 0, n_blocks, st));
    // gather rows back to the caller's packed frame order
    if (n_blocks < 0) {
        if (!out_dev) return fail(-1, "out_dev is null");
        hipLaunchKernelGGL(gather_rows_kernel, dim3(f0), dim3(128), 0, st, m->pred, 128, mel, f0, m->d_urow_c, out_dev, mel);
    } else {
        if (!h_out_dev) return fail(-1, "h_out_dev is null");
        hipLaunchKernelGGL(gather_rows_kernel, dim3(f0), dim3(128), 0, st, m->h, D, D, f0, m->d_urow_c, h_out_dev, D);
    }
    CKL("gather rows");
    return 0;
}

int f5hip_dit_read_tap(f5hip_dit* m, const char* tap, float* dst_dev, int64_t numel, void* stream) {
    if (!m || !tap || !dst_dev) return fail(-1, "read_tap: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (!strcmp(tap, "text_embed")) {
        const int Td = m->cfg.text_dim;
        if (numel != (int64_t)m->n_frames * Td) return fail(-1, "read_tap: numel mismatch");
        hipLaunchKernelGGL(gather_rows_kernel, dim3(m->n_frames), dim3(128), 0, st, m->te, Td, Td, m->n_frames, m->d_urow_c, dst_dev, Td);
        CKL("gather tap");
        return 0;
    }
    return fail(-1, "unknown tap %s", tap);
}

int f5hip_cfm_sample(f5hip_dit* m, int32_t n_utt, const int32_t* dur, const float* cond_dev, const uint8_t* cond_mask,
                     const int32_t* text, int32_t nt_max, const float* y0_dev, const float* t_grid, int32_t steps,
                     float cfg_strength, float* out_dev, void* stream) {
    return f5hip_cfm_sample_masked(m, n_utt, dur, nullptr, cond_dev, cond_mask, text, nt_max, y0_dev, t_grid, steps, cfg_strength, out_dev, stream);
}

int f5hip_set_attention_shape_invariant(int32_t on) {
    f5_set_attn_shape_invariant(on);
    return 0;
}

int f5hip_dit_set_attention_shape_invariant(f5hip_dit* m, int32_t on) {
    if (!m) return fail(-1, "null model");
    m->attn_invariant = on < 0 ? -1 : (on != 0);
    return 0;
}

int f5hip_dit_set_profiling(f5hip_dit* m, int32_t enabled) {
    if (!m) return fail(-1, "null model");
    if (!m->prof) m->prof = new ProfState();
    m->prof->set(enabled != 0);
    return 0;
}

int f5hip_dit_get_profile(f5hip_dit* m, const char* kernel_class, double* total_ms, int64_t* launches) {
    if (!m || !kernel_class) return fail(-1, "dit_get_profile: bad argument");
    if (!m->prof) return fail(-1, "dit_get_profile: f5hip_dit_set_profiling was never called on this handle");
    return m->prof->get(kernel_class, total_ms, launches);
}

int f5hip_dit_set_ode_method(f5hip_dit* m, int32_t method) {
    if (!m) return fail(-1, "null model");
    if (method != 0 && method != 1) return fail(-1, "ode method %d: 0 = euler, 1 = midpoint", method);
    m->ode_method = method;
    return 0;
}

int f5hip_cfm_sample_masked(f5hip_dit* m, int32_t n_utt, const int32_t* dur, const int32_t* kv_len, const float* cond_dev, const uint8_t* cond_mask,
                            const int32_t* text, int32_t nt_max, const float* y0_dev, const float* t_grid, int32_t steps,
                            float cfg_strength, float* out_dev, void* stream) {
    if (!m || !m->finalized) return fail(-1, "model not finalized");
    if (n_utt <= 0 || !dur || !cond_dev || !cond_mask || !text || !y0_dev || !t_grid || !out_dev || steps <= 0)
        return fail(-1, "cfm_sample: bad argument");
    ProfScope prof_scope(m->prof);
    hipStream_t st = (hipStream_t)stream;
    const bool use_cfg = !(cfg_strength < 1e-5f);
    const int mel = m->cfg.mel_dim;
    std::vector<SeqDesc> seqs;
    int f0 = 0;
    m->h_seq_len.clear();
    for (int u = 0; u < n_utt; u++) {
        if (dur[u] <= 0 || dur[u] > 4096) return fail(-1, "dur[%d] = %d out of range", u, dur[u]);
        const int kv = kv_len ? kv_len[u] : dur[u];
        if (kv <= 0 || kv > dur[u]) return fail(-1, "kv_len[%d] = %d out of range (1..%d)", u, kv, dur[u]);
        // MMDiT text stream: with batch-1 semantics a unit's text tensor is its own tokens (the reference's per-item call pads nothing); with
        // the padded-batch semantics every item carries the batch's nt positions, fillers included
        int c_len = nt_max;
        if (!kv_len) { c_len = 0; while (c_len < nt_max && text[(size_t)u * nt_max + c_len] != -1) c_len++; }
        seqs.push_back({dur[u], kv, f0, u, 0, 0, 0});
        seqs.back().c_len = std::max(c_len, 1);
        m->h_seq_len.push_back(dur[u]);
        if (use_cfg) {
            seqs.push_back({dur[u], kv, f0, u, 1, 1, 1});
            seqs.back().c_len = std::max(c_len, 1);
            m->h_seq_len.push_back(dur[u]);
        }
        f0 += dur[u];
    }
    CK(setup_sequences(m, seqs, f0, text, nt_max, cond_mask, st));
    const int M = m->M;
    if (hipMemcpyAsync(m->xstate, y0_dev, sizeof(float) * (size_t)f0 * mel, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(-6, "y0 copy");
    hipLaunchKernelGGL(split_rows_kernel, dim3(M), dim3(256), 0, st, m->xstate, mel, mel, M, m->d_row_frame, m->xs.hi, m->xs.lo, 128, 0);
    CKL("split x");
    CK(precompute_text_and_ce(m, cond_dev, st));
    if (m->ode_method == 0) {
        CK(precompute_time(m, t_grid, steps, st));
        for (int i = 0; i < steps; i++) {
            CK(forward_step(m, i, -1, st));
            prof_begin(PROF_OTHER, st);
            hipLaunchKernelGGL(cfg_euler_kernel, dim3(f0), dim3(128), 0, st, m->xstate, m->xstate, mel, f0, m->pred, 128, m->d_urow_c, m->d_urow_u,
                               cfg_strength, t_grid[i + 1] - t_grid[i], m->xs.hi, m->xs.lo, 128);
            prof_end(PROF_OTHER, st);
            CKL("cfg_euler");
        }
    } else {
        // explicit midpoint on the fixed grid (torchdiffeq method="midpoint"): time points 2i = t_i, 2i + 1 = t_i + dt_i / 2
        if (2 * steps > 128) return fail(-8, "midpoint: at most 64 steps per call (got %d)", steps);
        std::vector<float> t2((size_t)2 * steps);
        for (int i = 0; i < steps; i++) {
            const float half = 0.5f * (t_grid[i + 1] - t_grid[i]);
            t2[2 * i] = t_grid[i];
            t2[2 * i + 1] = t_grid[i] + half;
        }
        CK(precompute_time(m, t2.data(), 2 * steps, st));
        for (int i = 0; i < steps; i++) {
            const float dt = t_grid[i + 1] - t_grid[i];
            CK(forward_step(m, 2 * i, -1, st));
            prof_begin(PROF_OTHER, st);
            hipLaunchKernelGGL(cfg_euler_kernel, dim3(f0), dim3(128), 0, st, m->xmid, m->xstate, mel, f0, m->pred, 128, m->d_urow_c, m->d_urow_u,
                               cfg_strength, 0.5f * dt, m->xs.hi, m->xs.lo, 128);
            prof_end(PROF_OTHER, st);
            CKL("cfg_euler half");
            CK(forward_step(m, 2 * i + 1, -1, st));
            prof_begin(PROF_OTHER, st);
            hipLaunchKernelGGL(cfg_euler_kernel, dim3(f0), dim3(128), 0, st, m->xstate, m->xstate, mel, f0, m->pred, 128, m->d_urow_c, m->d_urow_u,
                               cfg_strength, dt, m->xs.hi, m->xs.lo, 128);
            prof_end(PROF_OTHER, st);
            CKL("cfg_euler full");
        }
    }
    hipLaunchKernelGGL(final_select_kernel, dim3(f0), dim3(128), 0, st, m->xstate, cond_dev, m->d_frame_is_cond, mel, f0, out_dev);
    CKL("final_select");
    return 0;
}

#include "vocos.h"
#include "bigvgan.h"
#include "unit_ops.h"
#ifdef F5HIP_EXPERIMENTS
#include "experiments/debug_bench.h"
#endif
