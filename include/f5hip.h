/* libf5hip — C ABI of the MI355X-native F5-TTS inference hot path.
 *
 * Plain C, plain pointers and sizes, no torch types.  Pointers named *_dev are HIP device pointers
 * (e.g. torch tensor .data_ptr() on a ROCm device), all other pointers are host memory.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Every function returns 0 on success and a negative
 * code on failure; f5hip_last_error() gives the message.  Nothing here falls back to the CPU.
 *
 * Each entry point names the reference interface it replaces (F/ = src/server/f5_tts/ of
 * dwani-ai/tts-indic-server-f5); INTEGRATION.md shows the ctypes binding a maintainer would add.
 */
#ifndef F5HIP_H
#define F5HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F5HIP_ABI_VERSION 1

int f5hip_abi_version(void);
const char* f5hip_last_error(void);

/* ---------------------------------------------------------------- DiT backbone + CFM sampler ---------- */

/* model.arch of F/configs/F5TTS_*_train.yaml:24-30 (+ mel_dim, vocab size: F/infer/utils_infer.py:240-242). */
typedef struct f5hip_dit_config {
    int32_t dim, depth, heads, ff_mult, text_dim, conv_layers, mel_dim, text_num_embeds;
    int32_t gemm_planes; /* GEMM operand precision, all with fp32 accumulation:
                            2 = split-bf16 "bf16x3" everywhere (strictest: 1.1e-4 mel RMS vs the fp32 reference at C2);
                            3 = mixed: fp16 operands for the transformer-block GEMMs (QKV, out, FF1, FF2), bf16x3 for every GEMM that
                                touches the ODE state / embeddings / U-skips (3.1e-4 mel RMS for F5-Base at 32 NFE, 4.9e-4 for E2-Base
                                at 64 NFE, against the reference's own CFM.sample outputs: inside the 1e-3 bound);
                            1 = plain bf16 (fast, ~8e-3 mel RMS: outside the bound) */
    int32_t arch;        /* 0 = DiT (F5-TTS, F/model/backbones/dit.py), 1 = UNetT (E2-TTS, F/model/backbones/unett.py: text_dim = mel_dim, conv_layers = 0),
                            2 = MMDiT (F/model/backbones/mmdit.py: text_dim = dim, conv_layers = 0; state_dict keys transformer.audio_embed.*,
                                transformer.transformer_blocks.{i}.attn_norm_{c,x} / attn.to_{q,k,v}[_c] / attn.to_out[_c] / ff_{c,x}) */
} f5hip_dit_config;

typedef struct f5hip_dit f5hip_dit;

/* Replaces DiT.__init__ (F/model/backbones/dit.py:94-128). */
f5hip_dit* f5hip_dit_create(const f5hip_dit_config* cfg);
void f5hip_dit_destroy(f5hip_dit* m);

/* Replaces model.load_state_dict (F/infer/utils_infer.py:195-209): one call per tensor, `name` is the
 * reference checkpoint key with the "ema_model." prefix stripped ("transformer.time_embed.time_mlp.0.weight", ...),
 * `data` is host fp32 in the tensor's own row-major layout. */
int f5hip_dit_load_param(f5hip_dit* m, const char* name, const float* data, int64_t numel);
/* Checks that every parameter arrived, packs the weights for the MFMA kernels (split bf16, padded, fused QKV /
 * AdaLN matrices) and uploads them.  Must be called once before any forward/sample call. */
int f5hip_dit_finalize(f5hip_dit* m);

/* One evaluation of DiT.forward (F/model/backbones/dit.py:130-163) for n_seq independent sequences.
 *   seq_len[i]   frames of sequence i (rows of x/cond/out belonging to it, packed back to back)
 *   kv_len[i]    valid keys (== seq_len[i] for mask=None; < seq_len reproduces the reference's padded-batch
 *                key-padding mask and zeroed attention rows, F/model/modules.py:429-447)
 *   x_dev, cond_dev  fp32 [sum(seq_len)][mel_dim];  text: int32 [n_seq][nt_max], -1 padded;  time: scalar t
 *   drop_audio_cond[i], drop_text[i]: the two CFG switches of the reference signature
 *   n_blocks     -1 = whole network (out_dev = [sum(seq_len)][mel_dim]); k >= 0 = stop after k transformer
 *                blocks and return the residual stream in h_out_dev [sum(seq_len)][dim] (parity taps)
 */
int f5hip_dit_forward(f5hip_dit* m, int32_t n_seq, const int32_t* seq_len, const int32_t* kv_len,
                      const float* x_dev, const float* cond_dev, const int32_t* text, int32_t nt_max, float time,
                      const uint8_t* drop_audio_cond, const uint8_t* drop_text, int32_t n_blocks,
                      float* out_dev, float* h_out_dev, void* stream);

/* Copies an internal fp32 activation of the last forward for parity taps: "text_embed" -> [sum(seq_len)][text_dim]. */
int f5hip_dit_read_tap(f5hip_dit* m, const char* tap, float* dst_dev, int64_t numel, void* stream);

/* The ODE loop of CFM.sample (F/model/cfm.py:160-204): Euler over t_grid with classifier-free guidance,
 * each utterance sampled with the reference's batch-1 semantics (mask=None).
 *   dur[u]          total frames of utterance u (already max(lens+1, duration) clamped, cfm.py:136-137)
 *   cond_dev        fp32 [sum(dur)][mel_dim] mel conditioning, zero padded to dur (cfm.py:144)
 *   cond_mask       uint8 [sum(dur)] 1 where the frame is conditioning (cfm.py:129-131,145-146)
 *   text            int32 [n_utt][nt_max] token ids, -1 padded (cfm.py:116-121)
 *   y0_dev          fp32 [sum(dur)][mel_dim] initial noise (cfm.py:181-186)
 *   t_grid          float [steps+1] (cfm.py:196-198)
 *   cfg_strength    < 1e-5 skips the unconditional branch (cfm.py:170-171)
 *   out_dev         fp32 [sum(dur)][mel_dim] = where(cond_mask, cond, x_1) (cfm.py:204)
 */
int f5hip_cfm_sample(f5hip_dit* m, int32_t n_utt, const int32_t* dur, const float* cond_dev,
                     const uint8_t* cond_mask, const int32_t* text, int32_t nt_max, const float* y0_dev,
                     const float* t_grid, int32_t steps, float cfg_strength, float* out_dev, void* stream);

/* The same loop with the reference's PADDED-BATCH semantics (what CFM.sample does for batch > 1: F/model/cfm.py:151-154, mask =
 * lens_to_mask(duration); F/model/modules.py:429-447): every item is laid out with dur[u] = the batch maximum, kv_len[u] = its own
 * duration; keys >= kv_len[u] are masked in every attention, the attention output rows >= kv_len[u] are zeroed, and everything
 * else (text / conv embeddings, feed-forward, the ODE update of the padded rows) runs over all dur[u] rows exactly like the
 * reference's padded tensors.  kv_len == NULL is f5hip_cfm_sample. */
int f5hip_cfm_sample_masked(f5hip_dit* m, int32_t n_utt, const int32_t* dur, const int32_t* kv_len, const float* cond_dev,
                            const uint8_t* cond_mask, const int32_t* text, int32_t nt_max, const float* y0_dev,
                            const float* t_grid, int32_t steps, float cfg_strength, float* out_dev, void* stream);

/* The fixed-grid solver both sample calls use: replaces CFM(odeint_kwargs=dict(method=...)) (F/model/cfm.py:37-41,72,200; set from
 * load_model(ode_method=...), F/infer/utils_infer.py:251).  0 = "euler" (default): x += dt * v(t_i, x).  1 = "midpoint":
 * x += dt * v(t_i + dt / 2, x + dt / 2 * v(t_i, x)), two backbone evaluations per step, at most 64 steps per call. */
int f5hip_dit_set_ode_method(f5hip_dit* m, int32_t method);

/* Attention kernel choice.  0 (default): the fastest form per launch shape -- a launch with 192-query tiles (e.g. one 10 s utterance) runs
 * the SIMD-balanced kernel, in which a third of the query blocks accumulate the two key halves of every tile separately and merge them at
 * the end: the same sums in a different fp32 association, so a sequence's output can differ in the last bits from what it gets inside a
 * larger batch (the softmax offsets and the fp16 probabilities are the same in every variant).  In the mixed GEMM mode any last-bit
 * difference grows to that mode's rounding-noise floor over a forward pass (profiles/r03_attn_mode_tapdiff.txt).  1: shape-invariant arithmetic -- every variant adds every query's terms in one order, so a
 * sequence's output does not depend on what it is batched with (bit-identical); ~3 % slower at batch 1.
 * f5hip_set_attention_shape_invariant sets the PROCESS DEFAULT; f5hip_dit_set_attention_shape_invariant sets it for one handle
 * (1 / 0, or -1 = follow the process default again), so two handles in one process -- a serving handle that promises batch-independent
 * results next to a latency-bound one -- do not share the setting. */
int f5hip_set_attention_shape_invariant(int32_t on);
int f5hip_dit_set_attention_shape_invariant(f5hip_dit* m, int32_t on);
/* Per-handle profiling: like f5hip_set_profiling / f5hip_get_profile below, but the HIP-event spans, their pool and the totals belong to this
 * handle alone (a handle that never called it records into the process-wide state when that is enabled).  Calls on one handle are still one
 * at a time; calls on different handles may come from different threads. */
int f5hip_dit_set_profiling(f5hip_dit* m, int32_t enabled);
int f5hip_dit_get_profile(f5hip_dit* m, const char* kernel_class, double* total_ms, int64_t* launches);
/* Per-kernel timing of the last f5hip_cfm_sample call when profiling was enabled with
 * f5hip_set_profiling(1): average milliseconds per launch of the named kernel class
 * ("gemm", "attn", "ln", "other") measured with HIP events on the launch stream, and launch counts. */
int f5hip_set_profiling(int32_t enabled);
int f5hip_get_profile(const char* kernel_class, double* total_ms, int64_t* launches);
/* Launch counters of the GEMM dispatcher since the last reset (test instrumentation: proves which kernel a config exercised):
 * "gemm5_rb11" / "gemm5_rb8" (exact-fit tile heights 176 / 128), "gemm5_wide" (128- and 192-column tiles), "gemm3_wide",
 * "gemm6" (256 x 256 ping-pong tiles: the batch-mode shapes);
 * name "reset" zeroes all of them (value may be NULL). */
int f5hip_get_counter(const char* name, int64_t* value);

/* ---------------------------------------------------------------- per-kernel unit ops ----------------- */
/* One production kernel each, fp32 device tensors in and out, through the same dispatcher the sampler uses (SURVEY section 8(b)-4:
 * "per-kernel ops for unit parity").  They allocate their operand planes per call: test / tooling entry points, not the hot path.
 *
 * f5hip_op_gemm: out = (act(A W^T + bias), rows with row_keep == 0 zeroed) * mul + res   -- nn.Linear + the fused epilogue of
 *   Attention.to_out / FeedForward (F/model/modules.py:324-328,441-447,566-571).
 *   a_dev [M][K], w_dev [N][K] (nn.Linear layout), bias_dev [N] | NULL, mul_dev [N] | NULL (AdaLN gate), res_dev [M][N] | NULL,
 *   row_keep_host uint8 [M] | NULL (host); prec 1 = bf16, 2 = split bf16 (bf16x3), 3 = fp16 operands, fp32 accumulate;
 *   act 0 none, 1 GELU(tanh), 2 GELU(erf), 3 Mish, 4 SiLU.  out_dev fp32 [M][N], or out16_dev: one fp16 plane [M][N] (the operand the
 *   next fp16 GEMM reads; saturates at +-65504).  iters > 0: also times `iters` launches with HIP events on `stream`, cycling through
 *   w_copies copies of the packed weights (a pool larger than the Infinity Cache makes them HBM-cold as in the real forward). */
int f5hip_op_gemm(int32_t M, int32_t N, int32_t K, const float* a_dev, const float* w_dev, const float* bias_dev, int32_t prec,
                  int32_t act, const float* mul_dev, const float* res_dev, const uint8_t* row_keep_host, float* out_dev,
                  uint16_t* out16_dev, int32_t w_copies, int32_t iters, double* avg_us, void* stream);
/* f5hip_op_qkv: fused to_q | to_k | to_v projection with its epilogue: bias, rotary embedding on channels 0..63 (head 0, interleaved
 *   pairs) of q and k, q * log2(e) / 8 (the attention kernel's scores are base-2 exponents), V transposed (F/model/modules.py:409-426).  a_dev [M][D], w_dev [3 D][D], bias_dev [3 D], row_pos host
 *   int32 [M] (rotary position of every row, 0..4096); outputs fp16 (saturated): qk_dev [ceil128(M)][2 D], vt_dev [D][ceil128(M)] with the tokens of
 *   every aligned group of 16 in the order 0-3, 8-11, 4-7, 12-15 (the order the attention kernel's PV fragments read them in). */
int f5hip_op_qkv(int32_t M, int32_t D, const float* a_dev, const float* w_dev, const float* bias_dev, const int32_t* row_pos,
                 int32_t prec, uint16_t* qk_dev, uint16_t* vt_dev, int32_t iters, double* avg_us, void* stream);
/* f5hip_op_attention: softmax(q k^T / 8 + key-padding mask) v per (sequence, head), head dim 64 -- F.scaled_dot_product_attention with the
 *   reference's [b, 1, 1, n] key mask (F/model/modules.py:424-436).  q_dev / k_dev / v_dev / out_dev fp32 [sum(seq_len)][64 heads], sequences
 *   packed back to back; kv_len[i] <= seq_len[i] valid keys (NULL: all).  Operands are rounded to fp16 (saturated) like the QKV epilogue's outputs.
 *   impl 3 = the production kernel (attn3); 4 / 5 = the experimental unequal-wave / ping-pong kernels (attn3 unless the library was built
 *   with experiments). */
int f5hip_op_attention(int32_t n_seq, const int32_t* seq_len, const int32_t* kv_len, int32_t heads, const float* q_dev, const float* k_dev,
                       const float* v_dev, float* out_dev, int32_t impl, int32_t iters, double* avg_us, void* stream);
/* f5hip_op_joint_attention: the joint attention of the MMDiT blocks (JointAttnProcessor, F/model/modules.py:496-522): per sequence the
 *   queries and the keys are its audio rows followed by its text rows; only audio keys can be padding (x_kvlen[i] <= x_len[i] valid; NULL:
 *   all).  q_dev / k_dev / v_dev / out_dev fp32 [sum(x_len) + sum(c_len)][64 heads]: all audio frames sequence by sequence, then all
 *   text tokens sequence by sequence.  Operands are rounded to fp16 (saturated) like the QKV epilogue's outputs. */
int f5hip_op_joint_attention(int32_t n_seq, const int32_t* x_len, const int32_t* x_kvlen, const int32_t* c_len, int32_t heads,
                             const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, void* stream);
/* f5hip_op_layernorm: y = LN(x) * (gain_off + scale) + shift (AdaLN: gain_off 1; affine LN: gain_off 0; F/model/modules.py:285-290),
 *   rms = 1: x-transformers RMSNorm y = x / max(|x|_2, 1e-12) * sqrt(D) * scale.  All fp32 [M][D] / [D]. */
int f5hip_op_layernorm(int32_t M, int32_t D, const float* x_dev, const float* scale_dev, const float* shift_dev, float gain_off, float eps,
                       int32_t rms, float* out_dev, void* stream);
/* f5hip_op_conv1d: one nn.Conv1d(c_in, c_out, k, dilation = dil, padding = dil (k - 1) / 2) + bias + res of the BigVGAN generator (its
 *   AMPBlock1 convolutions: k 3 / 7 / 11, dilation 1 / 3 / 5) over channel-last rows: `batch` sequences of pitch P rows (P % 128 == 0), T valid,
 *   zero padding at the sequence bounds.  x_dev fp32 [batch P][c_in], w_host [c_out][c_in][k] (the module's weight layout), bias_host [c_out]
 *   or NULL, res_dev fp32 [batch P][c_out] or NULL, out_dev fp32 [batch P][c_out] (rows >= T of a sequence are unspecified).
 *   prec 2 = split bf16, 3 = one fp16 plane.  impl 0 = implicit GEMM (gemm.h), 5 = sliding-window kernel (conv5.h; fails if it does not
 *   cover the shape).  iters > 0: average microseconds per launch in *avg_us.  stamps_host (optional, impl 5): [stamp_blocks][16] cycle
 *   stamps of a diagnostics launch (layout: csrc/conv5.h). */
int f5hip_op_conv1d(int32_t batch, int32_t P, int32_t T, int32_t c_in, int32_t c_out, int32_t k, int32_t dil, const float* x_dev,
                    const float* w_host, const float* bias_host, const float* res_dev, float* out_dev, int32_t prec, int32_t impl,
                    int32_t iters, double* avg_us, uint64_t* stamps_host, int32_t stamp_blocks, void* stream);

/* ---------------------------------------------------------------- Vocos vocoder ----------------------- */

typedef struct f5hip_vocos_config {
    int32_t in_channels, dim, intermediate_dim, num_layers, n_fft, hop_length;
    int32_t gemm_planes;
} f5hip_vocos_config;
typedef struct f5hip_vocos f5hip_vocos;

/* Replaces Vocos.from_hparams + load_state_dict (F/infer/utils_infer.py:104-115); names are vocos 0.1.0 keys
 * ("backbone.embed.weight", "backbone.convnext.0.dwconv.weight", ..., "head.out.weight"). */
f5hip_vocos* f5hip_vocos_create(const f5hip_vocos_config* cfg);
void f5hip_vocos_destroy(f5hip_vocos* v);
int f5hip_vocos_load_param(f5hip_vocos* v, const char* name, const float* data, int64_t numel);
int f5hip_vocos_finalize(f5hip_vocos* v);
/* Replaces vocoder.decode(mel) (F/infer/utils_infer.py:472): mel_dev fp32 [batch][in_channels][frames] ->
 * wave_dev fp32 [batch][hop_length * (frames - 1)]. */
int f5hip_vocos_decode(f5hip_vocos* v, int32_t batch, int32_t frames, const float* mel_dev, float* wave_dev,
                       void* stream);

/* ---------------------------------------------------------------- BigVGAN vocoder --------------------- */

/* BigVGAN v2 generator hyper-parameters (config.json of nvidia/bigvgan_v2_24khz_100band_256x: F/infer/utils_infer.py:122-126).
 * Supported family: upsample_kernel_sizes[i] == 2 * upsample_rates[i] (even), resblock "1" with three kernel sizes x three
 * dilations, snakebeta with log-scale parameters, no tanh / no bias at the final conv. */
typedef struct f5hip_bigvgan_config {
    int32_t num_mels, num_upsamples;
    int32_t upsample_rates[8], upsample_kernel_sizes[8];
    int32_t upsample_initial_channel;
    int32_t resblock_kernel_sizes[3];
    int32_t resblock_dilations[9];   /* [kernel index][dilation index] */
    int32_t gemm_planes;             /* conv operand precision, fp32 accumulation: 2 = split bf16, three MFMAs per product (the parity
                                        mode: 1.5e-5 max on the waveform); 3 = one fp16 plane (fast mode, NOT within the 1e-4 parity
                                        bound: ~1e-3 max / 2e-4 rms measured); 1 = plain bf16 */
} f5hip_bigvgan_config;
typedef struct f5hip_bigvgan f5hip_bigvgan;

/* Replaces bigvgan.BigVGAN.from_pretrained + remove_weight_norm (F/infer/utils_infer.py:116-129); names are the generator's
 * state_dict keys after remove_weight_norm ("conv_pre.weight", "ups.0.0.weight", "resblocks.0.convs1.0.weight",
 * "resblocks.0.activations.0.act.alpha", ..., "activation_post.act.beta", "conv_post.weight"). */
f5hip_bigvgan* f5hip_bigvgan_create(const f5hip_bigvgan_config* cfg);
void f5hip_bigvgan_destroy(f5hip_bigvgan* v);
int f5hip_bigvgan_load_param(f5hip_bigvgan* v, const char* name, const float* data, int64_t numel);
int f5hip_bigvgan_finalize(f5hip_bigvgan* v);
/* Replaces vocoder(mel) (F/infer/utils_infer.py:474): mel_dev fp32 [batch][num_mels][frames] ->
 * wave_dev fp32 [batch][frames * prod(upsample_rates)] (the reference's [batch, 1, n] squeezed), clamped to [-1, 1]. */
int f5hip_bigvgan_forward(f5hip_bigvgan* v, int32_t batch, int32_t frames, const float* mel_dev, float* wave_dev, void* stream);

/* ---------------------------------------------------------------- mel front-end ------------------------ */

/* Replaces MelSpec.forward with mel_spec_type="vocos" (F/model/modules.py:75-101,130-143): wave_dev fp32
 * [batch][n_samples] -> mel_dev fp32 [batch][n_mels][1 + n_samples / hop] (log of clamp(mel, 1e-5)). */
int f5hip_mel_spectrogram(int32_t batch, int32_t n_samples, const float* wave_dev, float* mel_dev, int32_t n_fft,
                          int32_t hop_length, int32_t n_mels, int32_t sample_rate, void* stream);

/* Replaces MelSpec.forward with mel_spec_type="bigvgan" (get_bigvgan_mel_spectrogram, F/model/modules.py:30-72): reflect pad
 * (n_fft - hop) / 2, STFT center=False, sqrt(re^2 + im^2 + 1e-9), librosa Slaney mel filterbank (fmax = sr / 2),
 * log(clamp(., 1e-5)): wave_dev fp32 [batch][n_samples] -> mel_dev fp32 [batch][n_mels][1 + (n_samples - hop) / hop]. */
int f5hip_mel_spectrogram_bigvgan(int32_t batch, int32_t n_samples, const float* wave_dev, float* mel_dev, int32_t n_fft,
                                  int32_t hop_length, int32_t n_mels, int32_t sample_rate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* F5HIP_H */
