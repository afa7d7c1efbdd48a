// HBM/L2-bound row kernels of the DiT / Vocos paths: LayerNorm (+AdaLN modulation, + optional depthwise
// conv k=7 in front), GRN, embedding gather, CFG + Euler update, packing helpers.  One wave (64 lanes) per
// row, float4 loads, wave-shuffle reductions, fp32 statistics, split-bf16 outputs that feed the MFMA GEMMs.
#pragma once
#include "common.h"

#include "ln_row.h"

// Row -> workgroup mapping, XCD-matched (XCD = 1; speed only, any bijection is correct): consecutive workgroup ids go round-robin to the 8 XCDs
// and the block GEMMs give XCD x the row slabs of rows [x M / 8, (x + 1) M / 8) (gemm5_tile_of_block) -- both the GEMM that just wrote the
// residual rows this kernel reads and the GEMM that reads the operand rows it writes.  With the same blocking here a row stays in one XCD's L2
// from the residual epilogue through the norm to the next GEMM's first fill instead of crossing the fabric twice (F5HIP_LN_XCD=0: off, A/B).
template <int NV, int XCD = 1>
__global__ __launch_bounds__(256) void ln_kernel(const LnArgs p) {
    unsigned b = blockIdx.x;
    if (XCD && (gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    ln_row<NV>(p, b * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Rotary factors per ROW: out_cos / out_sin [R][32] = table[row_pos[r]][32].  Once per sampler call (the positions do not change over the ODE steps):
// the QKV epilogues then fetch a row's factors with one load instead of two dependent ones (row_pos, then the table) -- the 32 rotary tiles of the C2
// launch finished 2.7 us behind the other 224.
__global__ __launch_bounds__(256) void rope_rows_kernel(const int* row_pos, const float* rope_cos, const float* rope_sin, int R, int max_pos, float* out_cos, float* out_sin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * 32) return;
    const int r = i >> 5, j = i & 31;
    const int pos = min(max(row_pos[r], 0), max_pos - 1);
    out_cos[i] = rope_cos[pos * 32 + j];
    out_sin[i] = rope_sin[pos * 32 + j];
}

// ------------------------------------------------------------------------------------------------
// GRN (F/model/modules.py:231-234): Gx[c] = ||y[:, c]||_2 over the frames of one sequence.
__global__ __launch_bounds__(256) void grn_stats_kernel(const float* y, int ldy, int C, const int* seq_row0,
                                                        const int* seq_len, float* gx /*[n_seq][C]*/) {
    const int seq = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = seq_row0[seq], n = seq_len[seq];
    // 8 independent loads in flight per thread: the serial one-row-at-a-time loop was pure load latency (327 us at N = 1404)
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* col = y + (size_t)r0 * ldy + c;
    int r = 0;
    for (; r + 8 <= n; r += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = col[(size_t)(r + u) * ldy];
#pragma unroll
        for (int u = 0; u < 8; u++) a8[u] += t[u] * t[u];
    }
    for (; r < n; r++) {
        const float t = col[(size_t)r * ldy];
        a8[0] += t * t;
    }
    const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    gx[(size_t)seq * C + c] = sqrtf(acc);
}

// y' = gamma * (y * Nx) + beta + y,  Nx = Gx / (mean_c(Gx) + 1e-6)  -> split bf16
__global__ __launch_bounds__(256) void grn_apply_kernel(const float* y, int ldy, int C, int M, const int* row_seq,
                                                        const float* gx, const float* gamma, const float* beta,
                                                        __bf16* out_hi, __bf16* out_lo, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int seq = row_seq[row];
    if (seq < 0) return;
    const float* g = gx + (size_t)seq * C;
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s += g[c];
    const float denom = wave_sum(s) / (float)C + 1e-6f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 yv = *reinterpret_cast<const float4*>(y + (size_t)row * ldy + c);
        const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
        bf16x4 hi, lo;
        float ov[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float nx = g[c + e] / denom;
            ov[e] = gamma[c + e] * (yy[e] * nx) + beta[c + e] + yy[e];
        }
        split_bf16x4(ov, hi, lo);
        *reinterpret_cast<bf16x4*>(out_hi + (size_t)row * ldo + c) = hi;
        if (out_lo) *reinterpret_cast<bf16x4*>(out_lo + (size_t)row * ldo + c) = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// Text embedding gather (F/model/backbones/dit.py:48-64): e[m] = Embedding[id[m]] + pos_table[min(pos, 4095)]
__global__ __launch_bounds__(256) void text_gather_kernel(const float* emb, const float* pos_table, int C, int M,
                                                          const int* row_token, const int* row_pos, int add_pos,
                                                          float* out, int ldo, int max_pos) {
    const int row = blockIdx.x;
    if (row >= M) return;
    const int tok = row_token[row];
    if (tok < 0) return;   // padding row of the packed layout
    const int pos = min(row_pos[row], max_pos);   // get_pos_embed_indices clamps below the table length: 4096 (DiT), 1024 (MMDiT)
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = emb[(size_t)tok * C + c];
        if (add_pos) v += pos_table[(size_t)pos * C + c];
        out[(size_t)row * ldo + c] = v;
    }
}

// fp32 [rows][C] -> split bf16 [rows][ldo] (columns >= C are left untouched: buffers are zero-initialised)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* x, int ldx, int C, int M, const int* row_src,
                                                         __bf16* out_hi, __bf16* out_lo, int ldo, int col0) {
    const int row = blockIdx.x;
    if (row >= M) return;
    const int src = row_src ? row_src[row] : row;
    for (int c = threadIdx.x; c < C; c += 256) {
        __bf16 hi, lo;
        const float v = src >= 0 ? x[(size_t)src * ldx + c] : 0.0f;
        split_bf16(v, hi, lo);
        out_hi[(size_t)row * ldo + col0 + c] = hi;
        if (out_lo) out_lo[(size_t)row * ldo + col0 + c] = lo;
    }
}

// fp32 [rows][C] -> ONE fp16 plane [rows][ldo] at column col0 (saturating; the operand of a PREC_F16 GEMM), 4 columns per lane
__global__ __launch_bounds__(256) void cast_rows_f16_kernel(const float* x, int ldx, int C, int M, __bf16* out, int ldo, int col0) {
    const int row = blockIdx.x;
    if (row >= M) return;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c);
        const float y[4] = {v[0], v[1], v[2], v[3]};
        store_f16x4(out + (size_t)row * ldo + col0 + c, y);
    }
}

// ------------------------------------------------------------------------------------------------
// CFG combine + Euler step (F/model/cfm.py:176 and torchdiffeq fixed-grid Euler):
//   v = p + (p - p0) * cfg ;  x += dt * v
// pred: [M_pad][ldp] rows of the conditional branch at row_c[u], unconditional at row_u[u] (or -1).
// Also refreshes the split-bf16 copy of x that feeds the next step's input projection for both branches.
// xout = xbase + dt * v: xout == xbase for an Euler step; the midpoint rule writes its half step to a scratch state and takes the full
// step from the untouched xbase.
__global__ __launch_bounds__(128) void cfg_euler_kernel(float* xout /*[U][mel]*/, const float* xbase, int mel, int U, const float* pred,
                                                        int ldp, const int* urow_c, const int* urow_u, float cfg,
                                                        float dt, __bf16* xs_hi, __bf16* xs_lo, int ldx) {
    const int u = blockIdx.x;
    if (u >= U) return;
    const int c = threadIdx.x;
    if (c >= mel) return;
    const int rc = urow_c[u], ru = urow_u[u];
    const float pc = pred[(size_t)rc * ldp + c];
    float v = pc;
    if (ru >= 0) v = pc + (pc - pred[(size_t)ru * ldp + c]) * cfg;
    const float xn = xbase[(size_t)u * mel + c] + dt * v;
    xout[(size_t)u * mel + c] = xn;
    __bf16 hi, lo;
    split_bf16(xn, hi, lo);
    xs_hi[(size_t)rc * ldx + c] = hi;
    xs_lo[(size_t)rc * ldx + c] = lo;
    if (ru >= 0) {
        xs_hi[(size_t)ru * ldx + c] = hi;
        xs_lo[(size_t)ru * ldx + c] = lo;
    }
}

// out = cond_mask ? cond : x  (F/model/cfm.py:204); one block per utterance frame
__global__ __launch_bounds__(128) void final_select_kernel(const float* xstate, const float* cond, const int* frame_is_cond,
                                                           int mel, int U, float* out) {
    const int u = blockIdx.x, c = threadIdx.x;
    if (u >= U || c >= mel) return;
    out[(size_t)u * mel + c] = frame_is_cond[u] ? cond[(size_t)u * mel + c] : xstate[(size_t)u * mel + c];
}

// fp32 [R][C] (row-major, ld = C) -> split bf16 [R_pad][C_pad] with zero fill; used by the weight packer
// (lo == nullptr: ONE fp16 plane in hi, the weight format of the PREC_F16 GEMMs)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* w, int R, int C, int ldw, __bf16* hi, __bf16* lo,
                                                          int C_pad) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < C_pad; c += 256) {
        float v = (r < R && c < C) ? w[(size_t)r * ldw + c] : 0.0f;
        if (!lo) {
            reinterpret_cast<_Float16*>(hi)[(size_t)r * C_pad + c] = sat_f16(v);
            continue;
        }
        __bf16 h, l;
        split_bf16(v, h, l);
        hi[(size_t)r * C_pad + c] = h;
        lo[(size_t)r * C_pad + c] = l;
    }
}

// one 16-bit plane [R_pad][ld] (row-major, K = C_pad contiguous) -> MFMA-fragment order (GemmArgs::Wf): thread = one 16-byte chunk
__global__ __launch_bounds__(256) void pack_frag_kernel(const __bf16* src, int R_pad, int K, int ld, __bf16* dst) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // chunk index in the destination
    const size_t n_chunks = (size_t)R_pad * K / 8;
    if (idx >= n_chunks) return;
    const int lane = (int)(idx & 63);
    const size_t blk = idx >> 6;
    const int kh = (int)(blk % (size_t)(K / 32)), cb = (int)(blk / (size_t)(K / 32));
    const int row = cb * 16 + (lane & 15), k = kh * 32 + (lane >> 4) * 8;
    *reinterpret_cast<u32x4*>(dst + idx * 8) = *reinterpret_cast<const u32x4*>(src + (size_t)row * ld + k);
}

// dst[f][0..C) = src[frame_row[f]][0..C): packed-row layout -> caller's frame order
__global__ __launch_bounds__(128) void gather_rows_kernel(const float* src, int lds, int C, int n_frames, const int* frame_row,
                                                          float* dst, int ldd) {
    const int f = blockIdx.x;
    if (f >= n_frames) return;
    const int r = frame_row[f];
    for (int c = threadIdx.x; c < C; c += 128) dst[(size_t)f * ldd + c] = src[(size_t)r * lds + c];
}

// UNetT: the time embedding is the first token of every sequence (F/model/backbones/unett.py:184)
__global__ __launch_bounds__(256) void set_time_token_kernel(float* h, int D, const int* seq_row0, const float* temb) {
    float* row = h + (size_t)seq_row0[blockIdx.x] * D;
    for (int c = threadIdx.x; c < D; c += 256) row[c] = temb[c];
}
