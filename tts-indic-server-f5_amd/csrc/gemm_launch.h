// Non-template entry points of the MFMA kernels.  Every kernel family is its own translation unit (tu_*.hip) so the library builds
// in parallel and a kernel edit recompiles one unit; the host orchestration (f5hip.hip) sees only these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "attn_common.h"
#include "gemm_epilogue.h"

// gemm.h: register-staged 128 x {128, 64} x 32 kernel (prec 1 bf16, 2 split bf16, 3 fp16); conv = implicit-GEMM operand
hipError_t f5_launch_gemm_reg(int prec, int bn, bool conv, int epi, const GemmArgs& a, int m_pad, int n_pad, hipStream_t st);
// gemm3.h: warp-specialised LDS-DMA kernel, 128 x bn (128, or 256 for one-plane operands) x 32
hipError_t f5_launch_gemm3(int prec, int epi, int bn, const GemmArgs& a, int m_pad, int n_pad, hipStream_t st);
// gemm5.h: exact-fit (16 rb) x (16 cb) tiles, 64-deep k-steps, fp16 operands
hipError_t f5_launch_gemm5_generic(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st);
hipError_t f5_launch_gemm5_qkv(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st);
// gemm6.h: ping-pong tiles of `rows` (256 or 176) x 256 columns, fp16 operands: the batch-mode shapes (hipErrorInvalidValue: n_pad % 256, K % 64, D % 256)
hipError_t f5_launch_gemm6(int epi, int rows, const GemmArgs& a, int n_pad, hipStream_t st);
// gemm6 tile height: fewest (rounds on the 256 CUs) x (cost of a tile: a 176-row tile measures ~ 0.85 of a 256-row one -- k-loop 22.0 against
// 26.0 us at K = 1024, profiles/r03_gemm6_stamps_*.txt -- so it pays where it saves a round: out / FF2 at the C3 share, not FF1 / QKV); 0 = too few tiles
static inline int gemm6_choose_rows(int m_rows, int n_pad) {
    const long long t256 = (long long)((m_rows + 255) / 256) * (n_pad / 256), t176 = (long long)((m_rows + 175) / 176) * (n_pad / 256);
    if (t256 < 224 && t176 < 224) return 0;
    const double c256 = (double)((t256 + 255) / 256), c176 = 0.85 * (double)((t176 + 255) / 256);
    return c176 < c256 ? 176 : 256;
}
// a residual GEMM (64-column tiles of a 1024-wide stream) with the LayerNorm that follows it fused behind the epilogue (GemmArgs::ln ...):
// one resident wave of workgroups, 16 column tiles per row slab; hipErrorInvalidValue otherwise.  Experiments builds only (measured
// slower than the two launches it replaces).
hipError_t f5_launch_gemm5_generic_lne(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st);
// conv5.h: sliding-window Conv1d (the window of a 256-row tile once in LDS, taps served from it); hipErrorInvalidValue = shape not covered
hipError_t f5_launch_conv5(int prec, const GemmArgs& a, int n_pad, hipStream_t st);
// attn3.h: flash attention forward, 256 queries per workgroup
hipError_t f5_launch_attn3(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st);
void f5_set_attn_shape_invariant(int on);
hipError_t f5_launch_attn5(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st);   // attn5.h: ping-pong kernel (single key range)
// experiments/attn4.h (unequal-height waves, 16 x 16 x 32 MFMA; attn3 unless built with -DF5HIP_EXPERIMENTS)
hipError_t f5_launch_attn4(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st);

// gemm5 tile choice: fewest operand bytes per CU over the whole launch = rounds on the 256 CUs x (BM + BN); ties -> the larger tile.
// rb == 0: no instantiated tile divides n_pad.
struct Gemm5Choice { int rb, cb; };
static inline Gemm5Choice gemm5_choose(int m_rows, int n_pad) {
    static const int rbs[2] = {11, 8}, cbs[3] = {4, 8, 12};
    Gemm5Choice best = {0, 0};
    double best_cost = 1e30;
    for (int rb : rbs)
        for (int cb : cbs) {
            if (n_pad % (cb * 16)) continue;
            const long long tiles = (long long)((m_rows + rb * 16 - 1) / (rb * 16)) * (n_pad / (cb * 16));
            const double rounds = (double)((tiles + 255) / 256);
            const double cost = rounds * (rb * 16 + cb * 16);
            if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && rb * cb > best.rb * best.cb)) { best_cost = cost; best = {rb, cb}; }
        }
    return best;
}
