// translation unit: attention forward (attn3.h)
#include <cstdlib>
#include "attn3.h"
#ifdef F5HIP_EXPERIMENTS
#include "experiments/attn4.h"
#include "experiments/attn5.h"
#endif
#include "gemm_launch.h"

// Which attn3 instance: the tile height 32 NW, NW in {4, 6, 8}, that needs the fewest (rounds on the 256 CUs) x (work per workgroup); ties -> the
// larger tile (fewer re-reads of K / V).  The grid is sized for the longest sequence; workgroups past a shorter one's end exit at once.
//   * 192-query tiles (NW = 6) run as the SIMD-balanced 8-wave kernel (attn3.h BAL): eight 240-register waves fill a CU, so it has the CU to
//     itself and takes the 9-stage ring.  f5hip_set_attention_shape_invariant(1) keeps the 6-wave form, whose arithmetic is that of every
//     other variant.
//   * ring depth otherwise: a tile takes 2-4 us from beyond L2 into LDS and is consumed in ~1 us, so 3 tiles in flight (5 stages, 80 KiB: two
//     workgroups per CU when the grid has more than one round) starve a LONE workgroup per CU -- 6- and 8-wave launches of at most 256
//     workgroups get 9 stages (144 KiB, 7 tiles in flight); at NW = 4 the deep ring measured slower (16.8 vs 14.3 us at 2 x 748 x 12 heads).
// -DF5HIP_EXPERIMENTS builds also carry QB = 2 (64 queries per wave, NW = 4, one wave per SIMD with the 512-register budget, every fragment read
// feeding two MFMAs; F5HIP_ATTN_QB=2): parity-tested and measured slower -- 38.8 us against 35.4 at C2, 225 against 186 at 16 x 1404
// (profiles/r02_attn_bench.txt): hipcc parks half of its score blocks in AGPRs (32 v_accvgpr_read per tile) and a lone in-order wave cannot
// cover its own waits, which two waves per SIMD do for each other -- and F5HIP_ATTN_BAL=0 (the 6-wave form, for A/B timing).
static int g_attn_shape_invariant = 0;   // f5hip_set_attention_shape_invariant: the default of launches whose AttnArgs::shape_invariant is -1
void f5_set_attn_shape_invariant(int on) { g_attn_shape_invariant = on != 0; }

template <bool SEG2>
static void attn3_launch(const AttnArgs& a, int best, bool deep, bool bal, dim3 grid, hipStream_t st) {
    if (bal) hipLaunchKernelGGL((attn3_fwd_kernel<8, SEG2, false, 9, 1, true>), grid, dim3(512), 0, st, a);
    else if (best == 8 && deep) hipLaunchKernelGGL((attn3_fwd_kernel<8, SEG2, false, 9>), grid, dim3(512), 0, st, a);
    else if (best == 8) hipLaunchKernelGGL((attn3_fwd_kernel<8, SEG2, false, 5>), grid, dim3(512), 0, st, a);
    else if (best == 6 && deep) hipLaunchKernelGGL((attn3_fwd_kernel<6, SEG2, false, 9>), grid, dim3(384), 0, st, a);
    else if (best == 6) hipLaunchKernelGGL((attn3_fwd_kernel<6, SEG2, false, 5>), grid, dim3(384), 0, st, a);
    else hipLaunchKernelGGL((attn3_fwd_kernel<4, SEG2, false, 5>), grid, dim3(256), 0, st, a);
}

// attn5 (experiments/attn5.h, -DF5HIP_EXPERIMENTS builds only): the ping-pong kernel, 256 queries per workgroup, single key range.  Measured
// 11-40 % slower than attn3 (profiles/r03_attn5_pingpong.txt); without the experiments flag this is attn3.
hipError_t f5_launch_attn3(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st);
hipError_t f5_launch_attn5(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st) {
#ifdef F5HIP_EXPERIMENTS
    if (a.seq_kv2_row0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((attn5_fwd_kernel<6>), dim3((max_len + 255) / 256, heads, n_seq), dim3(512), 0, st, a);
    return hipGetLastError();
#else
    return f5_launch_attn3(a, max_len, heads, n_seq, st);
#endif
}

hipError_t f5_launch_attn3(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st) {
    if (a.seq_kv2_row0 && (!a.seq_kv_row0 || !a.seq_kv2_len)) return hipErrorInvalidValue;   // two key ranges per (pseudo-)sequence: MMDiT joint attention
#ifdef F5HIP_EXPERIMENTS
    static const int use5 = getenv("F5HIP_ATTN5") ? atoi(getenv("F5HIP_ATTN5")) : 0;
    if (use5 && !a.seq_kv2_row0) return f5_launch_attn5(a, max_len, heads, n_seq, st);
#endif
    bool no_bal = false;
#ifdef F5HIP_EXPERIMENTS
    static const int force_qb = getenv("F5HIP_ATTN_QB") ? atoi(getenv("F5HIP_ATTN_QB")) : 0;
    if (force_qb == 2) {
        const dim3 grid((max_len + 255) / 256, heads, n_seq);
        if (a.seq_kv2_row0) hipLaunchKernelGGL((attn3_fwd_kernel<4, true, false, 9, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn3_fwd_kernel<4, false, false, 9, 2>), grid, dim3(256), 0, st, a);
        return hipGetLastError();
    }
    static const bool env_no_bal = getenv("F5HIP_ATTN_BAL") && atoi(getenv("F5HIP_ATTN_BAL")) == 0;
    no_bal = env_no_bal;
#endif
    int best = 8;
    long long best_cost = -1;
    for (int nw : {8, 6, 4}) {
        const long long wgs = (long long)((max_len + 32 * nw - 1) / (32 * nw)) * heads * n_seq;
        const long long cost = ((wgs + 255) / 256) * nw;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nw; }
    }
    static const int force_nw = getenv("F5HIP_ATTN_NW") ? atoi(getenv("F5HIP_ATTN_NW")) : 0;        // diagnostics (tools/attn_ab.py): 4 | 6 | 8 waves
    static const int force_deep = getenv("F5HIP_ATTN_DEEP") ? atoi(getenv("F5HIP_ATTN_DEEP")) : -1;   // 1: the 9-stage ring (one workgroup per CU)
    if (force_nw == 4 || force_nw == 6 || force_nw == 8) best = force_nw;
    const dim3 grid((max_len + 32 * best - 1) / (32 * best), heads, n_seq);
    const bool deep = best >= 6 && (force_deep >= 0 ? force_deep != 0 : (long long)grid.x * grid.y * grid.z <= 256);
    const bool invariant = a.shape_invariant < 0 ? g_attn_shape_invariant != 0 : a.shape_invariant != 0;
    const bool bal = best == 6 && !no_bal && !invariant;
    if (a.seq_kv2_row0) attn3_launch<true>(a, best, deep, bal, grid, st);
    else attn3_launch<false>(a, best, deep, bal, grid, st);
    return hipGetLastError();
}

// attn4 (experiments/attn4.h, -DF5HIP_EXPERIMENTS builds only): 16 x 16 x 32 MFMA blocks, waves 0-3 own 32 queries and waves 4-7 own 16
// (192 per workgroup, 48 per SIMD: exactly 256 workgroups at C2).  Measured and NOT used (profiles/r02_attn_bench.txt): per query the
// 16 x 16 formulation is ~1.3x slower than attn3's 32 x 32 x 16 one (C3 share with equal wave heights: 308 us against 237), the balanced
// C2 launch wins 8 % in isolation (41 us against 45) but 0.5 % end to end (98.3 ms against 98.85), and a kernel choice that depends on
// the batch shape breaks the bit-for-bit "batch of copies == single utterance" property.  Without the experiments flag this is attn3.
hipError_t f5_launch_attn4(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st) {
#ifdef F5HIP_EXPERIMENTS
    hipLaunchKernelGGL((attn4_fwd_kernel<2, 1>), dim3((max_len + 191) / 192, heads, n_seq), dim3(512), 0, st, a);
    return hipGetLastError();
#else
    return f5_launch_attn3(a, max_len, heads, n_seq, st);
#endif
}
