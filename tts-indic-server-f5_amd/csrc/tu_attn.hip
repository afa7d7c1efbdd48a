// translation unit: attention forward (attn3.h)
#include "attn3.h"
#ifdef F5HIP_EXPERIMENTS
#include "experiments/attn4.h"
#endif
#include "gemm_launch.h"

// Query-tile height 32 NW with NW in {4, 6, 8}: fewest (rounds on the 256 CUs) x (work per workgroup); ties -> the larger tile (fewer
// re-reads of K / V).  The grid is sized for the longest sequence; workgroups past a shorter one's end exit at once.
hipError_t f5_launch_attn3(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st) {
    int best = 8;
    long long best_cost = -1;
    for (int nw : {8, 6, 4}) {
        const long long wgs = (long long)((max_len + 32 * nw - 1) / (32 * nw)) * heads * n_seq;
        const long long cost = ((wgs + 255) / 256) * nw;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nw; }
    }
    const dim3 grid((max_len + 32 * best - 1) / (32 * best), heads, n_seq);
    if (a.seq_kv2_row0) {   // two key ranges per (pseudo-)sequence: MMDiT joint attention
        if (!a.seq_kv_row0 || !a.seq_kv2_len) return hipErrorInvalidValue;
        if (best == 8) hipLaunchKernelGGL((attn3_fwd_kernel<8, true>), grid, dim3(512), 0, st, a);
        else if (best == 6) hipLaunchKernelGGL((attn3_fwd_kernel<6, true>), grid, dim3(384), 0, st, a);
        else hipLaunchKernelGGL((attn3_fwd_kernel<4, true>), grid, dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (best == 8) hipLaunchKernelGGL(attn3_fwd_kernel<8>, grid, dim3(512), 0, st, a);
    else if (best == 6) hipLaunchKernelGGL(attn3_fwd_kernel<6>, grid, dim3(384), 0, st, a);
    else hipLaunchKernelGGL(attn3_fwd_kernel<4>, grid, dim3(256), 0, st, a);
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
