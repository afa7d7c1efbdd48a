// translation unit: attention forward (attn3.h)
#include "attn3.h"
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
    if (best == 8) hipLaunchKernelGGL(attn3_fwd_kernel<8>, grid, dim3(512), 0, st, a);
    else if (best == 6) hipLaunchKernelGGL(attn3_fwd_kernel<6>, grid, dim3(384), 0, st, a);
    else hipLaunchKernelGGL(attn3_fwd_kernel<4>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}
