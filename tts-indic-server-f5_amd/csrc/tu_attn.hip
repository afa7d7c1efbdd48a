// translation unit: attention forward (attn3.h)
#include "attn3.h"
#include "gemm_launch.h"

hipError_t f5_launch_attn3(const AttnArgs& a, int max_len, int heads, int n_seq, hipStream_t st) {
    hipLaunchKernelGGL(attn3_fwd_kernel, dim3((max_len + 255) / 256, heads, n_seq), dim3(512), 0, st, a);
    return hipGetLastError();
}
