// probe: what v_permlane16_swap / v_permlane32_swap deliver (diagnostics)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
    unsigned c = threadIdx.x;
    auto t = __builtin_amdgcn_permlane16_swap(c, c, false, false);
    out[256 + threadIdx.x] = t[0]; out[320 + threadIdx.x] = t[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1", "p16 same r0", "p16 same r1"};
    for (int i = 0; i < 6; i++) { printf("%-12s:", names[i]); for (int l = 0; l < 64; l += 4) printf(" %3u", h[i * 64 + l]); printf("\n"); }
    return 0;
}
