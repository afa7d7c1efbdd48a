// translation unit: gemm5 kernels with the generic epilogue (fp16 operands)
#include "gemm5.h"
#include "gemm_launch.h"

hipError_t f5_launch_gemm5_generic(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st) {
    return launch_gemm5<true, EPI_GENERIC>(a, rb, cb, n_pad, st);
}
#ifdef F5HIP_EXPERIMENTS
hipError_t f5_launch_gemm5_generic_lne(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st) {
    return launch_gemm5_lne<true>(a, rb, cb, n_pad, st);
}
#endif
