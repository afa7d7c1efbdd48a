// translation unit: gemm5 kernels with the QKV epilogue (fp16 operands)
#include "gemm5.h"
#include "gemm_launch.h"

hipError_t f5_launch_gemm5_qkv(const GemmArgs& a, int rb, int cb, int n_pad, hipStream_t st) {
    return launch_gemm5<true, EPI_QKV>(a, rb, cb, n_pad, st);
}
