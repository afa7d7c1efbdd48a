// translation unit: gemm6 kernels (256 x 256 ping-pong tiles, fp16 operands; generic and QKV epilogues)
#include "gemm6.h"
#include "gemm_launch.h"

hipError_t f5_launch_gemm6(int epi, const GemmArgs& a, int n_pad, hipStream_t st) {
    return epi == EPI_QKV ? launch_gemm6_t<true, EPI_QKV>(a, n_pad, st) : launch_gemm6_t<true, EPI_GENERIC>(a, n_pad, st);
}
