// translation unit: gemm6 kernels (ping-pong tiles of 256 / 176 rows x 256 columns, fp16 operands; generic and QKV epilogues)
#include "gemm6.h"
#include "gemm_launch.h"

hipError_t f5_launch_gemm6(int epi, int rows, const GemmArgs& a, int n_pad, hipStream_t st) {
    if (rows == 176) return epi == EPI_QKV ? launch_gemm6_t<true, EPI_QKV, 6>(a, n_pad, st) : launch_gemm6_t<true, EPI_GENERIC, 6>(a, n_pad, st);
    if (rows == 256) return epi == EPI_QKV ? launch_gemm6_t<true, EPI_QKV, 8>(a, n_pad, st) : launch_gemm6_t<true, EPI_GENERIC, 8>(a, n_pad, st);
    return hipErrorInvalidValue;
}
