// translation unit: gemm3 kernels (warp-specialised 128 x {128, 256} x 32): bf16 / split-bf16 operands, fp16 when K % 64 != 0
#include "gemm3.h"
#include "gemm_launch.h"

hipError_t f5_launch_gemm3(int prec, int epi, int bn, const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    const bool qkv = epi == EPI_QKV;
    if (bn == 256) {
        if (prec == 3) return qkv ? launch_gemm3_t<3, EPI_QKV, 0, 256>(a, m_pad, n_pad, st) : launch_gemm3_t<3, EPI_GENERIC, 0, 256>(a, m_pad, n_pad, st);
        return hipErrorInvalidValue;
    }
    if (prec == 3) return qkv ? launch_gemm3_t<3, EPI_QKV>(a, m_pad, n_pad, st) : launch_gemm3_t<3, EPI_GENERIC>(a, m_pad, n_pad, st);
    if (prec == 2) return qkv ? launch_gemm3_t<2, EPI_QKV>(a, m_pad, n_pad, st) : launch_gemm3_t<2, EPI_GENERIC>(a, m_pad, n_pad, st);
    return qkv ? launch_gemm3_t<1, EPI_QKV>(a, m_pad, n_pad, st) : launch_gemm3_t<1, EPI_GENERIC>(a, m_pad, n_pad, st);
}
