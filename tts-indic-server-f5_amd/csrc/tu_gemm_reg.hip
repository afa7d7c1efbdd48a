// translation unit: register-staged gemm.h kernels (state GEMMs, implicit-GEMM convolutions, many-tile shapes)
#include "gemm.h"
#include "gemm_launch.h"

hipError_t f5_launch_gemm_reg(int prec, int bn, bool conv, int epi, const GemmArgs& a, int m_pad, int n_pad, hipStream_t st) {
    if (prec == 3) {
        if (conv && bn == 64) return launch_gemm_t<3, 64, true, EPI_GENERIC>(a, m_pad, n_pad, st);
        if (conv) return launch_gemm_t<3, 128, true, EPI_GENERIC>(a, m_pad, n_pad, st);
        if (bn != 128) return hipErrorInvalidValue;
        return epi == EPI_QKV ? launch_gemm_t<3, 128, false, EPI_QKV>(a, m_pad, n_pad, st) : launch_gemm_t<3, 128, false, EPI_GENERIC>(a, m_pad, n_pad, st);
    }
    if (prec == 2) {
        if (epi == EPI_QKV) return launch_gemm_t<2, 128, false, EPI_QKV>(a, m_pad, n_pad, st);
        if (conv && bn == 64) return launch_gemm_t<2, 64, true, EPI_GENERIC>(a, m_pad, n_pad, st);
        if (conv) return launch_gemm_t<2, 128, true, EPI_GENERIC>(a, m_pad, n_pad, st);
        if (bn == 64) return launch_gemm_t<2, 64, false, EPI_GENERIC>(a, m_pad, n_pad, st);
        return launch_gemm_t<2, 128, false, EPI_GENERIC>(a, m_pad, n_pad, st);
    }
    if (epi == EPI_QKV) return launch_gemm_t<1, 128, false, EPI_QKV>(a, m_pad, n_pad, st);
    if (conv && bn == 64) return launch_gemm_t<1, 64, true, EPI_GENERIC>(a, m_pad, n_pad, st);
    if (conv) return launch_gemm_t<1, 128, true, EPI_GENERIC>(a, m_pad, n_pad, st);
    if (bn == 64) return launch_gemm_t<1, 64, false, EPI_GENERIC>(a, m_pad, n_pad, st);
    return launch_gemm_t<1, 128, false, EPI_GENERIC>(a, m_pad, n_pad, st);
}
