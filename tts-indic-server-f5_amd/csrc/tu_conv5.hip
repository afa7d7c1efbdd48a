// translation unit: sliding-window convolution kernels (conv5.h)
#include "conv5.h"
#include "gemm_launch.h"

hipError_t f5_launch_conv5(int prec, const GemmArgs& a, int n_pad, hipStream_t st) { return launch_conv5(prec, a, n_pad, st); }
