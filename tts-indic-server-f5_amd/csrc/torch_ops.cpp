// TORCH_LIBRARY registration of the hot path (north_star: "host Python calling HIP through PyTorch-ROCm custom ops"): thin operators over the
// C ABI of libf5hip (include/f5hip.h) -- torch tensors in, torch tensors out, the current HIP stream of the tensors' device, errors as
// c10::Error.  Handles are the opaque pointers the *_create functions of the ABI return, carried as int64.  Host C++ only (no kernels here):
// built by build.py into csrc/libf5hip_torch.so next to libf5hip.so, loaded with torch.ops.load_library (tts_indic_server_f5_amd/torch_ops.py).
//   torch.ops.f5hip.cfm_sample(handle, dur, kv_len?, cond, cond_mask, text, y0, t_grid, cfg_strength) -> Tensor   F/model/cfm.py:160-204
//   torch.ops.f5hip.vocos_decode(handle, mel) -> Tensor                                                          F/infer/utils_infer.py:472
//   torch.ops.f5hip.bigvgan_forward(handle, mel, total_upsample) -> Tensor                                       F/infer/utils_infer.py:474
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/f5hip.h"

namespace {

void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_dev_f32(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), "f5hip: ", name, " must be a contiguous fp32 tensor on the HIP device");
}
void check_host(const at::Tensor& t, at::ScalarType ty, const char* name) {
    TORCH_CHECK(!t.is_cuda() && t.scalar_type() == ty && t.is_contiguous(), "f5hip: ", name, " must be a contiguous host tensor of the documented dtype");
}

// The ODE loop of CFM.sample over packed rows: dur [b] int32 host (rows laid out per item), kv_len [b] int32 host or None (valid frames per item:
// the reference's padded-batch semantics), cond [sum(dur), mel] fp32 device, cond_mask [sum(dur)] uint8 host, text [b, nt] int32 host (-1 padded),
// y0 [sum(dur), mel] fp32 device, t_grid [steps + 1] fp32 host.  Returns the sampled mel rows [sum(dur), mel].
at::Tensor cfm_sample(int64_t handle, const at::Tensor& dur, const c10::optional<at::Tensor>& kv_len, const at::Tensor& cond, const at::Tensor& cond_mask,
                      const at::Tensor& text, const at::Tensor& y0, const at::Tensor& t_grid, double cfg_strength) {
    check_host(dur, at::kInt, "dur"); check_host(cond_mask, at::kByte, "cond_mask"); check_host(text, at::kInt, "text"); check_host(t_grid, at::kFloat, "t_grid");
    check_dev_f32(cond, "cond"); check_dev_f32(y0, "y0");
    TORCH_CHECK(text.dim() == 2 && text.size(0) == dur.numel() && t_grid.numel() >= 2 && cond.sizes() == y0.sizes(), "f5hip::cfm_sample: shapes");
    if (kv_len.has_value()) check_host(*kv_len, at::kInt, "kv_len");
    at::Tensor out = at::empty_like(y0);
    const int rc = f5hip_cfm_sample_masked((f5hip_dit*)handle, (int32_t)dur.numel(), dur.data_ptr<int32_t>(), kv_len.has_value() ? kv_len->data_ptr<int32_t>() : nullptr,
                                           cond.data_ptr<float>(), cond_mask.data_ptr<uint8_t>(), text.data_ptr<int32_t>(), (int32_t)text.size(1), y0.data_ptr<float>(),
                                           t_grid.data_ptr<float>(), (int32_t)t_grid.numel() - 1, (float)cfg_strength, out.data_ptr<float>(), stream_of(y0));
    TORCH_CHECK(rc == 0, "f5hip_cfm_sample: ", f5hip_last_error());
    return out;
}

at::Tensor vocos_decode(int64_t handle, const at::Tensor& mel, int64_t hop_length) {
    check_dev_f32(mel, "mel");
    TORCH_CHECK(mel.dim() == 3, "f5hip::vocos_decode: mel [b, 100, T]");
    at::Tensor wave = at::empty({mel.size(0), hop_length * (mel.size(2) - 1)}, mel.options());
    const int rc = f5hip_vocos_decode((f5hip_vocos*)handle, (int32_t)mel.size(0), (int32_t)mel.size(2), mel.data_ptr<float>(), wave.data_ptr<float>(), stream_of(mel));
    TORCH_CHECK(rc == 0, "f5hip_vocos_decode: ", f5hip_last_error());
    return wave;
}

at::Tensor bigvgan_forward(int64_t handle, const at::Tensor& mel, int64_t total_upsample) {
    check_dev_f32(mel, "mel");
    TORCH_CHECK(mel.dim() == 3, "f5hip::bigvgan_forward: mel [b, 100, T]");
    at::Tensor wave = at::empty({mel.size(0), 1, total_upsample * mel.size(2)}, mel.options());
    const int rc = f5hip_bigvgan_forward((f5hip_bigvgan*)handle, (int32_t)mel.size(0), (int32_t)mel.size(2), mel.data_ptr<float>(), wave.data_ptr<float>(), stream_of(mel));
    TORCH_CHECK(rc == 0, "f5hip_bigvgan_forward: ", f5hip_last_error());
    return wave;
}

}   // namespace

TORCH_LIBRARY(f5hip, m) {
    m.def("cfm_sample(int handle, Tensor dur, Tensor? kv_len, Tensor cond, Tensor cond_mask, Tensor text, Tensor y0, Tensor t_grid, float cfg_strength) -> Tensor", &cfm_sample);
    m.def("vocos_decode(int handle, Tensor mel, int hop_length) -> Tensor", &vocos_decode);
    m.def("bigvgan_forward(int handle, Tensor mel, int total_upsample) -> Tensor", &bigvgan_forward);
}
