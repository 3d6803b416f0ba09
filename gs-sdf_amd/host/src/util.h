// util.h — glue between libtorch tensors and the C ABI (include/gsdf_hip.h).
#pragma once
#include <c10/hip/HIPStream.h>
#include <torch/torch.h>

#include "gsdf_hip.h"

namespace gsdf_host {

inline gsdf_stream_t cur_stream() { return (gsdf_stream_t)c10::hip::getCurrentHIPStream().stream(); }

inline void check(int rc, const char *what) { TORCH_CHECK(rc == GSDF_OK, what, " failed (", rc, "): ", gsdf_last_error()); }

// contiguous fp32 device tensor (copies only when the caller handed a strided view)
inline torch::Tensor f32c(const torch::Tensor &t, const char *name) {
  TORCH_CHECK(t.defined(), name, ": undefined tensor");
  TORCH_CHECK(t.is_cuda(), name, ": expected a device tensor (the HIP path has no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, ": expected float32");
  return t.contiguous();
}
inline const float *fp(const torch::Tensor &t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }
inline float *fpm(torch::Tensor &t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

inline torch::Tensor empty_like_opts(const torch::Tensor &ref, at::IntArrayRef shape, torch::ScalarType dt) {
  return torch::empty(shape, ref.options().dtype(dt).requires_grad(false));
}
inline torch::Tensor zeros_like_opts(const torch::Tensor &ref, at::IntArrayRef shape, torch::ScalarType dt) {
  return torch::zeros(shape, ref.options().dtype(dt).requires_grad(false));
}
inline int64_t read_i64(const torch::Tensor &dev_scalar) { return dev_scalar.item<int64_t>(); }  // the one host sync

}  // namespace gsdf_host
