// gsplat_cpp/fully_fused_projection.h — drop-in for the reference's (absent) submodule header of the same
// name, included at /root/reference/include/neural_gaussian/neural_gaussian.cpp:5 and called at :188-192.
// Global namespace, same argument order; differentiable w.r.t. means, quats, scales (torch::autograd::Function
// over the C ABI of include/gsdf_hip.h).  Only the reference's configuration is implemented: packed = true,
// sparse_grad = false (neural_gaussian.cpp:526-530); anything else raises like TORCH_CHECK.
#pragma once
#include <torch/torch.h>

#include <tuple>

// -> camera_ids i64[M], gaussian_ids i64[M], radii i32[M], means2d [M,2], depths [M], ray_transforms [M,3,3],
//    normals [M,3], samples [M,3], samples_weights [M,1]
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
fully_fused_projection_2dgs(const torch::Tensor &means, const torch::Tensor &quats, const torch::Tensor &scales,
                            const torch::Tensor &viewmats, const torch::Tensor &Ks, int width, int height,
                            float near_plane, float far_plane, float radius_clip, bool packed, bool sparse_grad);

namespace gsplat_cpp {
// `samples` / `samples_weights` (SPEC S-3).  stochastic: one sample on every visible splat's disc, a fresh seed per call derived from
// torch's default generator as a CUDA op would (what the reference's default k_center_reg = 0 consumes, neural_gaussian.cpp:258-264);
// otherwise splat centres with unit weights and no draw, for callers that replace the samples anyway (k_center_reg = 1).
//
// The mode is an ARGUMENT wherever the signature is ours: this overload, and next_sample_seed(stochastic) for callers of the C ABI
// (gsdf_model::rasterization_2dgs_sdf, gsdf_extras::JointIteration pass their own configuration; nothing they do depends on ambient state).
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
fully_fused_projection_2dgs(const torch::Tensor &means, const torch::Tensor &quats, const torch::Tensor &scales,
                            const torch::Tensor &viewmats, const torch::Tensor &Ks, int width, int height,
                            float near_plane, float far_plane, float radius_clip, bool packed, bool sparse_grad, bool stochastic_samples);
// the seed a projection in that mode uses (0 in centre mode; a draw from torch's default CPU generator otherwise)
uint64_t next_sample_seed(bool stochastic);

// The reference's 12-argument signature above has no slot for the mode, so for THAT entry point only it is ambient — per THREAD
// (thread_local, default stochastic = what the unmodified reference expects): two NeuralGS objects rendering from two threads
// (neural_gaussian.cpp:498 serialises per object, not per process) cannot disturb each other.
void set_sample_mode(bool stochastic);
bool get_sample_mode();
uint64_t next_sample_seed();   // in the calling thread's ambient mode
struct SampleModeGuard {
  explicit SampleModeGuard(bool stochastic) : prev_(get_sample_mode()) { set_sample_mode(stochastic); }
  ~SampleModeGuard() { set_sample_mode(prev_); }
  SampleModeGuard(const SampleModeGuard &) = delete;
  SampleModeGuard &operator=(const SampleModeGuard &) = delete;

 private:
  bool prev_;
};
}  // namespace gsplat_cpp
