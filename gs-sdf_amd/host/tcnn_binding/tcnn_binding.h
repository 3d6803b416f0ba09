// tcnn_binding/tcnn_binding.h — drop-in for the reference's submodule header
// (/root/reference/include/neural_net/encoding_map.h:4, encodings/encodings.h:5, local_map.cpp:44-55).
// `TCNNEncoding`: multiresolution hash grid (tiny-cuda-nn "Grid"/"Hash"/"Linear") with first and second order
// autograd; `TCNNNetwork`: FullyFusedMLP (ReLU, no output activation) on the MFMA pipes, first and second order.
// `params_` is a plain fp32 leaf tensor the caller registers as an nn parameter (local_map.cpp:53-54,73-75).
#pragma once
#include <torch/torch.h>

#include <nlohmann/json.hpp>
#include <string>
#include <vector>

struct TCNNEncoding {
  TCNNEncoding() = default;
  TCNNEncoding(int n_input_dims, const nlohmann::json &config, const std::string &name) {
    init_encoding(n_input_dims, config, name);
  }
  virtual ~TCNNEncoding() = default;

  void init_encoding(int n_input_dims, const nlohmann::json &config, const std::string &name);
  torch::Tensor forward(const torch::Tensor &x);
  // NOT in the reference's interface (used by the gsdf_extras edits of INTEGRATION.md section 5): x = n base rows followed by the
  // 6 blocks of n central-difference rows that gsdf_extras::query_points(..., with_stencil = true) writes (LocalMap::get_gradient's
  // numerical branch, local_map.cpp:110-150), delta_unit = the stencil offset in unit-cube coordinates.  Same features and
  // gradients as forward(x); the 7 rows of a group are gathered together and the table-gradient scatter merges the rows that
  // share a grid cell.
  torch::Tensor forward_stencil(const torch::Tensor &x, int64_t n_groups, double delta_unit);
  virtual size_t get_out_dim() const {   // (the reference's SHEncoding overrides it with levels^2, encodings.h:26: the same value)
    return otype_ == "SphericalHarmonics" ? (size_t)sh_degree_ * sh_degree_ : (size_t)n_levels_ * n_feat_;
  }

  torch::Tensor params_;
  std::string name_;
  std::string otype_;
  int n_input_dims_ = 3, n_levels_ = 16, n_feat_ = 2, log2_hashmap_ = 19, base_res_ = 16, sh_degree_ = 0;
  float per_level_scale_ = 2.0f;
  std::vector<int64_t> offsets_;  // entry offsets per level
};

// First AND second order autograd (the analytic eikonal term of the reference's default configuration differentiates
// d sdf / d features again, local_map.cpp:151-172).  Config key "bias": true (NOT in tiny-cuda-nn; default false) adds per-layer
// biases `biases_` — the topology of the reference's torch::nn::Sequential decoder (local_map.cpp:29-42) on the fused kernels.
struct TCNNNetwork {
  TCNNNetwork(int n_input_dims, int n_output_dims, const nlohmann::json &config, const std::string &name);
  torch::Tensor forward(const torch::Tensor &x);

  torch::Tensor params_;
  torch::Tensor biases_;   // undefined for the bias-free FullyFusedMLP
  std::string name_;
  std::vector<int> dims_;
};
