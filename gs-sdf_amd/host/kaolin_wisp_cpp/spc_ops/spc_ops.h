// kaolin_wisp_cpp/spc_ops/spc_ops.h — drop-in for the header the reference includes at
// include/neural_net/sub_map.cpp:2, local_map.cpp:2, neural_mapping.cpp:19, utils/utils.cpp:2 (the submodule is absent).
// Only the functions the reference calls: quantize_points (sub_map.cpp:26), points_to_neighbors (:31, local_map.cpp:287,411),
// points_to_corners (utils.cpp:790), quantized_points_to_fpoints (neural_mapping.cpp:757-758).  kaolin's conventions:
// the cube is [-1,1]^3, q = clamp(floor(2^level (x+1)/2), 0, 2^level-1) as int16 (DESIGN.md SPEC A.9).
#pragma once
#include <torch/torch.h>

namespace spc_ops {

inline torch::Tensor quantize_points(const torch::Tensor &x, int level) {
  const double res = (double)(1 << level);
  return torch::floor(res * (x + 1.0) / 2.0).clamp(0, res - 1).to(torch::kInt16);
}

namespace detail {
inline torch::Tensor offsets(const torch::Tensor &like, int lo, int hi) {  // [(hi-lo)^3, 3], x fastest
  auto r = torch::arange(lo, hi, like.options());
  auto g = torch::meshgrid({r, r, r}, "ij");
  return torch::stack({g[2].reshape(-1), g[1].reshape(-1), g[0].reshape(-1)}, -1);
}
}  // namespace detail

// [n,3] -> [n,27,3]: the 3x3x3 neighbourhood, NOT clamped (the caller clamps, sub_map.cpp:31)
inline torch::Tensor points_to_neighbors(const torch::Tensor &qpts) {
  return qpts.unsqueeze(1) + detail::offsets(qpts, -1, 2).unsqueeze(0);
}
// [n,3] -> [n,8,3]: the corners of each voxel
inline torch::Tensor points_to_corners(const torch::Tensor &qpts) {
  return qpts.unsqueeze(1) + detail::offsets(qpts, 0, 2).unsqueeze(0);
}
// voxel coordinates -> minimum corner of the voxel in [-1,1]
inline torch::Tensor quantized_points_to_fpoints(const torch::Tensor &qpts, int level) {
  return qpts.to(torch::kFloat32) * (2.0 / (double)(1 << level)) - 1.0;
}

}  // namespace spc_ops
