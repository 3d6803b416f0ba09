// gsplat_cpp/rendering.h — drop-in for the reference's submodule header (neural_gaussian.cpp:9);
// call sites neural_gaussian.cpp:199-200 (get_view_colors) and :207-209 (tile_encode).
#pragma once
#include <torch/torch.h>

#include <tuple>

namespace gsplat_cpp {

// colors: SH coefficients [N,K,3] when sh_degree has a value (rgb = max(SH(dir).c + 0.5, 0)), else
// post-activation colours [N,D] which are gathered to the visible rows.  Differentiable w.r.t. colors, means.
torch::Tensor get_view_colors(const torch::Tensor &viewmats, const torch::Tensor &means, const torch::Tensor &radii,
                              const torch::Tensor &colors, const torch::Tensor &camera_ids,
                              const torch::Tensor &gaussian_ids, at::optional<int> sh_degree);

// -> tiles_per_gauss i32[M], flatten_ids i32[I], isect_offsets i32[C,tile_h,tile_w]   (non-differentiable)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>
tile_encode(int width, int height, int tile_size, const torch::Tensor &means2d, const torch::Tensor &radii,
            const torch::Tensor &depths, bool packed, int64_t C, const torch::Tensor &camera_ids,
            const torch::Tensor &gaussian_ids);

}  // namespace gsplat_cpp
