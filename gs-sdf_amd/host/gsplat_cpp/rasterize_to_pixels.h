// gsplat_cpp/rasterize_to_pixels.h — drop-in for the reference's submodule header
// (neural_gaussian.cpp:6); call site neural_gaussian.cpp:215-223, gradients of the leaf tensors `densify`
// and `means2d_absgrad` are read at :626-633.
#pragma once
#include <torch/torch.h>

#include <tuple>

// -> render_colors [C,H,W,3], render_depths [C,H,W,1], render_alphas [C,H,W,1], render_normals [C,H,W,3],
//    render_distort [C,H,W,1] (zeros: distloss is never enabled by the reference), render_median [C,H,W,1],
//    visibilities [M,1]
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_to_pixels_2dgs(const torch::Tensor &means2d, const torch::Tensor &ray_transforms, const torch::Tensor &colors,
                         const torch::Tensor &opacities, const torch::Tensor &normals, const torch::Tensor &densify,
                         int width, int height, int tile_size, const torch::Tensor &isect_offsets,
                         const torch::Tensor &flatten_ids, at::optional<torch::Tensor> backgrounds,
                         at::optional<torch::Tensor> masks, bool packed, const torch::Tensor &means2d_absgrad,
                         bool distloss);
