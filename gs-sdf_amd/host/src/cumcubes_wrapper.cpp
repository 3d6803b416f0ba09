// cumcubes_wrapper.cpp — mc::marching_cubes_wrapper over the C ABI (include/gsdf_hip.h section M1).
#include "cumcubes/cumcubes_wrapper.h"

#include "util.h"

using namespace gsdf_host;

std::vector<torch::Tensor> mc::marching_cubes_wrapper(const torch::Tensor &density_grid, const float thresh,
                                                      const float *lower, const float *upper) {
  torch::NoGradGuard ng;
  TORCH_CHECK(density_grid.dim() == 3, "marching_cubes: expected a [X,Y,Z] grid");
  torch::Tensor g = f32c(density_grid.detach(), "density_grid");
  const int rx = (int)g.size(0), ry = (int)g.size(1), rz = (int)g.size(2);
  const int64_t n = g.numel();
  torch::Tensor n_vert = empty_like_opts(g, {n}, torch::kInt32), n_tri = empty_like_opts(g, {n}, torch::kInt32);
  // the reference's own triangle table (utils.cuh:31-289): the mesh the reference's mesher would emit, cell by cell
  check(gsdf_mc_count(rx, ry, rz, GSDF_MC_TABLE_REFERENCE, fp(g), thresh, n_vert.data_ptr<int32_t>(), n_tri.data_ptr<int32_t>(), cur_stream()),
        "marching_cubes (count)");
  torch::Tensor v_incl = torch::cumsum(n_vert, 0, torch::kInt64), t_incl = torch::cumsum(n_tri, 0, torch::kInt64);
  torch::Tensor totals = torch::stack({v_incl[-1], t_incl[-1]}).cpu();   // the one host sync
  const int64_t V = totals[0].item<int64_t>(), F = totals[1].item<int64_t>();
  torch::Tensor vertices = empty_like_opts(g, {V, 3}, torch::kFloat32), faces = empty_like_opts(g, {F, 3}, torch::kInt32);
  if (V > 0) {
    torch::Tensor v_off = (v_incl - n_vert).contiguous(), t_off = (t_incl - n_tri).contiguous();
    check(gsdf_mc_emit(rx, ry, rz, GSDF_MC_TABLE_REFERENCE, fp(g), thresh, v_off.data_ptr<int64_t>(), t_off.data_ptr<int64_t>(), lower, upper,
                       fpm(vertices), F > 0 ? faces.data_ptr<int32_t>() : nullptr, cur_stream()),
          "marching_cubes (emit)");
  }
  return {vertices, faces};
}
