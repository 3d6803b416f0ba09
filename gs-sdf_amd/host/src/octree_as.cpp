// octree_as.cpp — OctreeAS of kaolin_wisp_cpp over the C ABI (include/gsdf_hip.h section A1).
#include "kaolin_wisp_cpp/octree_as/octree_as.h"

#include "util.h"

using namespace gsdf_host;

OctreeAS *from_quantized_points(const torch::Tensor &qpts, int level) {
  torch::NoGradGuard ng;
  TORCH_CHECK(qpts.dim() == 2 && qpts.size(1) == 3, "from_quantized_points: expected [n,3]");
  TORCH_CHECK(qpts.is_cuda(), "from_quantized_points: expected a device tensor (the HIP path has no CPU fallback)");
  const size_t bytes = gsdf_occ_bytes(level);
  TORCH_CHECK(bytes > 0, "from_quantized_points: level ", level, " outside [1,12]");
  // voxel centres in the [-1,1] frame re-quantise to the same voxels
  torch::Tensor centres = ((qpts.to(torch::kFloat32) + 0.5) * (2.0 / (double)(1 << level)) - 1.0).contiguous();
  torch::Tensor grid = torch::empty({(int64_t)(bytes / 4)}, qpts.options().dtype(torch::kInt32));
  check(gsdf_occ_build(level, centres.size(0), fp(centres), 0, grid.data_ptr(), cur_stream()), "from_quantized_points");
  return new OctreeAS(level, grid);
}

OctreeQueryResults OctreeAS::query(const torch::Tensor &xyz_, int level) const {
  torch::NoGradGuard ng;
  TORCH_CHECK(xyz_.dim() == 2 && xyz_.size(1) == 3, "OctreeAS::query: expected [n,3]");
  torch::Tensor xyz = f32c(xyz_.detach(), "xyz");
  torch::Tensor mask = empty_like_opts(xyz, {xyz.size(0)}, torch::kUInt8);
  check(gsdf_occ_query(max_level_, level, xyz.size(0), fp(xyz), grid_.data_ptr(),
                       xyz.size(0) ? mask.data_ptr<uint8_t>() : nullptr, cur_stream()), "OctreeAS::query");
  return {mask.to(torch::kInt64) - 1};
}

torch::Tensor OctreeAS::get_quantized_points() const {
  torch::NoGradGuard ng;
  const int64_t n_words = std::max<int64_t>(1, ((int64_t)1 << (3 * max_level_)) / 32);
  torch::Tensor counts = torch::empty({n_words}, grid_.options().dtype(torch::kInt32));
  check(gsdf_occ_voxel_counts(max_level_, grid_.data_ptr(), counts.data_ptr<int32_t>(), cur_stream()), "get_quantized_points");
  torch::Tensor incl = torch::cumsum(counts, 0, torch::kInt64);
  const int64_t total = read_i64(incl[-1]);
  torch::Tensor offs = (incl - counts).contiguous();
  torch::Tensor vox = torch::empty({total, 3}, grid_.options().dtype(torch::kInt16));
  if (total > 0)
    check(gsdf_occ_voxel_list(max_level_, grid_.data_ptr(), offs.data_ptr<int64_t>(), vox.data_ptr<int16_t>(), cur_stream()),
          "get_quantized_points");
  return vox;
}

OctreeRaymarchResults OctreeAS::raymarch(const torch::Tensor &origins_, const torch::Tensor &dirs_,
                                         const std::string &raymarch_type, int num_samples) const {
  torch::NoGradGuard ng;
  TORCH_CHECK(raymarch_type == "voxel", "OctreeAS::raymarch: only the 'voxel' mode the reference uses is implemented");
  TORCH_CHECK(origins_.dim() == 2 && origins_.size(1) == 3 && dirs_.sizes() == origins_.sizes(), "OctreeAS::raymarch: expected [n,3] origins and dirs");
  torch::Tensor o = f32c(origins_.detach(), "origins"), d = f32c(dirs_.detach(), "dirs");
  const int64_t n = o.size(0);
  torch::Tensor counts = empty_like_opts(o, {n}, torch::kInt32);
  check(gsdf_occ_raymarch_count(max_level_, n, fp(o), fp(d), grid_.data_ptr(), n ? counts.data_ptr<int32_t>() : nullptr, cur_stream()),
        "OctreeAS::raymarch (count)");
  int64_t total = 0;
  torch::Tensor offs;
  if (n > 0) {
    torch::Tensor incl = torch::cumsum(counts, 0, torch::kInt64);
    total = read_i64(incl[-1]) * num_samples;
    offs = (incl - counts).contiguous();
  }
  torch::Tensor ridx = empty_like_opts(o, {total}, torch::kInt32);
  torch::Tensor samples = empty_like_opts(o, {total, 3}, torch::kFloat32), depth = empty_like_opts(o, {total, 1}, torch::kFloat32);
  if (total > 0)
    check(gsdf_occ_raymarch_fill(max_level_, n, fp(o), fp(d), grid_.data_ptr(), offs.data_ptr<int64_t>(), num_samples,
                                 ridx.data_ptr<int32_t>(), fpm(samples), fpm(depth), cur_stream()), "OctreeAS::raymarch (fill)");
  return {ridx.to(torch::kInt64), samples, depth};
}
