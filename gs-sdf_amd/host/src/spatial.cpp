// spatial.cpp — distCUDA2 of simple-knn over the C ABI (reference call: neural_gaussian.cpp:314).
#include "spatial.h"

#include "util.h"

using namespace gsdf_host;

torch::Tensor distCUDA2(const torch::Tensor &points_) {
  torch::NoGradGuard ng;
  TORCH_CHECK(points_.dim() == 2 && points_.size(1) == 3, "distCUDA2: expected [N,3]");
  torch::Tensor points = f32c(points_.detach(), "points");
  const int64_t N = points.size(0);
  torch::Tensor out = empty_like_opts(points, {N}, torch::kFloat32);
  torch::Tensor ws = empty_like_opts(points, {(int64_t)gsdf_knn_ws_bytes(N)}, torch::kUInt8);
  check(gsdf_knn_mean_dist2(N, fp(points), fpm(out), ws.data_ptr(), cur_stream()), "distCUDA2");
  return out;
}
