// kaolin/csrc/render/spc/raytrace.h — the one kaolin symbol the reference's local_map.cpp includes this header for
// (include/neural_net/local_map.cpp:11; used at :480 inside a commented-out block): marks the first element of every run
// of equal values in a sorted pack index.
#pragma once
#include <torch/torch.h>

namespace kaolin {
inline at::Tensor mark_pack_boundaries_cuda(at::Tensor pack_ids) {
  auto flat = pack_ids.reshape(-1);
  if (flat.numel() == 0) return torch::zeros({0}, flat.options().dtype(torch::kBool));
  auto first = torch::ones({1}, flat.options().dtype(torch::kBool));
  return torch::cat({first, flat.slice(0, 1) != flat.slice(0, 0, flat.numel() - 1)});
}
}  // namespace kaolin
