// kaolin_wisp_cpp/octree_as/octree_as.h — drop-in for the header the reference includes at include/neural_net/sub_map.h:3.
// `OctreeAS` as the reference uses it: built by from_quantized_points (sub_map.cpp:33-34), `query(xyz, level).pidx`
// (sub_map.cpp:79, local_map.cpp:514), `raymarch(origin, dir, "voxel", n)` -> {ridx, samples, depth_samples}
// (local_map.cpp:467-476), `get_quantized_points()` (neural_mapping.cpp:755).  Backed by the bit pyramid of
// include/gsdf_hip.h section A1 (csrc/occupancy.hip); `pidx` is 0 for an occupied cell and -1 otherwise — the reference
// only ever tests `pidx > -1`.
#pragma once
#include <string>

#include <torch/torch.h>

struct OctreeQueryResults {
  torch::Tensor pidx;  // int64 [n]
};
struct OctreeRaymarchResults {
  torch::Tensor ridx;           // int64 [S]   ray index of every sample
  torch::Tensor samples;        // [S,3]       in the [-1,1]^3 frame
  torch::Tensor depth_samples;  // [S,1]       ray parameter t of origin + t*dir
};

class OctreeAS {
 public:
  OctreeAS(int level, torch::Tensor grid) : max_level_(level), grid_(std::move(grid)) {}
  OctreeQueryResults query(const torch::Tensor &xyz_m1p1, int level = -1) const;
  OctreeRaymarchResults raymarch(const torch::Tensor &origins_m1p1, const torch::Tensor &dirs,
                                 const std::string &raymarch_type, int num_samples) const;
  torch::Tensor get_quantized_points() const;  // int16 [V,3]
  int max_level_;
  torch::Tensor grid_;  // int32 words of the bit pyramid (gsdf_occ_bytes(level) bytes)
};

// Ownership passes to the caller, which wraps it in a shared_ptr (sub_map.cpp:33-34).
OctreeAS *from_quantized_points(const torch::Tensor &qpts, int level);
