// cumcubes_wrapper.h — the one symbol of the reference's in-tree mesher that lived in CUDA
// (/root/reference/include/mesher/cumcubes/src/cumcubes_kernel.cu:204-282, declared in include/cumcubes.hpp:15-19):
// mc::marching_cubes_wrapper.  The reference's own cumcubes.cpp (argument checks, save_mesh_as_ply) stays as it is and
// links against this definition; its header only needs `#include <cuda_runtime.h>` dropped.
#pragma once
#include <vector>

#include <torch/torch.h>

namespace mc {
// density_grid [X,Y,Z] float32 on the device -> {vertices [V,3] float32, faces [F,3] int32}
std::vector<torch::Tensor> marching_cubes_wrapper(const torch::Tensor &density_grid, const float thresh, const float *lower,
                                                  const float *upper);
}  // namespace mc
