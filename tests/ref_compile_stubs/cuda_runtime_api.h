// stand-in for <cuda_runtime_api.h> for the compile of the reference's host code (tests/test_reference_compiles_against_boundary.py,
// oracle/ref_link/): the three runtime calls utils/utils.cpp makes in its memory-report and point-cloud colour helpers, none of which the
// hot path reaches.  They report failure / throw instead of doing anything.
#pragma once
#include <cstddef>
#include <stdexcept>
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
inline cudaError_t cudaMemGetInfo(size_t *free_bytes, size_t *total_bytes) { *free_bytes = *total_bytes = 0; return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t) { return "cuda runtime stand-in"; }
inline cudaError_t cudaMemcpy(void *, const void *, size_t, cudaMemcpyKind) { throw std::runtime_error("cudaMemcpy: stand-in, not on the hot path"); }
// utils/utils.cpp:629 calls torch::linalg::eigh, which <torch/torch.h> provided in the libtorch the reference was written against; this image's
// libtorch headers ship without torch/linalg.h, so the one wrapper is supplied here (this header is the first stand-in utils.h includes after torch)
#if !__has_include(<torch/linalg.h>)
#include <ATen/ATen.h>
#include <tuple>
namespace torch {
namespace linalg {
inline std::tuple<at::Tensor, at::Tensor> eigh(const at::Tensor &a, c10::string_view uplo) { return at::linalg_eigh(a, uplo); }
}  // namespace linalg
}  // namespace torch
#endif

// neural_mapping.cpp calls c10::cuda::CUDACachingAllocator::emptyCache() before its memory reports (":301, :361, ..."); a CUDA libtorch brings
// the declaration in through <torch/torch.h>, this image's does not.  Inert here: it only affects the reported memory figures.
namespace c10 {
namespace cuda {
namespace CUDACachingAllocator {
inline void emptyCache() {}
}  // namespace CUDACachingAllocator
}  // namespace cuda
}  // namespace c10
