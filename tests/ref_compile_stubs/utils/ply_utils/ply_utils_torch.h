// stand-in for the reference's (un-vendored) ply_utils submodule + tinyply as the hot-path host code uses them
// (neural_gaussian.cpp:928-1188 gs.ply export / load; local_map.cpp).  Syntax-only test infrastructure.
#pragma once
#include <torch/torch.h>

#include <istream>
#include <memory>
#include <ostream>
#include <string>
#include <vector>
namespace tinyply {
enum class Type { INVALID, INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 };
struct Buffer { uint8_t *get() { return nullptr; } };
struct PlyData { Type t = Type::INVALID; Buffer buffer; size_t count = 0; bool isList = false; };
struct PlyFile {
  void add_properties_to_element(const std::string &, const std::vector<std::string> &, Type, size_t, uint8_t *, Type, size_t) {}
  void write(std::ostream &, bool) {}
  bool parse_header(std::istream &) { return true; }
  std::shared_ptr<PlyData> request_properties_from_element(const std::string &, const std::vector<std::string> &, uint32_t = 0) { return std::make_shared<PlyData>(); }
  void read(std::istream &) {}
  std::vector<std::string> &get_comments() { static std::vector<std::string> c; return c; }
};
}  // namespace tinyply
namespace ply_utils {
inline tinyply::Type torch_type_to_ply_type(c10::ScalarType) { return tinyply::Type::FLOAT32; }
inline bool export_to_ply(const std::string &, const torch::Tensor &, const torch::Tensor & = torch::Tensor(), const torch::Tensor & = torch::Tensor()) { return true; }
inline bool read_ply_file_to_map_tensor(const std::string &, std::map<std::string, torch::Tensor> &, const torch::Device & = torch::kCPU) { return false; }
inline bool read_ply_file_to_tensor(const std::string &, std::map<std::string, torch::Tensor> &, const torch::Device & = torch::kCPU) { return true; }
}  // namespace ply_utils
