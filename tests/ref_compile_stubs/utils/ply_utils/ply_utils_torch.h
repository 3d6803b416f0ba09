// stand-in for the reference's (un-vendored) ply_utils submodule + tinyply as the hot-path host code uses them
// (neural_gaussian.cpp:928-1188 gs.ply export / load; local_map.cpp; neural_mapping.cpp).  Test infrastructure.
// tinyply::PlyFile is FUNCTIONAL for what NeuralGS::export_gs_to_ply / load_ply_to_gs ask of it — fixed-size scalar properties of one or more
// elements, binary little endian, groups of properties added / requested together — so that the reference's own exporter and loader run
// (oracle/ref_link, tests/test_reference_intree_pins.py): which properties, in which order, from which transposes is the reference's logic;
// the container is the public PLY format as tinyply writes it (header lines "property <type> <name>", rows interleaved in the order added).
// The ply_utils:: point-cloud helpers stay inert (they report failure / success without touching files).
#pragma once
#include <torch/torch.h>

#include <cstring>
#include <istream>
#include <map>
#include <memory>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace tinyply {
enum class Type { INVALID, INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 };
inline size_t type_size(Type t) {
  switch (t) {
    case Type::INT8: case Type::UINT8: return 1;
    case Type::INT16: case Type::UINT16: return 2;
    case Type::INT32: case Type::UINT32: case Type::FLOAT32: return 4;
    case Type::FLOAT64: return 8;
    default: return 0;
  }
}
inline const char *type_name(Type t) {
  static const char *n[] = {"invalid", "char", "uchar", "short", "ushort", "int", "uint", "float", "double"};
  return n[(int)t];
}
inline Type type_from_name(const std::string &s) {
  static const std::map<std::string, Type> m = {{"char", Type::INT8}, {"int8", Type::INT8}, {"uchar", Type::UINT8}, {"uint8", Type::UINT8},
                                                {"short", Type::INT16}, {"int16", Type::INT16}, {"ushort", Type::UINT16}, {"uint16", Type::UINT16},
                                                {"int", Type::INT32}, {"int32", Type::INT32}, {"uint", Type::UINT32}, {"uint32", Type::UINT32},
                                                {"float", Type::FLOAT32}, {"float32", Type::FLOAT32}, {"double", Type::FLOAT64}, {"float64", Type::FLOAT64}};
  auto it = m.find(s);
  return it == m.end() ? Type::INVALID : it->second;
}
struct Buffer {
  std::vector<uint8_t> bytes;
  uint8_t *get() { return bytes.data(); }
};
struct PlyData { Type t = Type::INVALID; Buffer buffer; size_t count = 0; bool isList = false; };
struct PlyFile {
  struct Property { std::string name; Type t = Type::INVALID; };
  struct Element { std::string name; size_t count = 0; std::vector<Property> props; };
  struct Group { std::string element; size_t first = 0, n = 0; const uint8_t *src = nullptr; std::shared_ptr<PlyData> dst; };
  std::vector<Element> elements;
  std::vector<Group> groups;
  std::vector<std::string> comments;

  Element *find(const std::string &name) {
    for (auto &e : elements)
      if (e.name == name) return &e;
    return nullptr;
  }
  // ---- writing
  void add_properties_to_element(const std::string &element, const std::vector<std::string> &keys, Type t, size_t count, uint8_t *data, Type list_type,
                                 size_t list_count) {
    if (list_type != Type::INVALID || list_count != 0) throw std::invalid_argument("tinyply stand-in: list properties are not supported");
    Element *e = find(element);
    if (!e) { elements.push_back({element, count, {}}); e = &elements.back(); }
    if (e->count != count) throw std::invalid_argument("tinyply stand-in: inconsistent element count for " + element);
    Group g{element, e->props.size(), keys.size(), data, nullptr};
    for (auto &k : keys) e->props.push_back({k, t});
    groups.push_back(g);
  }
  void write(std::ostream &os, bool binary) {
    if (!binary) throw std::invalid_argument("tinyply stand-in: ascii output is not supported");
    os << "ply\nformat binary_little_endian 1.0\n";
    for (auto &c : comments) os << "comment " << c << "\n";
    for (auto &e : elements) {
      os << "element " << e.name << " " << e.count << "\n";
      for (auto &p : e.props) os << "property " << type_name(p.t) << " " << p.name << "\n";
    }
    os << "end_header\n";
    for (auto &e : elements)
      for (size_t r = 0; r < e.count; ++r)
        for (auto &g : groups) {
          if (g.element != e.name) continue;
          const size_t stride = g.n * type_size(e.props[g.first].t);
          os.write(reinterpret_cast<const char *>(g.src + r * stride), (std::streamsize)stride);
        }
  }
  // ---- reading
  bool parse_header(std::istream &is) {
    std::string line;
    if (!std::getline(is, line) || line.substr(0, 3) != "ply") return false;
    while (std::getline(is, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      std::istringstream ls(line);
      std::string tok;
      ls >> tok;
      if (tok == "end_header") return true;
      if (tok == "format") {
        std::string f;
        ls >> f;
        if (f != "binary_little_endian") throw std::invalid_argument("tinyply stand-in: only binary_little_endian files are supported");
      } else if (tok == "comment") {
        std::string rest;
        std::getline(ls, rest);
        comments.push_back(rest.empty() ? rest : rest.substr(1));
      } else if (tok == "element") {
        Element e;
        ls >> e.name >> e.count;
        elements.push_back(e);
      } else if (tok == "property") {
        std::string ty, name;
        ls >> ty;
        if (ty == "list") throw std::invalid_argument("tinyply stand-in: list properties are not supported");
        ls >> name;
        if (elements.empty()) return false;
        elements.back().props.push_back({name, type_from_name(ty)});
      }
    }
    return false;
  }
  std::shared_ptr<PlyData> request_properties_from_element(const std::string &element, const std::vector<std::string> &keys, uint32_t = 0) {
    Element *e = find(element);
    if (!e) throw std::invalid_argument("the element key was not found in the header: " + element);
    size_t first = e->props.size();
    for (size_t i = 0; i < e->props.size(); ++i)
      if (e->props[i].name == keys.at(0)) { first = i; break; }
    for (size_t k = 0; k < keys.size(); ++k)
      if (first + k >= e->props.size() || e->props[first + k].name != keys[k] || e->props[first + k].t != e->props[first].t)
        throw std::invalid_argument("the property key was not found in the header: " + keys[k]);
    auto d = std::make_shared<PlyData>();
    d->t = e->props[first].t;
    d->count = e->count;
    groups.push_back({element, first, keys.size(), nullptr, d});
    return d;
  }
  void read(std::istream &is) {
    for (auto &e : elements) {
      size_t row = 0;
      std::vector<size_t> off(e.props.size());
      for (size_t i = 0; i < e.props.size(); ++i) { off[i] = row; row += type_size(e.props[i].t); }
      std::vector<uint8_t> data(row * e.count);
      is.read(reinterpret_cast<char *>(data.data()), (std::streamsize)data.size());
      if ((size_t)is.gcount() != data.size()) throw std::runtime_error("tinyply stand-in: unexpected end of file");
      for (auto &g : groups) {
        if (g.element != e.name || !g.dst) continue;
        const size_t w = g.n * type_size(e.props[g.first].t);
        g.dst->buffer.bytes.resize(w * e.count);
        for (size_t r = 0; r < e.count; ++r) std::memcpy(g.dst->buffer.bytes.data() + r * w, data.data() + r * row + off[g.first], w);
      }
    }
  }
  std::vector<std::string> &get_comments() { return comments; }
};
}  // namespace tinyply
namespace ply_utils {
inline tinyply::Type torch_type_to_ply_type(c10::ScalarType t) {
  switch (t) {
    case torch::kFloat32: return tinyply::Type::FLOAT32;
    case torch::kFloat64: return tinyply::Type::FLOAT64;
    case torch::kInt32: return tinyply::Type::INT32;
    case torch::kInt16: return tinyply::Type::INT16;
    case torch::kUInt8: return tinyply::Type::UINT8;
    case torch::kInt8: return tinyply::Type::INT8;
    default: return tinyply::Type::INVALID;
  }
}
inline bool export_to_ply(const std::string &, const torch::Tensor &, const torch::Tensor & = torch::Tensor(), const torch::Tensor & = torch::Tensor()) { return true; }
inline bool read_ply_file_to_map_tensor(const std::string &, std::map<std::string, torch::Tensor> &, const torch::Device & = torch::kCPU) { return false; }
inline bool read_ply_file_to_tensor(const std::string &, std::map<std::string, torch::Tensor> &, const torch::Device & = torch::kCPU) { return true; }
}  // namespace ply_utils
