// Minimal stand-in for nlohmann::json, used ONLY when the real single header is not on the include path
// (the reference gets it from tiny-cuda-nn/dependencies, CMakeLists.txt:93-96).  Supports exactly what the
// reference's call sites and tcnn_binding.h need: brace-initialised objects {{"key", value}, ...} and
// value(key, default) / contains(key).
#pragma once
#include <initializer_list>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

namespace nlohmann {
class json {
 public:
  json() = default;
  json(const char *s) : kind_(kStr), str_(s) {}
  json(const std::string &s) : kind_(kStr), str_(s) {}
  json(bool b) : kind_(kNum), num_(b ? 1.0 : 0.0) {}
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  json(T v) : kind_(kNum), num_((double)v) {}
  json(std::initializer_list<json> init) {
    // {{"k", v}, {"k2", v2}} : every element is a 2-element list whose first entry is a string
    bool object = true;
    for (const json &e : init) object = object && e.kind_ == kList && e.list_.size() == 2 && e.list_.begin()->kind_ == kStr;
    if (object && init.size() > 0) {
      kind_ = kObj;
      for (const json &e : init) obj_[e.list_.begin()->str_] = *(e.list_.begin() + 1);
    } else {
      kind_ = kList;
      list_ = std::vector<json>(init);
    }
  }
  bool contains(const std::string &k) const { return kind_ == kObj && obj_.count(k); }
  template <typename T>
  T value(const std::string &k, const T &def) const {
    auto it = obj_.find(k);
    if (kind_ != kObj || it == obj_.end()) return def;
    return it->second.template get<T>();
  }
  std::string value(const std::string &k, const char *def) const { return value<std::string>(k, std::string(def)); }
  template <typename T>
  T get() const {
    if constexpr (std::is_same<T, std::string>::value) return str_;
    else return (T)num_;
  }

 private:
  enum Kind { kNull, kNum, kStr, kObj, kList };
  Kind kind_ = kNull;
  double num_ = 0;
  std::string str_;
  std::map<std::string, json> obj_;
  std::vector<json> list_;
};
}  // namespace nlohmann
