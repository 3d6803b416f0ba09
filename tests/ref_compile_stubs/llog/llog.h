// stand-in for the reference's (un-vendored) llog submodule: the calls the hot-path host code makes
// (neural_gaussian.cpp, local_map.cpp: CreateTimer/tic/toc_sum, RecordValue).  Syntax-only test infrastructure.
#pragma once
#include <memory>
#include <string>
namespace llog {
struct Timer { void tic() {} void toc() {} void toc_sum() {} void toc_avg() {} };
inline std::shared_ptr<Timer> CreateTimer(const std::string &) { return std::make_shared<Timer>(); }
template <class T> inline void RecordValue(const std::string &, T, bool = false) {}
inline void Reset() {}
inline std::string FlashValue(const std::string &, int = 3) { return ""; }
inline void PrintLog() {}
inline void InitValueFile(const std::string &) {}
inline void SaveLog(const std::string &) {}
}  // namespace llog
