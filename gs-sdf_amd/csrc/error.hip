// error.hip — thread-local error string + ABI version of libgsdf_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace gsdf {
static std::mutex g_xcd_mutex;
static std::unordered_map<void *, int> g_stream_xcds;
int xcd_count(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_xcd_mutex);
  const auto it = g_stream_xcds.find((void *)stream);
  return it == g_stream_xcds.end() ? 8 : it->second;
}

// ---- per-entry-point timing ------------------------------------------------------------------------------------------------
static std::atomic<int> g_timing_on{0};
static std::mutex g_timing_mutex;
static std::vector<std::string> g_timing_only;
struct TimedCall { const char *name; hipEvent_t a, b; };
static std::vector<TimedCall> g_timed;
bool timing_wants(const char *name) {
  if (!g_timing_on.load(std::memory_order_relaxed)) return false;
  std::lock_guard<std::mutex> lock(g_timing_mutex);
  if (g_timing_only.empty()) return true;
  for (const auto &n : g_timing_only) if (n == name) return true;
  return false;
}
void timing_push(const char *name, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::mutex> lock(g_timing_mutex);
  g_timed.push_back({name, a, b});
}

// process-wide, not per thread: a training step's backward kernels are launched by libtorch's autograd thread, not by the thread that asked
static std::atomic<int> g_deterministic{0};
bool deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gsdf

extern "C" const char *gsdf_last_error(void) { return gsdf::g_err; }
extern "C" int gsdf_abi_version(void) { return GSDF_ABI_VERSION; }

extern "C" int gsdf_host_words_alloc(int n_words, int64_t **host_view, int64_t **device_view) {
  GSDF_REQUIRE(n_words > 0 && host_view && device_view, "host_words_alloc: bad arguments");
  void *h = nullptr, *d = nullptr;
  // coherent (fine-grained) pinned memory: device stores are not held back in the L2 until the end of the kernel
  // portable: the words are mapped on every device of the process (a multi-GPU process may launch on a device other than the one that was
  // current at allocation time)
  if (hipHostMalloc(&h, (size_t)n_words * sizeof(int64_t), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    GSDF_HIP(hipHostMalloc(&h, (size_t)n_words * sizeof(int64_t), hipHostMallocMapped | hipHostMallocPortable), "host_words_alloc");
  }
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(h);
    gsdf::set_error("host_words_alloc: no device view of the pinned words");
    return GSDF_ERR_LAUNCH;
  }
  memset(h, 0, (size_t)n_words * sizeof(int64_t));
  *host_view = (int64_t *)h;
  *device_view = (int64_t *)d;
  return GSDF_OK;
}
extern "C" int gsdf_host_words_free(int64_t *host_view) {
  if (host_view != nullptr) GSDF_HIP(hipHostFree(host_view), "host_words_free");
  return GSDF_OK;
}

extern "C" int gsdf_deterministic(int on) {
  const int before = gsdf::g_deterministic.load();
  if (on >= 0) gsdf::g_deterministic.store(on != 0);
  return before;
}

extern "C" int gsdf_stream_set_xcds(gsdf_stream_t stream, int n_xcds) {
  GSDF_REQUIRE(n_xcds >= 0 && n_xcds <= 8, "stream_set_xcds: n_xcds %d outside [0,8]", n_xcds);
  std::lock_guard<std::mutex> lock(gsdf::g_xcd_mutex);
  if (n_xcds == 0) gsdf::g_stream_xcds.erase((void *)stream);
  else gsdf::g_stream_xcds[(void *)stream] = n_xcds;
  return GSDF_OK;
}

extern "C" int gsdf_timing_begin(const char *only_csv) {
  std::lock_guard<std::mutex> lock(gsdf::g_timing_mutex);
  for (auto &c : gsdf::g_timed) { (void)hipEventDestroy(c.a); (void)hipEventDestroy(c.b); }
  gsdf::g_timed.clear();
  gsdf::g_timing_only.clear();
  if (only_csv != nullptr) {
    std::string cur;
    for (const char *p = only_csv;; ++p) {
      if (*p == ',' || *p == 0) { if (!cur.empty()) gsdf::g_timing_only.push_back(cur); cur.clear(); if (*p == 0) break; }
      else if (*p != ' ') cur.push_back(*p);
    }
  }
  gsdf::g_timing_on.store(1);
  return GSDF_OK;
}

extern "C" size_t gsdf_timing_end(char *buf, size_t cap) {
  gsdf::g_timing_on.store(0);
  std::vector<gsdf::TimedCall> calls;
  {
    std::lock_guard<std::mutex> lock(gsdf::g_timing_mutex);
    calls.swap(gsdf::g_timed);
  }
  struct Agg { long n = 0; double sum = 0, mn = 1e30, mx = 0; std::vector<float> all; };
  std::map<std::string, Agg> agg;
  for (auto &c : calls) {
    float ms = 0.f;
    if (hipEventSynchronize(c.b) == hipSuccess && hipEventElapsedTime(&ms, c.a, c.b) == hipSuccess) {
      Agg &g = agg[c.name];
      g.n++; g.sum += ms; g.mn = ms < g.mn ? ms : g.mn; g.mx = ms > g.mx ? ms : g.mx; g.all.push_back(ms);
    }
    (void)hipEventDestroy(c.a); (void)hipEventDestroy(c.b);
  }
  std::string out;
  char line[256];
  for (auto &kv : agg) {
    std::vector<float> &v = kv.second.all;
    std::sort(v.begin(), v.end());
    const double med = v.empty() ? 0.0 : (v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]));
    snprintf(line, sizeof(line), "%s %ld %.6f %.6f %.6f %.6f\n", kv.first.c_str(), kv.second.n, kv.second.sum, kv.second.mn, kv.second.mx, med);
    out += line;
  }
  if (buf != nullptr && cap > 0) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

// one line per timed call in the order of the calls, "name begin_ms end_ms" relative to the first call's begin event (events of different streams
// compare on one device): a device timeline of the entry points without a profiler attached (bench.py --step-trace).  Stops the timing like _end.
extern "C" size_t gsdf_timing_trace(char *buf, size_t cap) {
  gsdf::g_timing_on.store(0);
  std::vector<gsdf::TimedCall> calls;
  {
    std::lock_guard<std::mutex> lock(gsdf::g_timing_mutex);
    calls.swap(gsdf::g_timed);
  }
  std::string out;
  char line[256];
  for (auto &c : calls) (void)hipEventSynchronize(c.b);
  for (auto &c : calls) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, calls[0].a, c.a) == hipSuccess && hipEventElapsedTime(&b, calls[0].a, c.b) == hipSuccess) {
      snprintf(line, sizeof(line), "%s %.6f %.6f\n", c.name, a, b);
      out += line;
    }
  }
  for (auto &c : calls) { (void)hipEventDestroy(c.a); (void)hipEventDestroy(c.b); }
  if (buf != nullptr && cap > 0) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}
