// error.hip — thread-local error string + ABI version of libgsdf_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace gsdf {
static std::mutex g_xcd_mutex;
static std::unordered_map<void *, int> g_stream_xcds;
int xcd_count(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_xcd_mutex);
  const auto it = g_stream_xcds.find((void *)stream);
  return it == g_stream_xcds.end() ? 8 : it->second;
}

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gsdf

extern "C" const char *gsdf_last_error(void) { return gsdf::g_err; }
extern "C" int gsdf_abi_version(void) { return 2; }

extern "C" int gsdf_stream_set_xcds(gsdf_stream_t stream, int n_xcds) {
  GSDF_REQUIRE(n_xcds >= 0 && n_xcds <= 8, "stream_set_xcds: n_xcds %d outside [0,8]", n_xcds);
  std::lock_guard<std::mutex> lock(gsdf::g_xcd_mutex);
  if (n_xcds == 0) gsdf::g_stream_xcds.erase((void *)stream);
  else gsdf::g_stream_xcds[(void *)stream] = n_xcds;
  return GSDF_OK;
}
