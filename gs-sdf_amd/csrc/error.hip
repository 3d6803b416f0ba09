// error.hip — thread-local error string + ABI version of libgsdf_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace gsdf {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gsdf

extern "C" const char *gsdf_last_error(void) { return gsdf::g_err; }
extern "C" int gsdf_abi_version(void) { return 1; }
