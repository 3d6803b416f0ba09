// scan.h — device-wide inclusive scan int32 -> int64 (three launches, no host sync).
#pragma once
#include "common.h"

namespace gsdf {
// workspace bytes for scanning n elements
size_t scan_ws_bytes(int64_t n);
// out[i] = sum_{j<=i} in[j]; *total = out[n-1] (0 when n == 0).  `total` is a device pointer.
int scan_inclusive_i32_i64(const int32_t *in, int64_t *out, int64_t n, void *ws, int64_t *total,
                           hipStream_t stream);
}  // namespace gsdf
