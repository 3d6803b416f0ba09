// scan.hip — reduce / downsweep (two launches: a workgroup sums the block totals before it itself).  2048 elements per 256-thread workgroup
// (8 per lane, int4 x2 loads), wave64 prefix by DPP-free shuffles over int64 (cold path: the scans
// here touch <= a few MB, they are launch-latency bound, not bandwidth bound).
#include "scan.h"

namespace gsdf {

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 8;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// block-wide inclusive scan of one int64 per thread (256 threads = 4 waves)
__device__ __forceinline__ int64_t block_incl_scan_i64(int64_t v, int64_t *lds4, int64_t *block_total) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int64_t s = wave_incl_scan_i64(v);
  if (lane == 63) lds4[wave] = s;
  __syncthreads();
  int64_t off = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < wave) off += lds4[w];
  if (block_total) *block_total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return s + off;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const int32_t *__restrict__ in, int64_t n,
                                                                   int64_t *__restrict__ partials) {
  __shared__ int64_t lds4[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) s += in[base + k];
  int64_t tot;
  block_incl_scan_i64(s, lds4, &tot);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// sum of v over the workgroup (every thread gets it)
__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t *lds4) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  const int64_t s = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return s;
}

// Second (and last) launch: a workgroup forms its own exclusive prefix from the block sums before it (a few thousand words at most:
// cheaper than a third launch for the scan of the partials); workgroup 0 sums all of them first and publishes the grand total — the
// host may be polling that word (gsdf_host_words_alloc).
__global__ void __launch_bounds__(SCAN_THREADS) scan_downsweep_kernel(const int32_t *__restrict__ in, int64_t n,
                                                                      const int64_t *__restrict__ partials, int64_t nb,
                                                                      int64_t *__restrict__ out, int64_t *__restrict__ total) {
  __shared__ int64_t lds4[4];
  if (blockIdx.x == 0) {
    int64_t t = 0;
    for (int64_t i = threadIdx.x; i < nb; i += SCAN_THREADS) t += partials[i];
    t = block_sum_i64(t, lds4);
    if (threadIdx.x == 0) store_host_visible(total, t);
  }
  int64_t before = 0;
  for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x; i += SCAN_THREADS) before += partials[i];
  before = block_sum_i64(before, lds4);
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t x[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    x[k] = base + k < n ? in[base + k] : 0;
    s += x[k];
  }
  const int64_t incl = block_incl_scan_i64(s, lds4, nullptr);
  int64_t run = before + incl - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    run += x[k];
    if (base + k < n) out[base + k] = run;
  }
}

__global__ void scan_zero_total_kernel(int64_t *total) { store_host_visible(total, 0); }   // the host may be polling the word

size_t scan_ws_bytes(int64_t n) {
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  return align_up((size_t)(nb > 0 ? nb : 1) * sizeof(int64_t), 256);
}

int scan_inclusive_i32_i64(const int32_t *in, int64_t *out, int64_t n, void *ws, int64_t *total,
                           hipStream_t stream) {
  if (n <= 0) {
    scan_zero_total_kernel<<<1, 1, 0, stream>>>(total);
    GSDF_CHECK_LAUNCH("scan_zero_total");
    return GSDF_OK;
  }
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  int64_t *partials = (int64_t *)ws;
  scan_reduce_kernel<<<(unsigned)nb, SCAN_THREADS, 0, stream>>>(in, n, partials);
  GSDF_CHECK_LAUNCH("scan_reduce");
  scan_downsweep_kernel<<<(unsigned)nb, SCAN_THREADS, 0, stream>>>(in, n, partials, nb, out, total);
  GSDF_CHECK_LAUNCH("scan_downsweep");
  return GSDF_OK;
}

}  // namespace gsdf
