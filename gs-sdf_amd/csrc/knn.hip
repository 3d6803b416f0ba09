// knn.hip — K1: distCUDA2(points) = mean of the squared distances to the 3 nearest neighbours (self excluded).
// Replaces the reference's (absent) simple-knn submodule, call site
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:314 (initial splat scale; init-time only).
// Exact, like simple-knn (Morton sort + box search there); here: uniform grid sized to ~2 points per cell,
// points sorted by cell id (two stable 11-bit passes of the hand-written radix sort of the tile binning, radix.hip: cell ids
// are < 2^22), then one lane per point walks Chebyshev shells of cells until the 3rd-best distance is closer than the
// unexplored region.
#include <cstring>

#include "radix.h"

namespace gsdf {

struct KnnGrid {
  float lo[3], inv_cell[3], cell[3];
  int res[3];
};

__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void knn_bbox_init_kernel(int *bb) {
  if (threadIdx.x < 3) { bb[threadIdx.x] = 0x7FFFFFFF; bb[3 + threadIdx.x] = (int)0x80000000; }
}

__global__ void __launch_bounds__(256) knn_bbox_kernel(int64_t N, const float *__restrict__ pts, int *bb) {
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float v = pts[3 * i + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int s = 32; s >= 1; s >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], s, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], s, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(bb + d, f2ord(lo[d])); atomicMax(bb + 3 + d, f2ord(hi[d])); }
  }
}

__global__ void knn_grid_kernel(int64_t N, int64_t max_cells, const int *bb, KnnGrid *g) {
  // cubic cells with ~2 points each, at most max_cells cells, at least 1 per axis.  Degenerate clouds (exactly planar
  // or collinear: one or two extents are 0) are sized from the NON-degenerate extents only — a cell size derived from a
  // clamped 1e-12 extent would ask for ~1e12 cells along the other axes and overflow the cell tables.
  float ext[3], lo[3];
  float emax = 0.f;
  for (int d = 0; d < 3; ++d) { lo[d] = ord2f(bb[d]); ext[d] = fmaxf(ord2f(bb[3 + d]) - lo[d], 0.f); emax = fmaxf(emax, ext[d]); }
  const float tiny = fmaxf(emax * 1e-6f, 1e-30f);
  const float target = fminf((float)max_cells, fmaxf((float)N * 0.5f, 1.f));
  float vol = 1.f;
  int dims = 0;
  for (int d = 0; d < 3; ++d)
    if (ext[d] > tiny) { vol *= ext[d]; ++dims; }
  float cell = dims == 0 ? 1.f : powf(vol / target, 1.f / (float)dims);
  cell = fmaxf(cell, emax / 2048.f);                 // never more than 2048 cells along an axis (int overflow guard)
  for (int it = 0; it < 64; ++it) {                  // grow until the grid fits: x1.26 per round = x2 cells per 3 rounds
    double cells = 1.0;
    for (int d = 0; d < 3; ++d) cells *= (double)((int)(ext[d] / cell) + 1);
    if (cells <= (double)max_cells) break;
    cell *= 1.26f;
  }
  double cells = 1.0;
  for (int d = 0; d < 3; ++d) {
    g->lo[d] = lo[d];
    g->res[d] = (int)(ext[d] / cell) + 1;
    cells *= (double)g->res[d];
    g->cell[d] = cell;
    g->inv_cell[d] = 1.f / cell;
  }
  if (cells > (double)max_cells) {                   // cannot happen after 64 rounds unless max_cells < 1: one cell
    for (int d = 0; d < 3; ++d) { g->res[d] = 1; g->cell[d] = fmaxf(emax, 1.f); g->inv_cell[d] = 1.f / g->cell[d]; }
  }
}

__device__ __forceinline__ void cell_of(const KnnGrid &g, float x, float y, float z, int c[3]) {
  c[0] = min(max((int)((x - g.lo[0]) * g.inv_cell[0]), 0), g.res[0] - 1);
  c[1] = min(max((int)((y - g.lo[1]) * g.inv_cell[1]), 0), g.res[1] - 1);
  c[2] = min(max((int)((z - g.lo[2]) * g.inv_cell[2]), 0), g.res[2] - 1);
}

__global__ void __launch_bounds__(256) knn_cellid_kernel(int64_t N, const float *__restrict__ pts, const KnnGrid *gp,
                                                         uint32_t *__restrict__ keys, int32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const KnnGrid g = *gp;
  int c[3];
  cell_of(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], c);
  keys[i] = (uint32_t)((c[2] * g.res[1] + c[1]) * g.res[0] + c[0]);
  vals[i] = (int32_t)i;
}

__global__ void __launch_bounds__(256)
    knn_gather_kernel(int64_t N, const float *__restrict__ pts, const uint32_t *__restrict__ keys,
                      const int32_t *__restrict__ vals, float4 *__restrict__ sorted, int32_t *__restrict__ cstart,
                      int32_t *__restrict__ cend) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int32_t src = vals[i];
  sorted[i] = make_float4(pts[3 * (int64_t)src], pts[3 * (int64_t)src + 1], pts[3 * (int64_t)src + 2], __int_as_float(src));
  const uint32_t k = keys[i];
  if (i == 0 || keys[i - 1] != k) cstart[k] = (int32_t)i;
  if (i == N - 1 || keys[i + 1] != k) cend[k] = (int32_t)i + 1;
}

__global__ void __launch_bounds__(256)
    knn_search_kernel(int64_t N, const float4 *__restrict__ sorted, const int32_t *__restrict__ cstart,
                      const int32_t *__restrict__ cend, const KnnGrid *gp, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const KnnGrid g = *gp;
  const float4 p = sorted[i];
  int c[3];
  cell_of(g, p.x, p.y, p.z, c);
  float b0 = 3.4e38f, b1 = 3.4e38f, b2 = 3.4e38f;
  const int rmax = max(g.res[0], max(g.res[1], g.res[2]));
  for (int r = 0; r <= rmax; ++r) {
    const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.res[2] - 1);
    const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.res[1] - 1);
    const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.res[0] - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const bool shell_zy = (abs(z - c[2]) == r) || (abs(y - c[1]) == r);
        for (int x = x0; x <= x1; ++x) {
          if (!shell_zy && abs(x - c[0]) != r) { x = max(x, c[0] + r - 1); continue; }  // interior: jump to the far face
          const int64_t cid = ((int64_t)z * g.res[1] + y) * g.res[0] + x;
          const int32_t s = cstart[cid], e = cend[cid];
          for (int32_t j = s; j < e; ++j) {
            if (j == i) continue;
            const float4 q = sorted[j];
            const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < b2) {
              if (d < b0) { b2 = b1; b1 = b0; b0 = d; }
              else if (d < b1) { b2 = b1; b1 = d; }
              else b2 = d;
            }
          }
        }
      }
    // everything unexplored lies outside the cube of radius r around the cell
    float dmin = 3.4e38f;
    bool covers = true;
    if (c[0] - r > 0) { dmin = fminf(dmin, p.x - (g.lo[0] + (c[0] - r) * g.cell[0])); covers = false; }
    if (c[0] + r < g.res[0] - 1) { dmin = fminf(dmin, (g.lo[0] + (c[0] + r + 1) * g.cell[0]) - p.x); covers = false; }
    if (c[1] - r > 0) { dmin = fminf(dmin, p.y - (g.lo[1] + (c[1] - r) * g.cell[1])); covers = false; }
    if (c[1] + r < g.res[1] - 1) { dmin = fminf(dmin, (g.lo[1] + (c[1] + r + 1) * g.cell[1]) - p.y); covers = false; }
    if (c[2] - r > 0) { dmin = fminf(dmin, p.z - (g.lo[2] + (c[2] - r) * g.cell[2])); covers = false; }
    if (c[2] + r < g.res[2] - 1) { dmin = fminf(dmin, (g.lo[2] + (c[2] + r + 1) * g.cell[2]) - p.z); covers = false; }
    if (covers) break;
    dmin = fmaxf(dmin, 0.f) * 0.9999f;  // fp32 slack on the face positions
    if (b2 <= dmin * dmin) break;
  }
  float sum = 0.f;
  if (b0 < 3.0e38f) sum += b0;
  if (b1 < 3.0e38f) sum += b1;
  if (b2 < 3.0e38f) sum += b2;
  out[__float_as_int(p.w)] = sum * (1.0f / 3.0f);
}

static int64_t knn_max_cells(int64_t N) {
  int64_t c = N < 4096 ? 4096 : N;
  return c > (1 << 22) ? (1 << 22) : c;
}
static size_t knn_sort_temp(int64_t N) { return radix_ws_bytes(N); }

}  // namespace gsdf

using namespace gsdf;

extern "C" size_t gsdf_knn_ws_bytes(int64_t N) {
  if (N <= 0) return 256;
  const int64_t mc = knn_max_cells(N);
  return 256 + 256 + 4 * align_up((size_t)N * 4, 256) + align_up((size_t)N * 16, 256) + 2 * align_up((size_t)mc * 4, 256) +
         align_up(knn_sort_temp(N), 256);
}

extern "C" int gsdf_knn_mean_dist2(int64_t N, const float *points, float *out, void *ws, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_knn_mean_dist2");
  if (N <= 0) return GSDF_OK;
  GSDF_REQUIRE(points && out && ws, "knn_mean_dist2: null buffer");
  GSDF_REQUIRE(N < (1LL << 31), "knn_mean_dist2: too many points");
  const int64_t mc = knn_max_cells(N);
  char *p = (char *)ws;
  int *bb = (int *)p; p += 256;
  KnnGrid *grid = (KnnGrid *)p; p += 256;
  uint32_t *keys = (uint32_t *)p; p += align_up((size_t)N * 4, 256);
  uint32_t *keys2 = (uint32_t *)p; p += align_up((size_t)N * 4, 256);
  int32_t *vals = (int32_t *)p; p += align_up((size_t)N * 4, 256);
  int32_t *vals2 = (int32_t *)p; p += align_up((size_t)N * 4, 256);
  float4 *sorted = (float4 *)p; p += align_up((size_t)N * 16, 256);
  int32_t *cstart = (int32_t *)p; p += align_up((size_t)mc * 4, 256);
  int32_t *cend = (int32_t *)p; p += align_up((size_t)mc * 4, 256);
  void *temp = p;
  size_t temp_bytes = knn_sort_temp(N);
  const unsigned nb = (unsigned)((N + 255) / 256);
  knn_bbox_init_kernel<<<1, 64, 0, stream>>>(bb);
  knn_bbox_kernel<<<nb < 1024 ? nb : 1024, 256, 0, stream>>>(N, points, bb);
  knn_grid_kernel<<<1, 1, 0, stream>>>(N, mc, bb, grid);
  knn_cellid_kernel<<<nb, 256, 0, stream>>>(N, points, grid, keys, vals);
  GSDF_CHECK_LAUNCH("knn prep kernels");
  // cell ids < max_cells <= 2^22: two stable passes, keys -> keys2 -> keys (values: row numbers from the first pass's iota hook)
  (void)temp_bytes;
  RadixHooks h0{true, nullptr, nullptr, nullptr, nullptr, 1, 0};
  int rc = radix_pass(N, 0, 11, keys, nullptr, keys2, (uint32_t *)vals2, (uint32_t *)temp, &h0, stream);
  if (!rc) rc = radix_pass(N, 11, 11, keys2, (const uint32_t *)vals2, keys, (uint32_t *)vals, (uint32_t *)temp, nullptr, stream);
  if (rc) return rc;
  GSDF_HIP(hipMemsetAsync(cstart, 0, (size_t)mc * 4, stream), "knn memset");
  GSDF_HIP(hipMemsetAsync(cend, 0, (size_t)mc * 4, stream), "knn memset");
  knn_gather_kernel<<<nb, 256, 0, stream>>>(N, points, keys, vals, sorted, cstart, cend);
  knn_search_kernel<<<nb, 256, 0, stream>>>(N, sorted, cstart, cend, grid, out);
  GSDF_CHECK_LAUNCH("knn_search_kernel");
  return GSDF_OK;
}
