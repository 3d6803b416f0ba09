// densify_stats.hip — a8: NeuralGS::update_state (/root/reference/include/neural_gaussian/neural_gaussian.cpp:626-680)
// in one launch: the per-iteration consumer of the compositing backward's `densify` (or `absgrad`) gradient.
//   grad2d[id] += || (g.x * W/2 * C, g.y * H/2 * C) ||_2        (:660-665)
//   vis[id]     = max(vis[id], visibilities[m])                  (:668-670)
//   count[id]  += 1                                              (:672-673)
//   radii[id]   = max(radii[id], radii_px[m] / max(W, H))        (:675-679, only when the 2-D scale criterion is on)
// The reference runs ~12 eager kernels (clone, 2 strided mul + index_put, norm, 2 index_add, index_select, maximum,
// index_put, ones_like, ...) over the M visible rows; they are launch latency, not work.  Rows of one camera carry unique
// Gaussian ids (packed projection), so C == 1 takes plain read-modify-writes; C > 1 uses atomics (float max through the
// bit pattern of non-negative values).
#include "common.h"

namespace gsdf {

template <bool UNIQUE>
__global__ void __launch_bounds__(256)
    densify_stats_kernel(int64_t M, const float2 *__restrict__ grad, const int64_t *__restrict__ ids,
                         const float *__restrict__ visibilities, const int32_t *__restrict__ radii_px, float sx, float sy,
                         float inv_image_size, float *__restrict__ grad2d, float *__restrict__ count,
                         float *__restrict__ vis, float *__restrict__ radii) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int64_t id = ids[m];
  const float2 g = grad[m];
  const float gx = g.x * sx, gy = g.y * sy;
  const float nrm = sqrtf(gx * gx + gy * gy);
  const float v = visibilities[m];
  if (UNIQUE) {
    grad2d[id] += nrm;
    count[id] += 1.0f;
    vis[id] = fmaxf(vis[id], v);
    if (radii) radii[id] = fmaxf(radii[id], (float)radii_px[m] * inv_image_size);
  } else {
    atomicAdd(grad2d + id, nrm);
    atomicAdd(count + id, 1.0f);
    atomicMax(reinterpret_cast<unsigned *>(vis) + id, __float_as_uint(fmaxf(v, 0.f)));
    if (radii) atomicMax(reinterpret_cast<unsigned *>(radii) + id, __float_as_uint((float)radii_px[m] * inv_image_size));
  }
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_densify_stats(int64_t M, int64_t N, int n_cameras, int width, int height, const float *grad,
                                  const int64_t *gaussian_ids, const float *visibilities, const int32_t *radii_px,
                                  float *grad2d, float *count, float *vis, float *radii, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_densify_stats");
  GSDF_REQUIRE(M >= 0 && N >= 0 && n_cameras >= 1 && width > 0 && height > 0, "densify_stats: bad arguments");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(grad && gaussian_ids && visibilities && grad2d && count && vis, "densify_stats: null buffer");
  GSDF_REQUIRE(!radii || radii_px, "densify_stats: radii state without radii");
  const float sx = (float)width * 0.5f * (float)n_cameras, sy = (float)height * 0.5f * (float)n_cameras;
  const float inv = 1.0f / (float)(width > height ? width : height);
  const unsigned nb = (unsigned)((M + 255) / 256);
  const float2 *g2 = reinterpret_cast<const float2 *>(grad);
  if (n_cameras == 1)
    densify_stats_kernel<true><<<nb, 256, 0, stream>>>(M, g2, gaussian_ids, visibilities, radii_px, sx, sy, inv, grad2d, count, vis, radii);
  else
    densify_stats_kernel<false><<<nb, 256, 0, stream>>>(M, g2, gaussian_ids, visibilities, radii_px, sx, sy, inv, grad2d, count, vis, radii);
  GSDF_CHECK_LAUNCH("densify_stats_kernel");
  return GSDF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row surgery on a FIELD-MAJOR flat buffer [field 0: n x w0 | field 1: n x w1 | ...] (trainer.SplatParams / FusedAdam moments):
// dst rows 0..n_keep-1 = src rows keep[r] (keep == nullptr: identity), field by field, ONE launch for all fields.  This is the
// index_select half of the reference's optimizer surgery at every refinement step (optimizer_utils.cpp:5-165 re-materialises
// every parameter and both Adam moments with index_select / cat, ~40 launches); the appended rows are written by the caller.
namespace gsdf {
static constexpr int ROWS_MAX_FIELDS = 16;
struct RowFields {
  int32_t width[ROWS_MAX_FIELDS], col0[ROWS_MAX_FIELDS + 1];  // col0: first column of the field within the concatenated row
  int n;
};
__global__ void __launch_bounds__(256)
    flat_rows_gather_kernel(RowFields rf, int64_t n_src, int64_t n_dst, int64_t n_keep, const int64_t *__restrict__ keep,
                            const float *__restrict__ src, float *__restrict__ dst) {
  const int wt = rf.col0[rf.n];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_keep * wt) return;
  const int64_t r = i / wt;
  const int c = (int)(i - r * wt);
  int f = 0;
  while (f + 1 < rf.n && c >= rf.col0[f + 1]) ++f;
  const int w = rf.width[f], cc = c - rf.col0[f];
  const int64_t sr = keep ? keep[r] : r;
  dst[(int64_t)rf.col0[f] * n_dst + r * w + cc] = src[(int64_t)rf.col0[f] * n_src + sr * w + cc];
}
}  // namespace gsdf

extern "C" int gsdf_flat_rows_gather(int n_fields, const int32_t *widths_host, int64_t n_src, int64_t n_dst, int64_t n_keep,
                                     const int64_t *keep_idx, const float *src, float *dst, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_flat_rows_gather");
  GSDF_REQUIRE(n_fields >= 1 && n_fields <= ROWS_MAX_FIELDS && widths_host, "flat_rows_gather: 1..%d fields", ROWS_MAX_FIELDS);
  GSDF_REQUIRE(n_src >= 0 && n_dst >= n_keep && n_keep >= 0 && (keep_idx || n_keep <= n_src), "flat_rows_gather: bad row counts");
  RowFields rf;
  rf.n = n_fields;
  int c = 0;
  for (int f = 0; f < n_fields; ++f) {
    GSDF_REQUIRE(widths_host[f] >= 0, "flat_rows_gather: negative field width");
    rf.width[f] = widths_host[f];
    rf.col0[f] = c;
    c += widths_host[f];
  }
  for (int f = n_fields; f <= ROWS_MAX_FIELDS; ++f) rf.col0[f] = c;
  for (int f = n_fields; f < ROWS_MAX_FIELDS; ++f) rf.width[f] = 0;
  const int64_t n = n_keep * c;
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(src && dst, "flat_rows_gather: null buffer");
  flat_rows_gather_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(rf, n_src, n_dst, n_keep, keep_idx, src, dst);
  GSDF_CHECK_LAUNCH("flat_rows_gather_kernel");
  return GSDF_OK;
}
