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
