// adam.hip — fused multi-segment Adam step over a flat parameter buffer (SURVEY 8f "next #1": the metric is
// TRAIN iters/s and the reference runs an unfused libtorch Adam over 14*N + 15.3 M parameters every iteration,
// /root/reference/include/neural_mapping/neural_mapping.cpp:466-469, groups at neural_gaussian.cpp:434-453).
// One launch per flat buffer; per-element traffic 16 B read + 12 B written (p, g, m, v) -> HBM-bound streaming
// kernel, float4 accesses.  Semantics = torch.optim.Adam (no amsgrad, no weight decay): bias-corrected step.
#include "common.h"

namespace gsdf {

static constexpr int ADAM_MAX_SEG = 16;
struct AdamSegs {
  int64_t begin[ADAM_MAX_SEG + 1];  // element offsets (multiples of 4 except the last end)
  float lr[ADAM_MAX_SEG];
  int n;
};

template <bool ZERO_GRAD>
__global__ void __launch_bounds__(256)
    adam_kernel(int64_t n_vec4, int64_t n, AdamSegs segs, float *__restrict__ p, float *__restrict__ g,
                float *__restrict__ m, float *__restrict__ v, float beta1, float beta2, float eps, float inv_bc1,
                float inv_sqrt_bc2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec4; i += (int64_t)gridDim.x * 256) {
    const int64_t e0 = i * 4;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = e0 + 4 <= n;
    if (full) {
      const float4 a = *reinterpret_cast<const float4 *>(p + e0), b = *reinterpret_cast<const float4 *>(g + e0);
      const float4 c = *reinterpret_cast<const float4 *>(m + e0), d = *reinterpret_cast<const float4 *>(v + e0);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = e0 + k < n;
        pv[k] = ok ? p[e0 + k] : 0.f; gv[k] = ok ? g[e0 + k] : 0.f; mv[k] = ok ? m[e0 + k] : 0.f; vv[k] = ok ? v[e0 + k] : 0.f;
      }
    }
    int s = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MAX_SEG; ++k) s += (k < segs.n && e0 >= segs.begin[k]) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // a vector of 4 may straddle segment boundaries that are not 4-aligned, several of them when segments are shorter
      // than 4 elements or empty (e.g. features_rest at sh_degree 0): the element's segment is the LAST one that begins
      // at or before it
      int sk = s;
      while (sk + 1 < segs.n && e0 + k >= segs.begin[sk + 1]) ++sk;
      const float lr = segs.lr[sk];
      mv[k] = beta1 * mv[k] + (1.f - beta1) * gv[k];
      vv[k] = beta2 * vv[k] + (1.f - beta2) * gv[k] * gv[k];
      const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
      pv[k] -= (lr * inv_bc1) * (mv[k] / denom);
    }
    if (full) {
      *reinterpret_cast<float4 *>(p + e0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4 *>(m + e0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4 *>(v + e0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (ZERO_GRAD) *reinterpret_cast<float4 *>(g + e0) = make_float4(0.f, 0.f, 0.f, 0.f);   // the step's zero_grad, without a pass of its own
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (e0 + k < n) { p[e0 + k] = pv[k]; m[e0 + k] = mv[k]; v[e0 + k] = vv[k]; if (ZERO_GRAD) g[e0 + k] = 0.f; }
    }
  }
}

}  // namespace gsdf

using namespace gsdf;

static int adam_step_impl(int64_t n, int n_segments, const int64_t *seg_begin_host, const float *seg_lr_host, float *params, float *grads,
                          float *exp_avg, float *exp_avg_sq, float beta1, float beta2, float eps, int64_t step, bool zero_grad, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_adam_step");
  GSDF_REQUIRE(n >= 0 && n_segments >= 1 && n_segments <= ADAM_MAX_SEG, "adam_step: %d segments not in [1,%d]", n_segments,
               ADAM_MAX_SEG);
  GSDF_REQUIRE(step >= 1, "adam_step: step counts from 1");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(params && grads && exp_avg && exp_avg_sq && seg_begin_host && seg_lr_host, "adam_step: null buffer");
  AdamSegs segs;
  segs.n = n_segments;
  for (int k = 0; k < n_segments; ++k) {
    segs.begin[k] = seg_begin_host[k];
    segs.lr[k] = seg_lr_host[k];
    GSDF_REQUIRE(k == 0 ? seg_begin_host[0] == 0 : seg_begin_host[k] >= seg_begin_host[k - 1], "adam_step: segments must be sorted from 0");
  }
  segs.begin[n_segments] = n;
  for (int k = n_segments; k < ADAM_MAX_SEG; ++k) { segs.begin[k + 1] = n; segs.lr[k] = 0.f; }
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t n4 = (n + 3) / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (zero_grad)
    adam_kernel<true><<<(unsigned)blocks, 256, 0, stream>>>(n4, n, segs, params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, (float)(1.0 / bc1),
                                                            (float)(1.0 / sqrt(bc2)));
  else
    adam_kernel<false><<<(unsigned)blocks, 256, 0, stream>>>(n4, n, segs, params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, (float)(1.0 / bc1),
                                                             (float)(1.0 / sqrt(bc2)));
  GSDF_CHECK_LAUNCH("adam_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_adam_step(int64_t n, int n_segments, const int64_t *seg_begin_host, const float *seg_lr_host, float *params,
                              const float *grads, float *exp_avg, float *exp_avg_sq, float beta1, float beta2, float eps, int64_t step,
                              gsdf_stream_t stream) {
  return adam_step_impl(n, n_segments, seg_begin_host, seg_lr_host, params, const_cast<float *>(grads), exp_avg, exp_avg_sq, beta1, beta2, eps, step, false,
                        stream);
}

extern "C" int gsdf_adam_step_zero_grad(int64_t n, int n_segments, const int64_t *seg_begin_host, const float *seg_lr_host, float *params, float *grads,
                                        float *exp_avg, float *exp_avg_sq, float beta1, float beta2, float eps, int64_t step, gsdf_stream_t stream) {
  return adam_step_impl(n, n_segments, seg_begin_host, seg_lr_host, params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, step, true, stream);
}
