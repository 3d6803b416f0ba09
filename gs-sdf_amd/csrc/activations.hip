// activations.hip — a2: the splat parameter activations of NeuralGS::generate_gaussian / get_xyz / get_scale / get_opacity
// (/root/reference/include/neural_gaussian/neural_gaussian.cpp:463-492): xyz = anchors + offsets, scales = exp(scaling),
// opacity = sigmoid(opacity) — one pass forward instead of three libtorch kernels, and one pass backward that ACCUMULATES
// straight into the (flat) parameter-gradient buffers instead of three backward kernels + three autograd adds.
#include "common.h"

namespace gsdf {

__global__ void __launch_bounds__(256)
    splat_act_fwd_kernel(int64_t n, const float *__restrict__ anchors, const float *__restrict__ offsets,
                         const float *__restrict__ log_scales, const float *__restrict__ logit_opac,
                         float *__restrict__ xyz, float *__restrict__ scales, float *__restrict__ opac) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    xyz[3 * i + k] = anchors[3 * i + k] + offsets[3 * i + k];
    scales[3 * i + k] = expf(log_scales[3 * i + k]);
  }
  opac[i] = 1.0f / (1.0f + expf(-logit_opac[i]));
}

__global__ void __launch_bounds__(256)
    splat_act_bwd_kernel(int64_t n, const float *__restrict__ scales, const float *__restrict__ opac,
                         const float *__restrict__ v_xyz, const float *__restrict__ v_scales,
                         const float *__restrict__ v_opac, float *__restrict__ g_offsets,
                         float *__restrict__ g_log_scales, float *__restrict__ g_logit_opac) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (v_xyz != nullptr) g_offsets[3 * i + k] += v_xyz[3 * i + k];
    if (v_scales != nullptr) g_log_scales[3 * i + k] += v_scales[3 * i + k] * scales[3 * i + k];
  }
  if (v_opac != nullptr) {
    const float o = opac[i];
    g_logit_opac[i] += v_opac[i] * (o * (1.0f - o));
  }
}

static __device__ DetScalarSlot g_det_isotropic;   // deterministic mode: the ordered finish of the loss value (both kernels: never at once)

// ---- isotropic regulariser of the visible splats (neural_mapping.cpp:268-276): scale = get_scale()[gaussian_ids][:, 0:2];
// loss = (scale - scale.mean(-1, keepdim)).abs().mean() = sum_m |s_u - s_v| / (2 M).  One launch each way instead of ~12.
__global__ void __launch_bounds__(256)
    isotropic_fwd_kernel(int64_t M, const float *__restrict__ scales, const int64_t *__restrict__ ids, float inv_2m,
                         float *__restrict__ loss, bool det) {
  __shared__ float s_part[4];
  float c = 0.f;
  // capped grid + grid-stride loop: atomics on ONE address serialise (~88 per microsecond)
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const int64_t g = ids[m];
    c += fabsf(scales[3 * g] - scales[3 * g + 1]) * inv_2m;
  }
  const float ws = wave_sum_to_lane63(c);
  if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = ws;
  __syncthreads();
  finish_scalars((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]), 0.f, loss, nullptr, det ? &g_det_isotropic : nullptr);
}

__global__ void __launch_bounds__(256)
    isotropic_bwd_kernel(int64_t M, const float *__restrict__ scales, const int64_t *__restrict__ ids, float inv_2m,
                         const float *__restrict__ v_loss, float *__restrict__ v_scales) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int64_t g = ids[m];
  const float d = scales[3 * g] - scales[3 * g + 1];
  const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);      // torch's abs backward: sign(0) = 0
  const float v = s * inv_2m * v_loss[0];
  if (v != 0.f) { atomicAdd(v_scales + 3 * g, v); atomicAdd(v_scales + 3 * g + 1, -v); }   // ids repeat when C > 1
}

// value and gradient in one launch (the joint step: no separate launch, no memset for a number only the log reads); `loss` ACCUMULATES
__global__ void __launch_bounds__(256)
    isotropic_fwd_bwd_kernel(int64_t M, const float *__restrict__ scales, const int64_t *__restrict__ ids, float inv_2m,
                             const float *__restrict__ v_loss, float *__restrict__ loss, float *__restrict__ v_scales, bool det) {
  __shared__ float s_part[4];
  float c = 0.f;
  const float vl = v_loss[0];
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const int64_t g = ids[m];
    const float d = scales[3 * g] - scales[3 * g + 1];
    c += fabsf(d) * inv_2m;
    const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    const float v = s * inv_2m * vl;
    if (v != 0.f) { atomicAdd(v_scales + 3 * g, v); atomicAdd(v_scales + 3 * g + 1, -v); }
  }
  const float ws = wave_sum_to_lane63(c);
  if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = ws;
  __syncthreads();
  finish_scalars((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]), 0.f, loss, nullptr, det ? &g_det_isotropic : nullptr);
}

// ---- NeuralGS::prune_nan_gs's test (neural_gaussian.cpp:907-916): rows with a NaN in offsets / scaling / quaternion
__global__ void __launch_bounds__(256)
    nan_rows_kernel(int64_t n, const float *__restrict__ offsets, const float *__restrict__ scaling,
                    const float *__restrict__ quaternion, int32_t *__restrict__ count, uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (i < n) {
#pragma unroll
    for (int k = 0; k < 3; ++k) bad |= (offsets[3 * i + k] != offsets[3 * i + k]) | (scaling[3 * i + k] != scaling[3 * i + k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) bad |= quaternion[4 * i + k] != quaternion[4 * i + k];
    if (mask != nullptr) mask[i] = bad ? 1 : 0;
  }
  const unsigned long long b = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (int32_t)__popcll(b));
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_splat_activations_fwd(int64_t n, const float *anchors, const float *offsets, const float *log_scales,
                                          const float *logit_opacities, float *xyz, float *scales, float *opacities,
                                          gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_splat_activations_fwd");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(n > 0 && anchors && offsets && log_scales && logit_opacities && xyz && scales && opacities,
               "splat_activations_fwd: bad arguments");
  splat_act_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, anchors, offsets, log_scales, logit_opacities, xyz,
                                                                         scales, opacities);
  GSDF_CHECK_LAUNCH("splat_act_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_splat_activations_bwd(int64_t n, const float *scales, const float *opacities, const float *v_xyz,
                                          const float *v_scales, const float *v_opacities, float *g_offsets,
                                          float *g_log_scales, float *g_logit_opacities, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_splat_activations_bwd");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(n > 0 && scales && opacities && g_offsets && g_log_scales && g_logit_opacities,
               "splat_activations_bwd: bad arguments");
  splat_act_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, scales, opacities, v_xyz, v_scales, v_opacities,
                                                                         g_offsets, g_log_scales, g_logit_opacities);
  GSDF_CHECK_LAUNCH("splat_act_bwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_isotropic_loss_fwd(int64_t M, const float *scales, const int64_t *gaussian_ids, float *loss, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_isotropic_loss_fwd");
  GSDF_REQUIRE(M >= 0 && loss, "isotropic_loss_fwd: bad arguments");
  GSDF_HIP(hipMemsetAsync(loss, 0, 4, stream), "isotropic_loss memset");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(scales && gaussian_ids, "isotropic_loss_fwd: null buffer");
  const int64_t blocks = (M + 255) / 256;
  isotropic_fwd_kernel<<<(unsigned)(blocks > 512 ? 512 : blocks), 256, 0, stream>>>(M, scales, gaussian_ids, 0.5f / (float)M, loss, deterministic());
  GSDF_CHECK_LAUNCH("isotropic_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_isotropic_loss_bwd(int64_t M, const float *scales, const int64_t *gaussian_ids, const float *v_loss, float *v_scales,
                                       gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_isotropic_loss_bwd");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(M > 0 && scales && gaussian_ids && v_loss && v_scales, "isotropic_loss_bwd: bad arguments");
  isotropic_bwd_kernel<<<(unsigned)((M + 255) / 256), 256, 0, stream>>>(M, scales, gaussian_ids, 0.5f / (float)M, v_loss, v_scales);
  GSDF_CHECK_LAUNCH("isotropic_bwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_isotropic_loss_fwd_bwd(int64_t M, const float *scales, const int64_t *gaussian_ids, const float *v_loss, float *loss,
                                           float *v_scales, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_isotropic_loss_fwd_bwd");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(M > 0 && scales && gaussian_ids && v_loss && loss && v_scales, "isotropic_loss_fwd_bwd: bad arguments");
  const int64_t blocks = (M + 255) / 256;
  isotropic_fwd_bwd_kernel<<<(unsigned)(blocks > 1024 ? 1024 : blocks), 256, 0, stream>>>(M, scales, gaussian_ids, 0.5f / (float)M, v_loss, loss, v_scales, deterministic());
  GSDF_CHECK_LAUNCH("isotropic_fwd_bwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_nan_rows_accumulate(int64_t n, const float *offsets, const float *scaling, const float *quaternion, int32_t *total,
                                        gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_nan_rows_accumulate");
  GSDF_REQUIRE(n >= 0 && total, "nan_rows_accumulate: bad arguments");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(offsets && scaling && quaternion, "nan_rows_accumulate: null buffer");
  nan_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, offsets, scaling, quaternion, total, nullptr);
  GSDF_CHECK_LAUNCH("nan_rows_kernel");
  return GSDF_OK;
}

// ---- row gathers / scatters of the visible set (what the reference writes as xyz.index_select(0, gaussian_ids) and index_add_) ----------
namespace gsdf {
__global__ void __launch_bounds__(256)
    visible_gather_kernel(int64_t M, const int64_t *__restrict__ ids, const float *__restrict__ xyz, const float *__restrict__ opac,
                          float *__restrict__ xyz_rows, float *__restrict__ opac_rows, float *__restrict__ ones) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int64_t g = ids[m];
  if (xyz_rows != nullptr) { xyz_rows[3 * m] = xyz[3 * g]; xyz_rows[3 * m + 1] = xyz[3 * g + 1]; xyz_rows[3 * m + 2] = xyz[3 * g + 2]; }
  if (opac_rows != nullptr) opac_rows[m] = opac[g];
  if (ones != nullptr) ones[m] = 1.0f;
}
template <int COLS, bool UNIQUE>
__global__ void __launch_bounds__(256)
    rows_scatter_add_kernel(int64_t M, const int64_t *__restrict__ ids, const float *__restrict__ src, float *__restrict__ dst) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int64_t g = ids[m];
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    if (UNIQUE) dst[COLS * g + c] += src[COLS * m + c];      // every destination row appears once: plain read-modify-write
    else atomicAdd(dst + COLS * g + c, src[COLS * m + c]);
  }
}
}  // namespace gsdf

extern "C" int gsdf_visible_gather(int64_t M, const int64_t *gaussian_ids, const float *xyz, const float *opacities, float *xyz_rows,
                                   float *opacity_rows, float *ones, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_visible_gather");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(M > 0 && gaussian_ids && (!xyz_rows || xyz) && (!opacity_rows || opacities), "visible_gather: bad arguments");
  visible_gather_kernel<<<(unsigned)((M + 255) / 256), 256, 0, stream>>>(M, gaussian_ids, xyz, opacities, xyz_rows, opacity_rows, ones);
  GSDF_CHECK_LAUNCH("visible_gather_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_rows_scatter_add(int64_t M, int cols, const int64_t *ids, int ids_unique, const float *src, float *dst, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_rows_scatter_add");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(M > 0 && ids && src && dst && (cols == 1 || cols == 3), "rows_scatter_add: bad arguments (cols 1 or 3)");
  const unsigned grid = (unsigned)((M + 255) / 256);
  if (cols == 3) {
    if (ids_unique) rows_scatter_add_kernel<3, true><<<grid, 256, 0, stream>>>(M, ids, src, dst);
    else rows_scatter_add_kernel<3, false><<<grid, 256, 0, stream>>>(M, ids, src, dst);
  } else {
    if (ids_unique) rows_scatter_add_kernel<1, true><<<grid, 256, 0, stream>>>(M, ids, src, dst);
    else rows_scatter_add_kernel<1, false><<<grid, 256, 0, stream>>>(M, ids, src, dst);
  }
  GSDF_CHECK_LAUNCH("rows_scatter_add_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_nan_rows(int64_t n, const float *offsets, const float *scaling, const float *quaternion, int32_t *count,
                             uint8_t *mask, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_nan_rows");
  GSDF_REQUIRE(n >= 0 && count, "nan_rows: bad arguments");
  GSDF_HIP(hipMemsetAsync(count, 0, 4, stream), "nan_rows memset");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(offsets && scaling && quaternion, "nan_rows: null buffer");
  nan_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, offsets, scaling, quaternion, count, mask);
  GSDF_CHECK_LAUNCH("nan_rows_kernel");
  return GSDF_OK;
}
