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

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_splat_activations_fwd(int64_t n, const float *anchors, const float *offsets, const float *log_scales,
                                          const float *logit_opacities, float *xyz, float *scales, float *opacities,
                                          gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
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
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(n > 0 && scales && opacities && g_offsets && g_log_scales && g_logit_opacities,
               "splat_activations_bwd: bad arguments");
  splat_act_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, scales, opacities, v_xyz, v_scales, v_opacities,
                                                                         g_offsets, g_log_scales, g_logit_opacities);
  GSDF_CHECK_LAUNCH("splat_act_bwd_kernel");
  return GSDF_OK;
}
