// render_post.hip — fused per-pixel epilogue of rasterization_2dgs_sdf
// (/root/reference/include/neural_gaussian/neural_gaussian.cpp:229-240): expected depth = depth/alpha with
// nan_to_num, cat(colours, depth) and the rotation of the rendered normals to world space
// (normals @ inverse(viewmats)[0,:3,:3]^T).  The reference issues 6 libtorch ops (div, nan_to_num, cat,
// inverse, matmul via a GEMM, index) + their autograd twins; here it is ONE streaming pass each way:
// 44 B read + 28 B written per pixel, HBM-bound.
#include "common.h"

namespace gsdf {

__device__ __forceinline__ void inv3x3_of_viewmat(const float *__restrict__ vm, float R[9]) {
  const float a = vm[0], b = vm[1], c = vm[2], d = vm[4], e = vm[5], f = vm[6], g = vm[8], h = vm[9], i = vm[10];
  const float A00 = e * i - f * h, A01 = c * h - b * i, A02 = b * f - c * e;
  const float A10 = f * g - d * i, A11 = a * i - c * g, A12 = c * d - a * f;
  const float A20 = d * h - e * g, A21 = b * g - a * h, A22 = a * e - b * d;
  const float id = 1.0f / (a * A00 + b * A10 + c * A20);
  R[0] = A00 * id; R[1] = A01 * id; R[2] = A02 * id;
  R[3] = A10 * id; R[4] = A11 * id; R[5] = A12 * id;
  R[6] = A20 * id; R[7] = A21 * id; R[8] = A22 * id;
}

__global__ void __launch_bounds__(256)
    render_post_fwd_kernel(int64_t n_pix, int expected_depth, const float *__restrict__ viewmat0,
                           const float *__restrict__ colors, const float *__restrict__ depths,
                           const float *__restrict__ alphas, const float *__restrict__ normals,
                           float *__restrict__ renders, float *__restrict__ normals_world,
                           float *__restrict__ color3, float *__restrict__ depth1) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  float R[9];
  inv3x3_of_viewmat(viewmat0, R);  // = R_c2w (upper 3x3 of the inverse of a rigid transform)
  float d = depths[p];
  if (expected_depth) {
    d = d / alphas[p];
    if (d != d) d = 0.f;                                   // nan_to_num: nan -> 0, +-inf -> +-FLT_MAX
    else if (d > 3.4028234663852886e38f) d = 3.4028234663852886e38f;
    else if (d < -3.4028234663852886e38f) d = -3.4028234663852886e38f;
  }
  const float c0 = colors[3 * p], c1 = colors[3 * p + 1], c2 = colors[3 * p + 2];
  *reinterpret_cast<float4 *>(renders + 4 * p) = make_float4(c0, c1, c2, d);
  if (color3 != nullptr) { color3[3 * p] = c0; color3[3 * p + 1] = c1; color3[3 * p + 2] = c2; depth1[p] = d; }
  const float nx = normals[3 * p], ny = normals[3 * p + 1], nz = normals[3 * p + 2];
  normals_world[3 * p] = R[0] * nx + R[1] * ny + R[2] * nz;
  normals_world[3 * p + 1] = R[3] * nx + R[4] * ny + R[5] * nz;
  normals_world[3 * p + 2] = R[6] * nx + R[7] * ny + R[8] * nz;
}

__global__ void __launch_bounds__(256)
    render_post_bwd_kernel(int64_t n_pix, int expected_depth, const float *__restrict__ viewmat0,
                           const float *__restrict__ depths, const float *__restrict__ alphas,
                           const float *__restrict__ v_renders, const float *__restrict__ v_normals_world,
                           const float *__restrict__ v_color3, const float *__restrict__ v_depth1,
                           float *__restrict__ v_colors, float *__restrict__ v_depths, float *__restrict__ v_alphas,
                           float *__restrict__ v_normals) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  float R[9];
  inv3x3_of_viewmat(viewmat0, R);
  float4 v = v_renders != nullptr ? *reinterpret_cast<const float4 *>(v_renders + 4 * p) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (v_color3 != nullptr) { v.x += v_color3[3 * p]; v.y += v_color3[3 * p + 1]; v.z += v_color3[3 * p + 2]; }
  if (v_depth1 != nullptr) v.w += v_depth1[p];
  v_colors[3 * p] = v.x; v_colors[3 * p + 1] = v.y; v_colors[3 * p + 2] = v.z;
  float vd = v.w, va = 0.f;
  if (expected_depth) {
    const float a = alphas[p], q = depths[p] / a;
    const bool finite = (q == q) && fabsf(q) <= 3.4028234663852886e38f && a != 0.f;
    vd = finite ? v.w / a : 0.f;     // pixels with alpha == 0 have no contributor: their gradient is dropped
    va = finite ? -v.w * q / a : 0.f;
  }
  v_depths[p] = vd;
  v_alphas[p] = va;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (v_normals_world != nullptr) { gx = v_normals_world[3 * p]; gy = v_normals_world[3 * p + 1]; gz = v_normals_world[3 * p + 2]; }
  v_normals[3 * p] = R[0] * gx + R[3] * gy + R[6] * gz;
  v_normals[3 * p + 1] = R[1] * gx + R[4] * gy + R[7] * gz;
  v_normals[3 * p + 2] = R[2] * gx + R[5] * gy + R[8] * gz;
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_render_post_fwd(int64_t n_pix, int expected_depth, const float *viewmat0, const float *render_colors,
                                    const float *render_depths, const float *render_alphas,
                                    const float *render_normals, float *renders, float *normals_world, float *color3,
                                    float *depth1, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_render_post_fwd");
  if (n_pix == 0) return GSDF_OK;
  GSDF_REQUIRE(viewmat0 && render_colors && render_depths && render_alphas && render_normals && renders && normals_world,
               "render_post_fwd: null buffer");
  GSDF_REQUIRE((color3 == nullptr) == (depth1 == nullptr), "render_post_fwd: color3 and depth1 go together");
  render_post_fwd_kernel<<<(unsigned)((n_pix + 255) / 256), 256, 0, stream>>>(n_pix, expected_depth, viewmat0, render_colors,
                                                                            render_depths, render_alphas, render_normals,
                                                                            renders, normals_world, color3, depth1);
  GSDF_CHECK_LAUNCH("render_post_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_render_post_bwd(int64_t n_pix, int expected_depth, const float *viewmat0, const float *render_depths,
                                    const float *render_alphas, const float *v_renders, const float *v_normals_world,
                                    const float *v_color3, const float *v_depth1, float *v_render_colors, float *v_render_depths, float *v_render_alphas,
                                    float *v_render_normals, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_render_post_bwd");
  if (n_pix == 0) return GSDF_OK;
  GSDF_REQUIRE(viewmat0 && render_depths && render_alphas && v_render_colors && v_render_depths && v_render_alphas &&
                   v_render_normals,
               "render_post_bwd: null buffer");
  render_post_bwd_kernel<<<(unsigned)((n_pix + 255) / 256), 256, 0, stream>>>(n_pix, expected_depth, viewmat0, render_depths,
                                                                            render_alphas, v_renders, v_normals_world,
                                                                            v_color3, v_depth1, v_render_colors, v_render_depths,
                                                                            v_render_alphas, v_render_normals);
  GSDF_CHECK_LAUNCH("render_post_bwd_kernel");
  return GSDF_OK;
}
