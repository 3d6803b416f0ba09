// view_colors.hip — P2 / P2': view-dependent colours from spherical harmonics.
// Replaces gsplat_cpp::get_view_colors (call site
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:199-200): dirs = mean - campos,
// rgb = clamp_min(SH(dir).coeffs + 0.5, 0) over the active degree.  One lane per visible splat;
// HBM-bound gather of 12*K bytes per splat.
#include "common.h"

namespace gsdf {

static constexpr int VT = 256;

__device__ __forceinline__ void cam_pos(const float *__restrict__ vm, float cp[3]) {
  // campos = inverse(viewmat)[:3,3] = -A^-1 t (adjugate inverse; A is only approximately orthonormal)
  const float a = vm[0], b = vm[1], c = vm[2], d = vm[4], e = vm[5], f = vm[6], g = vm[8], h = vm[9], i = vm[10];
  const float A00 = e * i - f * h, A01 = c * h - b * i, A02 = b * f - c * e;
  const float A10 = f * g - d * i, A11 = a * i - c * g, A12 = c * d - a * f;
  const float A20 = d * h - e * g, A21 = b * g - a * h, A22 = a * e - b * d;
  const float id = 1.0f / (a * A00 + b * A10 + c * A20);
  const float t0 = vm[3], t1 = vm[7], t2 = vm[11];
  cp[0] = -((A00 * t0 + A01 * t1) + A02 * t2) * id;
  cp[1] = -((A10 * t0 + A11 * t1) + A12 * t2) * id;
  cp[2] = -((A20 * t0 + A21 * t1) + A22 * t2) * id;
}

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,  -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float *b) {
  b[0] = SH_C0;
  if (DEG >= 1) { b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x; }
  if (DEG >= 2) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2 * zz - xx - yy);
    b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
    if (DEG >= 3) {
      b[9] = SH_C3[0] * y * (3 * xx - yy); b[10] = SH_C3[1] * xy * z;
      b[11] = SH_C3[2] * y * (4 * zz - xx - yy); b[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
      b[13] = SH_C3[4] * x * (4 * zz - xx - yy); b[14] = SH_C3[5] * z * (xx - yy);
      b[15] = SH_C3[6] * x * (xx - 3 * yy);
    }
  }
}

// d basis_k / d (x,y,z) contracted with g_k (k-th coefficient gradient weight): returns v_dir
template <int DEG>
__device__ __forceinline__ void sh_basis_vjp(float x, float y, float z, const float *g, float vd[3]) {
  vd[0] = vd[1] = vd[2] = 0.f;
  if (DEG >= 1) { vd[1] += -SH_C1 * g[1]; vd[2] += SH_C1 * g[2]; vd[0] += -SH_C1 * g[3]; }
  if (DEG >= 2) {
    const float xx = x * x, yy = y * y, zz = z * z;
    vd[0] += SH_C2[0] * y * g[4];           vd[1] += SH_C2[0] * x * g[4];
    vd[1] += SH_C2[1] * z * g[5];           vd[2] += SH_C2[1] * y * g[5];
    vd[0] += SH_C2[2] * (-2 * x) * g[6];    vd[1] += SH_C2[2] * (-2 * y) * g[6];  vd[2] += SH_C2[2] * (4 * z) * g[6];
    vd[0] += SH_C2[3] * z * g[7];           vd[2] += SH_C2[3] * x * g[7];
    vd[0] += SH_C2[4] * (2 * x) * g[8];     vd[1] += SH_C2[4] * (-2 * y) * g[8];
    if (DEG >= 3) {
      vd[0] += SH_C3[0] * 6 * x * y * g[9];             vd[1] += SH_C3[0] * (3 * xx - 3 * yy) * g[9];
      vd[0] += SH_C3[1] * y * z * g[10];                vd[1] += SH_C3[1] * x * z * g[10];   vd[2] += SH_C3[1] * x * y * g[10];
      vd[0] += SH_C3[2] * (-2 * x * y) * g[11];         vd[1] += SH_C3[2] * (4 * zz - xx - 3 * yy) * g[11];
      vd[2] += SH_C3[2] * 8 * y * z * g[11];
      vd[0] += SH_C3[3] * (-6 * x * z) * g[12];         vd[1] += SH_C3[3] * (-6 * y * z) * g[12];
      vd[2] += SH_C3[3] * (6 * zz - 3 * xx - 3 * yy) * g[12];
      vd[0] += SH_C3[4] * (4 * zz - 3 * xx - yy) * g[13]; vd[1] += SH_C3[4] * (-2 * x * y) * g[13];
      vd[2] += SH_C3[4] * 8 * x * z * g[13];
      vd[0] += SH_C3[5] * 2 * x * z * g[14];            vd[1] += SH_C3[5] * (-2 * y * z) * g[14];
      vd[2] += SH_C3[5] * (xx - yy) * g[14];
      vd[0] += SH_C3[6] * (3 * xx - 3 * yy) * g[15];    vd[1] += SH_C3[6] * (-6 * x * y) * g[15];
    }
  }
}

template <int DEG>
__global__ void __launch_bounds__(VT)
    view_colors_fwd_kernel(int64_t M, int64_t K, const float *__restrict__ viewmats, const float *__restrict__ means,
                           const float *__restrict__ sh, const int64_t *__restrict__ camera_ids,
                           const int64_t *__restrict__ gaussian_ids, float *__restrict__ colors) {
  const int64_t m = (int64_t)blockIdx.x * VT + threadIdx.x;
  if (m >= M) return;
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const int64_t n = gaussian_ids[m];
  float b[NB];
  if (DEG > 0) {
    float cp[3];
    cam_pos(viewmats + 16 * camera_ids[m], cp);
    const float dx = means[3 * n] - cp[0], dy = means[3 * n + 1] - cp[1], dz = means[3 * n + 2] - cp[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float inv = len > 0 ? 1.0f / len : 0.f;
    sh_basis<DEG>(dx * inv, dy * inv, dz * inv, b);
  } else {
    b[0] = SH_C0;
  }
  const float *co = sh + n * K * 3;
  float r = 0.f, g = 0.f, bl = 0.f;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    r += b[k] * co[3 * k];
    g += b[k] * co[3 * k + 1];
    bl += b[k] * co[3 * k + 2];
  }
  colors[3 * m] = fmaxf(r + 0.5f, 0.f);
  colors[3 * m + 1] = fmaxf(g + 0.5f, 0.f);
  colors[3 * m + 2] = fmaxf(bl + 0.5f, 0.f);
}

template <int DEG, bool ATOMIC>
__global__ void __launch_bounds__(VT)
    view_colors_bwd_kernel(int64_t M, int64_t K, const float *__restrict__ viewmats, const float *__restrict__ means,
                           const float *__restrict__ sh, const int64_t *__restrict__ camera_ids,
                           const int64_t *__restrict__ gaussian_ids, const float *__restrict__ v_colors,
                           float *__restrict__ v_sh, float *__restrict__ v_means) {
  const int64_t m = (int64_t)blockIdx.x * VT + threadIdx.x;
  if (m >= M) return;
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const int64_t n = gaussian_ids[m];
  float b[NB], ux = 0.f, uy = 0.f, uz = 0.f, inv = 0.f;
  if (DEG > 0) {
    float cp[3];
    cam_pos(viewmats + 16 * camera_ids[m], cp);
    const float dx = means[3 * n] - cp[0], dy = means[3 * n + 1] - cp[1], dz = means[3 * n + 2] - cp[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    inv = len > 0 ? 1.0f / len : 0.f;
    ux = dx * inv; uy = dy * inv; uz = dz * inv;
    sh_basis<DEG>(ux, uy, uz, b);
  } else {
    b[0] = SH_C0;
  }
  const float *co = sh + n * K * 3;
  float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    acc3[0] += b[k] * co[3 * k];
    acc3[1] += b[k] * co[3 * k + 1];
    acc3[2] += b[k] * co[3 * k + 2];
  }
  float gk[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) gk[k] = 0.f;
  float *vs = v_sh + n * K * 3;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    if (!(acc3[ch] + 0.5f > 0.f)) continue;
    const float g = v_colors[3 * m + ch];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (ATOMIC) atomicAdd(vs + 3 * k + ch, g * b[k]); else vs[3 * k + ch] += g * b[k];
      gk[k] += g * co[3 * k + ch];
    }
  }
  if (DEG > 0) {
    float vu[3];
    sh_basis_vjp<DEG>(ux, uy, uz, gk, vu);
    const float dot = vu[0] * ux + vu[1] * uy + vu[2] * uz;
    const float o0 = (vu[0] - dot * ux) * inv, o1 = (vu[1] - dot * uy) * inv, o2 = (vu[2] - dot * uz) * inv;
    if (ATOMIC) {
      atomicAdd(v_means + 3 * n, o0); atomicAdd(v_means + 3 * n + 1, o1); atomicAdd(v_means + 3 * n + 2, o2);
    } else {
      v_means[3 * n] += o0; v_means[3 * n + 1] += o1; v_means[3 * n + 2] += o2;
    }
  }
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_view_colors_fwd(int64_t M, int64_t K, int sh_degree, const float *viewmats, const float *means,
                                    const float *sh_coeffs, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                    float *colors, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_view_colors_fwd");
  GSDF_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "view_colors: sh_degree %d not in [0,3]", sh_degree);
  GSDF_REQUIRE((sh_degree + 1) * (sh_degree + 1) <= K, "view_colors: degree %d needs %d bases, have %ld", sh_degree,
               (sh_degree + 1) * (sh_degree + 1), (long)K);
  if (M == 0) return GSDF_OK;
  const unsigned nb = (unsigned)((M + VT - 1) / VT);
#define LAUNCH(D) view_colors_fwd_kernel<D><<<nb, VT, 0, stream>>>(M, K, viewmats, means, sh_coeffs, camera_ids, gaussian_ids, colors)
  switch (sh_degree) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; default: LAUNCH(3); }
#undef LAUNCH
  GSDF_CHECK_LAUNCH("view_colors_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_view_colors_bwd(int64_t M, int64_t K, int sh_degree, const float *viewmats, const float *means,
                                    const float *sh_coeffs, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                    const float *v_colors, float *v_sh, float *v_means, int unique_gaussians,
                                    gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_view_colors_bwd");
  GSDF_REQUIRE(sh_degree >= 0 && sh_degree <= 3, "view_colors_bwd: sh_degree %d not in [0,3]", sh_degree);
  GSDF_REQUIRE((sh_degree + 1) * (sh_degree + 1) <= K, "view_colors_bwd: degree %d needs more than %ld bases",
               sh_degree, (long)K);
  if (M == 0) return GSDF_OK;
  const unsigned nb = (unsigned)((M + VT - 1) / VT);
#define LAUNCH(D, A) view_colors_bwd_kernel<D, A><<<nb, VT, 0, stream>>>(M, K, viewmats, means, sh_coeffs, camera_ids, gaussian_ids, v_colors, v_sh, v_means)
  if (unique_gaussians) {
    switch (sh_degree) { case 0: LAUNCH(0, false); break; case 1: LAUNCH(1, false); break; case 2: LAUNCH(2, false); break; default: LAUNCH(3, false); }
  } else {
    switch (sh_degree) { case 0: LAUNCH(0, true); break; case 1: LAUNCH(1, true); break; case 2: LAUNCH(2, true); break; default: LAUNCH(3, true); }
  }
#undef LAUNCH
  GSDF_CHECK_LAUNCH("view_colors_bwd_kernel");
  return GSDF_OK;
}
