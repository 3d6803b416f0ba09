// proj_math.h — per-splat 2DGS projection math (SPEC A.1 / A.6 in DESIGN.md).
// Reference operator: fully_fused_projection_2dgs, called at
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:188-192.
// Translation units including this header are compiled with -ffp-contract=off: the radius
// `ceil(3*sqrt(.))` and the tile rectangle derived from it are integer outputs that must be
// reproducible bit-for-bit, so no FMA contraction and a fixed left-to-right operation order.
#pragma once
#include "common.h"

namespace gsdf {

struct Proj {
  float mc[3];
  float Rq[9];
  float Rc[9];
  float qn[4];
  float inv_norm;
  float Mu[3], Mv[3], Mw[3];
  float f[3];
  float mean2d[2];
  float radius;
  float mult;
};

__device__ __forceinline__ void quat_to_rotmat(const float q[4], float qn[4], float &inv_norm, float R[9]) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  const float n2 = ((w * w + x * x) + y * y) + z * z;
  const float inv = 1.0f / sqrtf(n2);
  w *= inv; x *= inv; y *= inv; z *= inv;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z;
  inv_norm = inv;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// returns false when the splat is culled.  `cull` = false skips the depth/radius/screen tests
// (backward pass: rows are known to be visible).
template <bool CULL>
__device__ __forceinline__ bool project_splat(const float mean[3], const float quat[4], float su, float sv,
                                              const float *__restrict__ vm, const float *__restrict__ K, int W,
                                              int H, float near_p, float far_p, float radius_clip, Proj &o) {
  const float R00 = vm[0], R01 = vm[1], R02 = vm[2], t0 = vm[3];
  const float R10 = vm[4], R11 = vm[5], R12 = vm[6], t1 = vm[7];
  const float R20 = vm[8], R21 = vm[9], R22 = vm[10], t2 = vm[11];
  o.mc[0] = ((R00 * mean[0] + R01 * mean[1]) + R02 * mean[2]) + t0;
  o.mc[1] = ((R10 * mean[0] + R11 * mean[1]) + R12 * mean[2]) + t1;
  o.mc[2] = ((R20 * mean[0] + R21 * mean[1]) + R22 * mean[2]) + t2;
  if (CULL && (o.mc[2] < near_p || o.mc[2] > far_p)) return false;
  quat_to_rotmat(quat, o.qn, o.inv_norm, o.Rq);
  const float *q = o.Rq;
  float *c = o.Rc;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    c[0 + j] = (R00 * q[0 + j] + R01 * q[3 + j]) + R02 * q[6 + j];
    c[3 + j] = (R10 * q[0 + j] + R11 * q[3 + j]) + R12 * q[6 + j];
    c[6 + j] = (R20 * q[0 + j] + R21 * q[3 + j]) + R22 * q[6 + j];
  }
  const float H0[3] = {su * c[0], sv * c[1], o.mc[0]};
  const float H1[3] = {su * c[3], sv * c[4], o.mc[1]};
  const float H2[3] = {su * c[6], sv * c[7], o.mc[2]};
  const float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.Mu[j] = fx * H0[j] + cx * H2[j];
    o.Mv[j] = fy * H1[j] + cy * H2[j];
    o.Mw[j] = H2[j];
  }
  const float *Mu = o.Mu, *Mv = o.Mv, *Mw = o.Mw;
  const float dist = (Mw[0] * Mw[0] + Mw[1] * Mw[1]) - Mw[2] * Mw[2];
  if (dist == 0.0f) return false;
  const float inv = 1.0f / dist;
  o.f[0] = inv; o.f[1] = inv; o.f[2] = -inv;
  const float *f = o.f;
  o.mean2d[0] = (f[0] * Mu[0] * Mw[0] + f[1] * Mu[1] * Mw[1]) + f[2] * Mu[2] * Mw[2];
  o.mean2d[1] = (f[0] * Mv[0] * Mw[0] + f[1] * Mv[1] * Mw[1]) + f[2] * Mv[2] * Mw[2];
  const float dotv = (-c[2]) * o.mc[0] + (-c[5]) * o.mc[1] + (-c[8]) * o.mc[2];
  o.mult = dotv > 0 ? 1.0f : -1.0f;
  if (CULL) {
    const float tx = (f[0] * Mu[0] * Mu[0] + f[1] * Mu[1] * Mu[1]) + f[2] * Mu[2] * Mu[2];
    const float ty = (f[0] * Mv[0] * Mv[0] + f[1] * Mv[1] * Mv[1]) + f[2] * Mv[2] * Mv[2];
    const float hx = o.mean2d[0] * o.mean2d[0] - tx;
    const float hy = o.mean2d[1] * o.mean2d[1] - ty;
    const float ext = fmaxf(1e-4f, fmaxf(hx, hy));
    const float radius = ceilf(3.0f * sqrtf(ext));
    if (!(radius > radius_clip)) return false;
    if (o.mean2d[0] + radius <= 0 || o.mean2d[0] - radius >= (float)W || o.mean2d[1] + radius <= 0 ||
        o.mean2d[1] - radius >= (float)H)
      return false;
    if (!(radius < 2147483000.0f)) return false;
    o.radius = radius;
  }
  return true;
}

// counter-based N(0,1)^2 pair for the stochastic splat sample (SPEC S-3).  seed == 0 -> (0,0).
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ void sample_eps(uint64_t seed, uint32_t gid, float &eu, float &ev) {
  if (seed == 0) { eu = 0.f; ev = 0.f; return; }
  const uint32_t h1 = mix32(gid * 2u + 0x9E3779B9u * (uint32_t)seed + (uint32_t)(seed >> 32));
  const uint32_t h2 = mix32(h1 ^ 0x68E31DA4u);
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  eu = r * cs; ev = r * sn;
}

}  // namespace gsdf
