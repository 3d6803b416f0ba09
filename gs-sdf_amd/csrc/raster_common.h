// raster_common.h — shared pieces of the 2DGS compositing kernels (SPEC A.4 / A.5).
// Reference operator: rasterize_to_pixels_2dgs, called at
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223.
//
// Workgroup = one 16x16 pixel tile = 256 lanes = 4 wave64, each wave an 8x8 pixel quadrant.  The tile's depth-sorted splat list is
// staged through LDS in batches (one splat per lane, copied from the packed per-splat records of raster_pack.hip).
//
// Per splat the pack pass precomputes the AFFINE form of the ray-splat cross product.  With
// h_u = p_x M_w - M_u, h_v = p_y M_w - M_v the vector z = h_u x h_v is exactly affine in the pixel:
//     z(p) = C0 + (p_x - m_x) A + (p_y - m_y) B,   A = M_v x M_w,  B = M_w x M_u,  C0 = z(m),  m = mean2d
// (the p_x p_y term is M_w x M_w = 0), and the intersection depth s.M_w.xy + M_w.z equals D / z.z with
// D = z . M_w = det(M) constant per splat.  The expansion point is the splat's OWN projected centre: there h_u.z and h_v.z
// vanish (the centre ray hits the splat at (u, v) = 0), so C0.xy is small and z.xy(p) = C0.xy + dx A.xy + dy B.xy carries no
// cancellation — s = z.xy / z.z keeps its RELATIVE accuracy down to the pixels next to the centre, where the gradient
// v_sigma * s is proportional to it.  (Round 2 expanded about the tile's first pixel: z.xy(p) was then the difference of
// terms ~|C0| ~ 500 x larger, an absolute error of 5e-5 in s, i.e. 2e-3 of the whole gradient of a splat whose only
// 3-D-branch pixel sits next to its centre.)  p - m is exact in fp32 (Sterbenz) and is needed for the low-pass term anyway,
// so a pixel costs 6 FMAs and the depth one multiply.  One staged splat = 5 LDS vectors (80 B) + the 64-bit reach mask:
//   q0 = (A.x, A.y, A.z, mean2d.x)   q1 = (B.x, B.y, B.z, mean2d.y)   q2 = (C0.x, C0.y, C0.z, opacity)
//   q3 = (D, M_w.z, r, g)            q4 = (b, n.x, n.y, n.z)
// The forward composites the depth as D / z.z (one multiply).  The BACKWARD must not differentiate that form: d(D / z.z) splits
// into (1 / z.z) dD/dM and -(D / z.z^2) dz.z/dM, two terms ~1e3 x larger than their sum that would be accumulated in separate
// fp32 sums (measured: 1e-3 relative error of dL/dM on small splats).  It stages M_w.x, M_w.y instead of D and
// differentiates dep = s . M_w.xy + M_w.z as the specification writes it (SPEC A.5).
#pragma once
#include <stdlib.h>

#include "common.h"
#include "reach_mask.h"

namespace gsdf {

static constexpr int TILE = 16;
static constexpr int RT = 256;  // threads per workgroup
static constexpr float ALPHA_MIN = 1.0f / 255.0f;
static constexpr float ALPHA_MAX = 0.999f;
static constexpr float T_EPS = 1e-4f;
static constexpr float FILTER_INV_SQUARE = 2.0f;

// XCD-aware tile assignment: workgroup b runs on XCD (b % 8); give each XCD one contiguous band
// of tiles so that the splats shared by neighbouring tiles stay in that XCD's 4 MiB L2.
// n_xcd = number of XCDs the launching queue may use (8, or fewer under a CU mask: workgroups are then dealt round-robin
// over the enabled XCDs only).
__device__ __forceinline__ int64_t xcd_tile_index(int64_t total_tiles, int n_xcd) {
  const int64_t b = blockIdx.x;
  const int64_t chunk = gridDim.x / n_xcd;
  return (b % n_xcd) * chunk + (b / n_xcd);
}
static inline unsigned xcd_grid(int64_t total_tiles, int n) { return (unsigned)(((total_tiles + n - 1) / n) * n); }

struct PairEval {
  float zx, zy, zz, inv;
  float sx, sy, dx, dy;
  float vis, alpha, dep;
  bool b3, ok, clamped;
};

// Evaluates one (pixel, splat) pair.  (lx, ly) = pixel offset from the tile's first pixel; (px, py) = pixel centre.
// `ok` is false when the pair does not contribute.  depth: forward q3.x = D, q3.y = M_w.z (dep = D / z.z);
// backward (BWD) q3.x = M_w.x, mwy = M_w.y (dep = s . M_w.xy + M_w.z).
template <bool BWD = false>
__device__ __forceinline__ void eval_pair(float lx, float ly, float px, float py, const float4 &a0, const float4 &a1,
                                          const float4 &a2, float D, float mwz, PairEval &e, float mwy = 0.f) {
  (void)lx; (void)ly;
  e.dx = a0.w - px; e.dy = a1.w - py;              // mean2d - p (exact)
  e.zx = fmaf(-e.dy, a1.x, fmaf(-e.dx, a0.x, a2.x));
  e.zy = fmaf(-e.dy, a1.y, fmaf(-e.dx, a0.y, a2.y));
  e.zz = fmaf(-e.dy, a1.z, fmaf(-e.dx, a0.z, a2.z));
  e.inv = __builtin_amdgcn_rcpf(e.zz);
  e.sx = e.zx * e.inv; e.sy = e.zy * e.inv;
  // explicit FMAs: every instantiation of the compositing kernels (forward, backward, instrumented) must take bit-identical decisions
  const float g3 = fmaf(e.sx, e.sx, e.sy * e.sy);
  const float g2 = FILTER_INV_SQUARE * fmaf(e.dx, e.dx, e.dy * e.dy);
  e.b3 = g3 <= g2;
  const float sigma = 0.5f * (e.b3 ? g3 : g2);
  e.vis = __expf(-sigma);
  const float a = a2.w * e.vis;
  e.clamped = a > ALPHA_MAX;
  e.alpha = fminf(ALPHA_MAX, a);
  e.ok = (e.zz != 0.0f) && (sigma >= 0.0f) && (e.alpha >= ALPHA_MIN);
  e.dep = e.b3 ? (BWD ? fmaf(e.sx, D, fmaf(e.sy, mwy, mwz)) : D * e.inv) : mwz;
}

}  // namespace gsdf
