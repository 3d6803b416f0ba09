// raster_common.h — shared pieces of the 2DGS compositing kernels (SPEC A.4 / A.5).
// Reference operator: rasterize_to_pixels_2dgs, called at
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223.
//
// Workgroup = one 16x16 pixel tile = 256 lanes = 4 wave64.  Each wave owns an 8x8 pixel quadrant
// (lane -> (lane&7, lane>>3)) rather than a 16x4 strip: the alpha test and the early-termination
// branch are per-lane, and a compact square keeps the lanes of one wave on the same splats.
// The tile's depth-sorted splat list is staged through LDS in batches of 256 (one splat per lane,
// gathered from the packed per-splat arrays); every lane then reads the same LDS address
// (broadcast, conflict-free ds_read_b128).
#pragma once
#include "common.h"

namespace gsdf {

static constexpr int TILE = 16;
static constexpr int RT = 256;  // threads per workgroup
static constexpr float ALPHA_MIN = 1.0f / 255.0f;
static constexpr float ALPHA_MAX = 0.999f;
static constexpr float T_EPS = 1e-4f;
static constexpr float FILTER_INV_SQUARE = 2.0f;

// One staged splat = 5 LDS vectors (80 B):
//   q0 = (Mu.x, Mu.y, Mu.z, mean2d.x)   q1 = (Mv.x, Mv.y, Mv.z, mean2d.y)
//   q2 = (Mw.x, Mw.y, Mw.z, opacity)    q3 = (r, g, b, n.x)   q4 = (n.y, n.z)
struct SplatBatch {
  float4 q0[RT], q1[RT], q2[RT], q3[RT];
  float2 q4[RT];
};

__device__ __forceinline__ void stage_splat(SplatBatch &s, int slot, int g, const float *__restrict__ means2d,
                                            const float *__restrict__ ray_transforms,
                                            const float *__restrict__ colors, const float *__restrict__ opacities,
                                            const float *__restrict__ normals) {
  const float *m = ray_transforms + 9 * (int64_t)g;
  const float2 xy = *reinterpret_cast<const float2 *>(means2d + 2 * (int64_t)g);
  const float *c = colors + 3 * (int64_t)g;
  const float *n = normals + 3 * (int64_t)g;
  s.q0[slot] = make_float4(m[0], m[1], m[2], xy.x);
  s.q1[slot] = make_float4(m[3], m[4], m[5], xy.y);
  s.q2[slot] = make_float4(m[6], m[7], m[8], opacities[g]);
  s.q3[slot] = make_float4(c[0], c[1], c[2], n[0]);
  s.q4[slot] = make_float2(n[1], n[2]);
}

// XCD-aware tile assignment: workgroup b runs on XCD (b % 8); give each XCD one contiguous band
// of tiles so that the splats shared by neighbouring tiles stay in that XCD's 4 MiB L2.
__device__ __forceinline__ int64_t xcd_tile_index(int64_t total_tiles) {
  const int64_t b = blockIdx.x;
  const int64_t chunk = gridDim.x / 8;
  return (b & 7) * chunk + (b >> 3);
}
static inline unsigned xcd_grid(int64_t total_tiles) { return (unsigned)(((total_tiles + 7) / 8) * 8); }

struct PairEval {
  float hux, huy, huz, hvx, hvy, hvz;
  float zx, zy, zz;
  float sx, sy, dx, dy;
  float vis, alpha, dep;
  bool b3, ok, clamped;
};

// Evaluates one (pixel, splat) pair.  `ok` is false when the pair does not contribute.
__device__ __forceinline__ void eval_pair(float px, float py, const float4 &a0, const float4 &a1, const float4 &a2,
                                          PairEval &e) {
  e.hux = px * a2.x - a0.x; e.huy = px * a2.y - a0.y; e.huz = px * a2.z - a0.z;
  e.hvx = py * a2.x - a1.x; e.hvy = py * a2.y - a1.y; e.hvz = py * a2.z - a1.z;
  e.zx = e.huy * e.hvz - e.huz * e.hvy;
  e.zy = e.huz * e.hvx - e.hux * e.hvz;
  e.zz = e.hux * e.hvy - e.huy * e.hvx;
  const float inv = __builtin_amdgcn_rcpf(e.zz);
  e.sx = e.zx * inv; e.sy = e.zy * inv;
  const float g3 = e.sx * e.sx + e.sy * e.sy;
  e.dx = a0.w - px; e.dy = a1.w - py;
  const float g2 = FILTER_INV_SQUARE * (e.dx * e.dx + e.dy * e.dy);
  e.b3 = g3 <= g2;
  const float sigma = 0.5f * (e.b3 ? g3 : g2);
  e.vis = __expf(-sigma);
  const float a = a2.w * e.vis;
  e.clamped = a > ALPHA_MAX;
  e.alpha = fminf(ALPHA_MAX, a);
  e.ok = (e.zz != 0.0f) && (sigma >= 0.0f) && (e.alpha >= ALPHA_MIN);
  e.dep = e.b3 ? (e.sx * a2.x + e.sy * a2.y) + a2.z : a2.z;
}

}  // namespace gsdf
