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

// Per (tile, splat) the staging lane precomputes the AFFINE form of the ray-splat cross product.  With
// h_u = p_x M_w - M_u, h_v = p_y M_w - M_v the vector z = h_u x h_v is exactly affine in the pixel:
//     z(p) = C0 + (p_x - m_x) A + (p_y - m_y) B,   A = M_v x M_w,  B = M_w x M_u,  C0 = z(m),  m = mean2d
// (the p_x p_y term is M_w x M_w = 0), and the intersection depth s.M_w.xy + M_w.z equals D / z.z with
// D = z . M_w = det(M) constant per splat.  The expansion point is the splat's OWN projected centre: there h_u.z and h_v.z
// vanish (the centre ray hits the splat at (u, v) = 0), so C0.xy is small and z.xy(p) = C0.xy + dx A.xy + dy B.xy carries no
// cancellation — s = z.xy / z.z keeps its RELATIVE accuracy down to the pixels next to the centre, where the gradient
// v_sigma * s is proportional to it.  (Round 2 expanded about the tile's first pixel: z.xy(p) was then the difference of
// terms ~|C0| ~ 500 x larger, an absolute error of 5e-5 in s, i.e. 2e-3 of the whole gradient of a splat whose only
// 3-D-branch pixel sits next to its centre.)  p - m is exact in fp32 (Sterbenz) and is needed for the low-pass term anyway,
// so a pixel still costs 6 FMAs and the depth one multiply.  One staged splat = 5 LDS vectors (80 B) + a 4-bit quadrant mask:
//   q0 = (A.x, A.y, A.z, mean2d.x)   q1 = (B.x, B.y, B.z, mean2d.y)   q2 = (C0.x, C0.y, C0.z, opacity)
//   q3 = (D, M_w.z, r, g)            q4 = (b, n.x, n.y, n.z)
// The forward composites the depth as D / z.z (one multiply).  The BACKWARD must not differentiate that form: d(D / z.z) splits
// into (1 / z.z) dD/dM and -(D / z.z^2) dz.z/dM, two terms ~1e3 x larger than their sum that would be accumulated in separate
// fp32 sums (measured: 1e-3 relative error of dL/dM on small splats).  It stages M_w.x, M_w.y instead of D (q3.x, extra) and
// differentiates dep = s . M_w.xy + M_w.z as the specification writes it (SPEC A.5).
template <int CAP, bool BWD>
struct SplatBatchT {
  float4 q0[CAP], q1[CAP], q2[CAP], q3[CAP], q4[CAP];
  float extra[BWD ? CAP : 1];  // backward: M_w.y (q3.x = M_w.x instead of D)
  unsigned short m16[CAP];     // bit 4 q + s set <=> the splat's conservative box reaches the 4x4-pixel sub-block s of wave q's quadrant
                               // (subblock_mask4x4).  EVERY compositing kernel takes its skip decisions from this one mask — a quadrant is
                               // visited when its nibble is non-zero, a pixel blends only when its own sub-block bit is set — so that the forward
                               // and the backward drop a pair in exactly the same pixels even where the box is not conservative (the backward
                               // replays the forward's transmittance and must see the same blended set)
};
using SplatBatch = SplatBatchT<RT, false>;

// Conservative per-quadrant reach test, evaluated ONCE per (tile, splat) by the staging lane.
// A pair contributes only if alpha = o*exp(-min(g3,g2)/2) >= 1/255, i.e. min(g3,g2) <= tau = 2 ln(255 o).
//   {g3 <= tau}: projection of the splat disk u^2+v^2 <= tau.  Its exact screen bounding box follows from the
//     dual conic W diag(1,1,-1/tau) W^T (same construction as the 3-sigma box of SPEC A.1): with
//     D = (1,1,-1/tau), d = M_w.D.M_w (< 0 for an ellipse), c = (M_u.D.M_w)/d, h^2 = c^2 - (M_u.D.M_u)/d.
//   {g2 <= tau}: screen disk of radius sqrt(tau/2) around mean2d (the low-pass branch).
// The box is the union of both, grown by 0.3 px (fp32 cancellation in h^2 is < 0.1 px for |c| < 4096).
// Hyperbolic / degenerate conics (d >= 0) are not culled.  Skipping an unreachable quadrant removes only
// pairs whose alpha test would have failed, so results are unchanged.
__device__ __forceinline__ unsigned quadrant_mask(const float *__restrict__ m, float mx, float my, float opac,
                                                  float tile_x0, float tile_y0) {
  const float o255 = 255.0f * opac;
  if (!(o255 > 1.0f)) return 0u;  // alpha < 1/255 everywhere
  const float tau = 2.0f * __logf(o255) * 1.0001f + 1e-4f;
  const float r2 = sqrtf(0.5f * tau);
  float x0 = mx - r2, x1 = mx + r2, y0 = my - r2, y1 = my + r2;
  const float it = 1.0f / tau;
  const float d = m[6] * m[6] + m[7] * m[7] - it * m[8] * m[8];
  bool bounded = d < 0.0f;
  if (bounded) {
    const float id = 1.0f / d;
    const float cx = (m[0] * m[6] + m[1] * m[7] - it * m[2] * m[8]) * id;
    const float cy = (m[3] * m[6] + m[4] * m[7] - it * m[5] * m[8]) * id;
    const float hx2 = cx * cx - (m[0] * m[0] + m[1] * m[1] - it * m[2] * m[2]) * id;
    const float hy2 = cy * cy - (m[3] * m[3] + m[4] * m[4] - it * m[5] * m[5]) * id;
    const float hx = sqrtf(fmaxf(hx2, 0.0f)), hy = sqrtf(fmaxf(hy2, 0.0f));
    bounded = (hx == hx) && (hy == hy) && (cx == cx) && (cy == cy);  // NaN guard
    x0 = fminf(x0, cx - hx); x1 = fmaxf(x1, cx + hx);
    y0 = fminf(y0, cy - hy); y1 = fmaxf(y1, cy + hy);
  }
  if (!bounded) return 0xFu;
  const float mg = 0.3f;
  x0 -= mg; x1 += mg; y0 -= mg; y1 += mg;
  unsigned mask = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float qx = tile_x0 + (float)((q & 1) * 8), qy = tile_y0 + (float)((q >> 1) * 8);
    // pixel centres of the quadrant span [qx+0.5, qx+7.5]
    if (x1 >= qx + 0.5f && x0 <= qx + 7.5f && y1 >= qy + 0.5f && y0 <= qy + 7.5f) mask |= 1u << q;
  }
  return mask;
}

// The same conservative box against the tile's sixteen 4x4-pixel sub-blocks (bit 4 q + s: quadrant q, sub-block s = 2 (y >> 2 & 1) +
// (x >> 2 & 1) within it) and against its sixteen 8x2 strips (bit 4 q + r, r = row pair within the quadrant): diagnostics of
// tools/exp_raster_pairs.py (how many iterations a wave would need if each of its four 16-lane rows followed its own list).
__device__ __forceinline__ void subblock_masks(const float *__restrict__ m, float mx, float my, float opac, float tile_x0, float tile_y0,
                                               unsigned &m4x4, unsigned &m8x2) {
  m4x4 = 0u; m8x2 = 0u;
  const float o255 = 255.0f * opac;
  if (!(o255 > 1.0f)) return;
  const float tau = 2.0f * __logf(o255) * 1.0001f + 1e-4f;
  const float r2 = sqrtf(0.5f * tau);
  float x0 = mx - r2, x1 = mx + r2, y0 = my - r2, y1 = my + r2;
  const float it = 1.0f / tau;
  const float d = m[6] * m[6] + m[7] * m[7] - it * m[8] * m[8];
  bool bounded = d < 0.0f;
  if (bounded) {
    const float id = 1.0f / d;
    const float cx = (m[0] * m[6] + m[1] * m[7] - it * m[2] * m[8]) * id;
    const float cy = (m[3] * m[6] + m[4] * m[7] - it * m[5] * m[8]) * id;
    const float hx2 = cx * cx - (m[0] * m[0] + m[1] * m[1] - it * m[2] * m[2]) * id;
    const float hy2 = cy * cy - (m[3] * m[3] + m[4] * m[4] - it * m[5] * m[5]) * id;
    const float hx = sqrtf(fmaxf(hx2, 0.0f)), hy = sqrtf(fmaxf(hy2, 0.0f));
    bounded = (hx == hx) && (hy == hy) && (cx == cx) && (cy == cy);
    x0 = fminf(x0, cx - hx); x1 = fmaxf(x1, cx + hx);
    y0 = fminf(y0, cy - hy); y1 = fmaxf(y1, cy + hy);
  }
  if (!bounded) { m4x4 = 0xFFFFu; m8x2 = 0xFFFFu; return; }
  const float mg = 0.3f;
  x0 -= mg; x1 += mg; y0 -= mg; y1 += mg;
  for (int q = 0; q < 4; ++q) {
    const float qx = tile_x0 + (float)((q & 1) * 8), qy = tile_y0 + (float)((q >> 1) * 8);
    for (int s = 0; s < 4; ++s) {
      const float sx = qx + (float)((s & 1) * 4), sy = qy + (float)((s >> 1) * 4);
      if (x1 >= sx + 0.5f && x0 <= sx + 3.5f && y1 >= sy + 0.5f && y0 <= sy + 3.5f) m4x4 |= 1u << (4 * q + s);
      const float ry = qy + (float)(2 * s);
      if (x1 >= qx + 0.5f && x0 <= qx + 7.5f && y1 >= ry + 0.5f && y0 <= ry + 1.5f) m8x2 |= 1u << (4 * q + s);
    }
  }
}

template <int CAP, bool BWD>
__device__ __forceinline__ void stage_splat(SplatBatchT<CAP, BWD> &s, int slot, int g, const float *__restrict__ means2d,
                                            const float *__restrict__ ray_transforms,
                                            const float *__restrict__ colors, const float *__restrict__ opacities,
                                            const float *__restrict__ normals, float tile_x0, float tile_y0) {
  const float *m = ray_transforms + 9 * (int64_t)g;
  const float2 xy = *reinterpret_cast<const float2 *>(means2d + 2 * (int64_t)g);
  const float *c = colors + 3 * (int64_t)g;
  const float *n = normals + 3 * (int64_t)g;
  const float opac = opacities[g];
  const float mu0 = m[0], mu1 = m[1], mu2 = m[2], mv0 = m[3], mv1 = m[4], mv2 = m[5], mw0 = m[6], mw1 = m[7], mw2 = m[8];
  // Explicit FMAs: h = p M_w - M is the cancelling step (|p M_w.z| ~ |M.z| ~ 1e3 |h.z|) and must be ONE rounding of the exact
  // value; left to the compiler, the SLP vectoriser turns two of the six into v_pk_mul_f32 + v_pk_add_f32 (seen in the ISA).
  // A = M_v x M_w, B = M_w x M_u
  const float ax = fmaf(mv1, mw2, -(mv2 * mw1)), ay = fmaf(mv2, mw0, -(mv0 * mw2)), az = fmaf(mv0, mw1, -(mv1 * mw0));
  const float bx = fmaf(mw1, mu2, -(mw2 * mu1)), by = fmaf(mw2, mu0, -(mw0 * mu2)), bz = fmaf(mw0, mu1, -(mw1 * mu0));
  // C0 = h_u x h_v at the splat's projected centre
  const float p0x = xy.x, p0y = xy.y;
  const float hux = fmaf(p0x, mw0, -mu0), huy = fmaf(p0x, mw1, -mu1), huz = fmaf(p0x, mw2, -mu2);
  const float hvx = fmaf(p0y, mw0, -mv0), hvy = fmaf(p0y, mw1, -mv1), hvz = fmaf(p0y, mw2, -mv2);
  const float cx = fmaf(huy, hvz, -(huz * hvy)), cy = fmaf(huz, hvx, -(hux * hvz)), cz = fmaf(hux, hvy, -(huy * hvx));
  const float D = fmaf(cz, mw2, fmaf(cx, mw0, cy * mw1));
  s.q0[slot] = make_float4(ax, ay, az, xy.x);
  s.q1[slot] = make_float4(bx, by, bz, xy.y);
  s.q2[slot] = make_float4(cx, cy, cz, opac);
  s.q3[slot] = make_float4(BWD ? mw0 : D, mw2, c[0], c[1]);
  if (BWD) s.extra[slot] = mw1;
  s.q4[slot] = make_float4(c[2], n[0], n[1], n[2]);
  s.m16[slot] = (unsigned short)subblock_mask4x4(m, xy.x, xy.y, opac, tile_x0, tile_y0);
}

// row sum / max over the 16 lanes of a DPP row; valid in lane 15 of each row
__device__ __forceinline__ float row_sum_to_lane15(float v) {
  v += dpp_mov<0x111>(v);
  v += dpp_mov<0x112>(v);
  v += dpp_mov<0x114>(v);
  v += dpp_mov<0x118>(v);
  return v;
}
__device__ __forceinline__ unsigned row_umax_to_lane15(unsigned v) {
  v = max(v, dpp_mov_u<0x111>(v));
  v = max(v, dpp_mov_u<0x112>(v));
  v = max(v, dpp_mov_u<0x114>(v));
  v = max(v, dpp_mov_u<0x118>(v));
  return v;
}
// pixel of a lane in the row-list kernels: 16-lane row s of wave q owns the 4x4 sub-block s of quadrant q
__device__ __forceinline__ void row_pixel(int wave, int lane, int &lx, int &ly) {
  const int sb = lane >> 4, j = lane & 15;
  lx = (wave & 1) * 8 + (sb & 1) * 4 + (j & 3);
  ly = (wave >> 1) * 8 + (sb >> 1) * 4 + (j >> 2);
}

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
