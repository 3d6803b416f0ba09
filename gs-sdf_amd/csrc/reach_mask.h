// reach_mask.h — the reach mask of the compositing kernels (raster_pack.hip computes it, raster_fwd / raster_bwd take every skip decision from
// it), in a header of its own so that the SAME source also compiles for the host: tests/test_reach_mask_conservative.py builds it with g++
// (GSDF_REACH_MASK_HOST: plain libm in place of the device intrinsics) and checks on random splats that no pixel with alpha >= 1/255 lies in
// a block whose bit is clear.
#pragma once
#include <math.h>

#ifdef GSDF_REACH_MASK_HOST
#define GSDF_RM_FN static inline
#define GSDF_RM_LOGF logf
#define GSDF_RM_RSQRTF(x) (1.0f / sqrtf(x))
#define GSDF_RM_SQRTF_FAST(x) sqrtf(x)
#define GSDF_RM_RESTRICT
#else
#define GSDF_RM_FN __device__ __forceinline__
#define GSDF_RM_LOGF __logf
#define GSDF_RM_RSQRTF(x) rsqrtf(x)
#define GSDF_RM_SQRTF_FAST(x) __builtin_amdgcn_sqrtf(x)   /* v_sqrt_f32, 1 ulp: the intervals carry a 1e-3 px slack */
#define GSDF_RM_RESTRICT __restrict__
#endif

namespace gsdf {

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the reach mask at 2x2-PIXEL granularity (one bit per lane quad of the compositing kernels: 8 x 8 blocks per 16 x 16 tile, bit
// 8 by + bx).  A pair contributes to a pixel p only if min(g3, g2) <= tau = 2 ln(255 o), i.e. p lies in the ELLIPSE E = {g3 <= tau} (the screen
// projection of the splat disc u^2 + v^2 <= tau: centre c and shape S from the dual conic W diag(1, 1, -1/tau) W^T, the construction of the
// 3-sigma box of SPEC A.1; hyperbolic / degenerate conics keep every block) or in the low-pass DISK
// {g2 <= tau} of radius sqrt(tau / 2) about mean2d.  Instead of a box cut by a strip, every PIXEL ROW of the tile gets the exact x-interval of
// both convex regions, and a block's bit is the OR over its two pixel rows: no waste inside the box.  The work is split:
//   reach_params   once per visible splat (the pack pass of the compositing kernels): c, the row-interval coefficients of the ellipse inflated
//                  by the safety margin, the inflated disk radius;
//   reach_mask2x2  once per (tile, splat) pair (the mask pass): 16 pixel rows x two intervals.
// Safety margin: both regions are grown by RM_MARGIN px in every direction (the conic is formed about mean2d, where nothing cancels; the margin
// covers the kernels' own fp32 evaluation of the alpha test and the 1e-3 px slack of the interval rounding).  The ellipse is grown as a matrix: E (+) disk(m) lies inside the ellipse of S' = (1 + e) S + (1 + 1/e) m^2 I for any e > 0
// (Cauchy-Schwarz on the support functions); e = m / (minor semi-axis) makes S' tight across the splat, where the lanes are won.
// ---------------------------------------------------------------------------------------------------------------------------------------
#ifndef GSDF_RM_MARGIN
#define GSDF_RM_MARGIN 0.1f
#endif
static constexpr float RM_MARGIN = GSDF_RM_MARGIN;
// p[0] cx, p[1] cy, p[2] 1 / S'yy, p[3] S'xy / S'yy, p[4] kk = S'xx - S'xy^2 / S'yy (> 0), p[5] (disk radius + margin)^2, p[6] mean2d.x, p[7] mean2d.y;
// kk = -1: unbounded conic or a NaN on the way (every block reached), kk = -2: alpha < 1/255 everywhere (no block reached)
GSDF_RM_FN void reach_params(const float *GSDF_RM_RESTRICT m, float mx, float my, float opac, float *GSDF_RM_RESTRICT p) {
  p[0] = p[1] = p[2] = p[3] = p[5] = 0.f;
  p[4] = -2.f;
  p[6] = mx; p[7] = my;
  const float o255 = 255.0f * opac;
  if (!(o255 > 1.0f)) return;
  const float tau = 2.0f * GSDF_RM_LOGF(o255) * 1.0001f + 1e-4f;
  const float r2 = sqrtf(0.5f * tau) + RM_MARGIN;
  p[5] = r2 * r2;
  p[4] = -1.f;
  const float it = 1.0f / tau;
  const float d = m[6] * m[6] + m[7] * m[7] - it * m[8] * m[8];
  if (!(d < 0.0f)) return;
  // The conic in screen coordinates RELATIVE TO mean2d: rows M_u - mx M_w, M_v - my M_w (one rounding each, fmaf).  In absolute coordinates the
  // shape S = c c^T - (...) / d is the difference of two terms ~ |c|^2 ~ 4e6 at 1080p, i.e. +- 0.5 px^2 of fp32 rounding on a splat of 1 px^2;
  // about mean2d the centre offset is a fraction of the splat's own size and nothing cancels.
  const float u0 = fmaf(-mx, m[6], m[0]), u1 = fmaf(-mx, m[7], m[1]), u2 = fmaf(-mx, m[8], m[2]);
  const float v0 = fmaf(-my, m[6], m[3]), v1 = fmaf(-my, m[7], m[4]), v2 = fmaf(-my, m[8], m[5]);
  const float id = 1.0f / d;
  const float cxr = (u0 * m[6] + u1 * m[7] - it * u2 * m[8]) * id;
  const float cyr = (v0 * m[6] + v1 * m[7] - it * v2 * m[8]) * id;
  float sxx = cxr * cxr - (u0 * u0 + u1 * u1 - it * u2 * u2) * id;
  float syy = cyr * cyr - (v0 * v0 + v1 * v1 - it * v2 * v2) * id;
  const float sxy = cxr * cyr - (u0 * v0 + u1 * v1 - it * u2 * v2) * id;
  const float cx = mx + cxr, cy = my + cyr;
  sxx = fmaxf(sxx, 0.0f); syy = fmaxf(syy, 0.0f);
  const float tr = sxx + syy, df = sxx - syy;
  const float lmin = fmaxf(0.5f * (tr - sqrtf(df * df + 4.0f * sxy * sxy)), 0.0f);
  const float b = fmaxf(sqrtf(lmin), RM_MARGIN);                  // e <= 1
  const float e = RM_MARGIN / b;
  const float add = (1.0f + 1.0f / e) * RM_MARGIN * RM_MARGIN;
  const float Sxx = (1.0f + e) * sxx * 1.0001f + add, Syy = (1.0f + e) * syy * 1.0001f + add, Sxy = (1.0f + e) * sxy;
  const float isyy = 1.0f / Syy, slope = Sxy * isyy, kk = Sxx - Sxy * slope;
  if (!((cx == cx) && (cy == cy) && (isyy == isyy) && (slope == slope) && (kk == kk)) || !(kk > 0.0f) || !(fabsf(cx) < 1e7f) || !(fabsf(cy) < 1e7f) ||
      !(Sxx < 1e12f) || !(Syy < 1e12f))
    return;   // NaN / overflow: keep everything
  p[0] = cx; p[1] = cy; p[2] = isyy; p[3] = slope; p[4] = kk;
}

// bits [lo, hi] of an 8-bit block row for the x-interval [xa, xb] (tile-relative pixel-centre coordinates: pixel i has its centre at i + 0.5, block
// bx = pixels 2 bx, 2 bx + 1).  Pixel i is inside when xa <= i + 0.5 <= xb, i.e. ceil(xa - 0.5) <= i <= floor(xb - 0.5); both roundings are taken
// as ONE round-to-nearest of a value moved 1e-3 px outwards (rint(xa - 1e-3) <= ceil(xa - 0.5), rint(xb - 1 + 1e-3) >= floor(xb - 0.5), ties
// included): a pixel whose centre lies within 1e-3 px outside the interval may be kept, none inside it is dropped.  The caller passes
// xa - 1e-3 and xb - (1 - 1e-3); an empty interval is a NaN in either bound (every comparison with it fails).
GSDF_RM_FN unsigned rm_block_bits(float xa_m, float xb_m) {
  const float flo = rintf(xa_m), fhi = rintf(xb_m);
  const bool ok = (flo <= fhi) && (fhi >= 0.0f) && (flo <= 15.0f);
  const unsigned lo = (unsigned)fmaxf(flo, 0.0f) >> 1, hi = (unsigned)fminf(fmaxf(fhi, 0.0f), 15.0f) >> 1;
  return ok ? (2u << hi) - (1u << lo) : 0u;
}
// NaN-ignoring min / max (fminf / fmaxf return the other operand when one is a NaN: an empty pixel row does not widen the block row)
GSDF_RM_FN unsigned long long reach_mask2x2(const float *GSDF_RM_RESTRICT p, float tile_x0, float tile_y0) {
  if (!(p[4] > 0.0f)) return p[4] == -2.f ? 0ull : ~0ull;
  const float cy = p[1] - tile_y0, isyy = p[2], slope = p[3], kk = p[4], R2 = p[5];
  const float dmy = p[7] - tile_y0;
  const float cxa = (p[0] - tile_x0) - 1e-3f, cxb = (p[0] - tile_x0) - (1.0f - 1e-3f);
  const float dxa = (p[6] - tile_x0) - 1e-3f, dxb = (p[6] - tile_x0) - (1.0f - 1e-3f);
  unsigned lo32 = 0u, hi32 = 0u;
#pragma unroll
  for (int by = 0; by < 8; ++by) {
    // the block row's two pixel rows: x-interval of the ellipse (xm -+ w, w = sqrt(kk (1 - dy^2 / S'yy)); NaN when the row misses it) and of
    // the disk; the union of two rows of one convex shape is taken as [min, max] (a superset when the rows do not overlap)
    const float y0 = (float)(2 * by) + 0.5f, y1 = y0 + 1.0f;
    const float e0 = y0 - cy, e1 = y1 - cy;
    const float w0 = GSDF_RM_SQRTF_FAST(kk * (1.0f - e0 * e0 * isyy)), w1 = GSDF_RM_SQRTF_FAST(kk * (1.0f - e1 * e1 * isyy));
    const float ea = fminf((cxa + slope * e0) - w0, (cxa + slope * e1) - w1), eb = fmaxf((cxb + slope * e0) + w0, (cxb + slope * e1) + w1);
    const float d0 = y0 - dmy, d1 = y1 - dmy;
    const float v0 = GSDF_RM_SQRTF_FAST(R2 - d0 * d0), v1 = GSDF_RM_SQRTF_FAST(R2 - d1 * d1);
    const float da = dxa - fmaxf(v0, v1), db = dxb + fmaxf(v0, v1);
    const unsigned blk = rm_block_bits(ea, eb) | rm_block_bits(da, db);
    if (by < 4) lo32 |= blk << (8 * by);
    else hi32 |= blk << (8 * (by - 4));
  }
  return ((unsigned long long)hi32 << 32) | lo32;
}

}  // namespace gsdf
