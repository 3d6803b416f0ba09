// reach_mask.h — the 4x4 reach mask of the compositing kernels (raster_common.h: stage_splat), in a header of its own so that the SAME source
// also compiles for the host: tests/test_reach_mask_conservative.py builds it with g++ (GSDF_REACH_MASK_HOST: plain libm in place of the device
// intrinsics) and checks on random splats that no pixel with alpha >= 1/255 lies in a sub-block whose bit is clear.
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

// Reach mask at 4x4-pixel granularity (bit 4 q + s: quadrant q = wave, sub-block s = 2 (y >> 2 & 1) + (x >> 2 & 1) = the 16-lane DPP row of
// the wave that owns those pixels): the same conservative box as quadrant_mask against the tile's sixteen sub-blocks.  Round 4: each
// 16-lane row of a wave follows ITS OWN list of the staged splats that reach its 4x4 pixels (row lists, compositing kernels), so a visit
// evaluates 16 pixels that the splat's box touches instead of the 64 of the whole quadrant.
GSDF_RM_FN unsigned subblock_mask4x4(const float *GSDF_RM_RESTRICT m, float mx, float my, float opac, float tile_x0, float tile_y0) {
  const float o255 = 255.0f * opac;
  if (!(o255 > 1.0f)) return 0u;
  const float tau = 2.0f * GSDF_RM_LOGF(o255) * 1.0001f + 1e-4f;
  const float r2 = sqrtf(0.5f * tau);
  float x0 = mx - r2, x1 = mx + r2, y0 = my - r2, y1 = my + r2;
  const float it = 1.0f / tau;
  const float d = m[6] * m[6] + m[7] * m[7] - it * m[8] * m[8];
  bool bounded = d < 0.0f;
  float cx = 0.f, cy = 0.f, hx2 = 0.f, hy2 = 0.f, sxy = 0.f;
  if (bounded) {
    const float id = 1.0f / d;
    cx = (m[0] * m[6] + m[1] * m[7] - it * m[2] * m[8]) * id;
    cy = (m[3] * m[6] + m[4] * m[7] - it * m[5] * m[8]) * id;
    hx2 = cx * cx - (m[0] * m[0] + m[1] * m[1] - it * m[2] * m[2]) * id;
    hy2 = cy * cy - (m[3] * m[3] + m[4] * m[4] - it * m[5] * m[5]) * id;
    sxy = cx * cy - (m[0] * m[3] + m[1] * m[4] - it * m[2] * m[5]) * id;
    const float hx = sqrtf(fmaxf(hx2, 0.0f)), hy = sqrtf(fmaxf(hy2, 0.0f));
    bounded = (hx == hx) && (hy == hy) && (cx == cx) && (cy == cy);
    x0 = fminf(x0, cx - hx); x1 = fmaxf(x1, cx + hx);
    y0 = fminf(y0, cy - hy); y1 = fmaxf(y1, cy + hy);
  }
  if (!bounded) return 0xFFFFu;
  const float mg = 0.3f;
  x0 -= mg; x1 += mg; y0 -= mg; y1 += mg;
  // columns / rows of 4 pixels: column c spans pixel centres [tile_x0 + 4c + 0.5, tile_x0 + 4c + 3.5]
  unsigned cols = 0u, rows = 0u;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float sx = tile_x0 + (float)(4 * c), sy = tile_y0 + (float)(4 * c);
    if (x1 >= sx + 0.5f && x0 <= sx + 3.5f) cols |= 1u << c;
    if (y1 >= sy + 0.5f && y0 <= sy + 3.5f) rows |= 1u << c;
  }
  // Round 5: the box of an elongated, oblique splat covers sub-blocks its region never reaches.  The region is {g3 <= tau} U {g2 <= tau} = the
  // ellipse (p - c)^T S^-1 (p - c) <= 1 of the same dual conic, S = [[hx2, sxy], [sxy, hy2]], united with the low-pass disk of radius r2 about
  // mean2d; both lie in the strip lo <= n . (p - c) <= hi across the ellipse's MINOR axis n (half-width sqrt(lambda_min) for the ellipse, the
  // disk's own interval), so a sub-block whose pixel centres project outside the strip is dropped: 4 flops per sub-block once n is known.
  // (Measured and not adopted: the exact ellipse / rectangle distance per sub-block, ~40 flops each: 32 % useful lanes instead of 26 %, but the
  // staging lane's ~600 flops per pair cost more than the visits they saved; a second strip across the major axis: no visit fewer.)
  float nx = 0.f, ny = 0.f, lo = 0.f, hi = 0.f;
  bool strip = false;
  if ((cols & (cols - 1u)) != 0u && (rows & (rows - 1u)) != 0u) {   // at least 2 x 2 sub-blocks in the box
    const float tr = hx2 + hy2, df = hx2 - hy2;
    const float disc = sqrtf(df * df + 4.0f * sxy * sxy);
    if (disc > 0.2f * tr) {   // clearly elongated: the minor axis is well conditioned
      const float l0 = 0.5f * (tr - disc);
      float ax = sxy, ay = l0 - hx2;
      const float bx = l0 - hy2, by = sxy;
      if (bx * bx + by * by > ax * ax + ay * ay) { ax = bx; ay = by; }
      const float nn = ax * ax + ay * ay;
      if (nn > 0.0f) {
        const float inv = GSDF_RM_RSQRTF(nn);
        nx = ax * inv; ny = ay * inv;
        const float sw = sqrtf(fmaxf(l0, 0.0f) + 1e-5f * tr) * 1.01f;
        const float dm = nx * (mx - cx) + ny * (my - cy);
        lo = fminf(-sw, dm - r2); hi = fmaxf(sw, dm + r2);
        strip = (nx == nx) && (ny == ny) && (lo == lo) && (hi == hi);
      }
    }
  }
  const float rb = (1.5f + 0.5f) * (fabsf(nx) + fabsf(ny));             // half extent of a sub-block's pixel centres along n, + 0.5 px
  const float d00 = nx * (tile_x0 + 2.0f - cx) + ny * (tile_y0 + 2.0f - cy);   // centre of sub-block (0, 0)
  unsigned mask = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      const int c = 2 * (q & 1) + (sb & 1), r = 2 * (q >> 1) + (sb >> 1);
      bool hit = ((cols >> c) & 1u) && ((rows >> r) & 1u);
      if (strip) {
        const float dd = d00 + 4.0f * ((float)c * nx + (float)r * ny);
        hit = hit && (dd - rb <= hi) && (dd + rb >= lo);
      }
      if (hit) mask |= 1u << (4 * q + sb);
    }
  return mask;
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the reach mask at 2x2-PIXEL granularity (one bit per lane quad of the compositing kernels: 8 x 8 blocks per 16 x 16 tile, bit
// 8 by + bx).  A pair contributes to a pixel p only if min(g3, g2) <= tau = 2 ln(255 o), i.e. p lies in the ELLIPSE E = {g3 <= tau} (the screen
// projection of the splat disc u^2 + v^2 <= tau: centre c and shape S from the dual conic, as in the 4x4 mask above) or in the low-pass DISK
// {g2 <= tau} of radius sqrt(tau / 2) about mean2d.  Instead of a box cut by a strip, every PIXEL ROW of the tile gets the exact x-interval of
// both convex regions, and a block's bit is the OR over its two pixel rows: no waste inside the box.  The work is split:
//   reach_params   once per visible splat (the pack pass of the compositing kernels): c, the row-interval coefficients of the ellipse inflated
//                  by the safety margin, the inflated disk radius;
//   reach_mask2x2  once per (tile, splat) pair (the mask pass): 16 pixel rows x two intervals.
// Safety margin: both regions are grown by RM_MARGIN px in every direction (fp32 cancellation in c and S stays below 0.1 px for |c| < 4096, as
// for the 4x4 mask).  The ellipse is grown as a matrix: E (+) disk(m) lies inside the ellipse of S' = (1 + e) S + (1 + 1/e) m^2 I for any e > 0
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
