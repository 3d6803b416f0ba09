// reach_mask.h — the 4x4 reach mask of the compositing kernels (raster_common.h: stage_splat), in a header of its own so that the SAME source
// also compiles for the host: tests/test_reach_mask_conservative.py builds it with g++ (GSDF_REACH_MASK_HOST: plain libm in place of the device
// intrinsics) and checks on random splats that no pixel with alpha >= 1/255 lies in a sub-block whose bit is clear.
#pragma once
#include <math.h>

#ifdef GSDF_REACH_MASK_HOST
#define GSDF_RM_FN static inline
#define GSDF_RM_LOGF logf
#define GSDF_RM_RSQRTF(x) (1.0f / sqrtf(x))
#define GSDF_RM_RESTRICT
#else
#define GSDF_RM_FN __device__ __forceinline__
#define GSDF_RM_LOGF __logf
#define GSDF_RM_RSQRTF(x) rsqrtf(x)
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

}  // namespace gsdf
