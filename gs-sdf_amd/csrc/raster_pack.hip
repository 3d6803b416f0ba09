// raster_pack.hip — round 6: the two streaming passes in front of the compositing kernels (raster_quad.h).
//   raster_pack_kernel: one lane per visible splat -> one 128-byte record (affine form of the ray-splat cross product, blend payload, reach
//                       parameters).  Replaces the per-(tile, splat) recomputation of round 5's staging lanes.
//   raster_mask_kernel: one lane per (tile, splat) pair -> the 64-bit 2x2 reach mask (reach_mask.h).
// Reference operator: rasterize_to_pixels_2dgs (/root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223).
#include "raster_quad.h"

namespace gsdf {

__global__ void __launch_bounds__(256)
    raster_pack_kernel(int64_t M, const float *__restrict__ means2d, const float *__restrict__ ray_transforms,
                       const float *__restrict__ colors, const float *__restrict__ opacities, const float *__restrict__ normals,
                       float4 *__restrict__ rec) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= M) return;
  const float *m = ray_transforms + 9 * g;
  const float2 xy = *reinterpret_cast<const float2 *>(means2d + 2 * g);
  const float *c = colors + 3 * g;
  const float *n = normals + 3 * g;
  const float opac = opacities[g];
  const float mu0 = m[0], mu1 = m[1], mu2 = m[2], mv0 = m[3], mv1 = m[4], mv2 = m[5], mw0 = m[6], mw1 = m[7], mw2 = m[8];
  // Explicit FMAs (round 3): h = p M_w - M is the cancelling step and must be ONE rounding of the exact value.
  // A = M_v x M_w, B = M_w x M_u, C0 = h_u x h_v at the splat's projected centre, D = C0 . M_w
  const float ax = fmaf(mv1, mw2, -(mv2 * mw1)), ay = fmaf(mv2, mw0, -(mv0 * mw2)), az = fmaf(mv0, mw1, -(mv1 * mw0));
  const float bx = fmaf(mw1, mu2, -(mw2 * mu1)), by = fmaf(mw2, mu0, -(mw0 * mu2)), bz = fmaf(mw0, mu1, -(mw1 * mu0));
  const float p0x = xy.x, p0y = xy.y;
  const float hux = fmaf(p0x, mw0, -mu0), huy = fmaf(p0x, mw1, -mu1), huz = fmaf(p0x, mw2, -mu2);
  const float hvx = fmaf(p0y, mw0, -mv0), hvy = fmaf(p0y, mw1, -mv1), hvz = fmaf(p0y, mw2, -mv2);
  const float cx = fmaf(huy, hvz, -(huz * hvy)), cy = fmaf(huz, hvx, -(hux * hvz)), cz = fmaf(hux, hvy, -(huy * hvx));
  const float D = fmaf(cz, mw2, fmaf(cx, mw0, cy * mw1));
  float p[8];
  const float mm[9] = {mu0, mu1, mu2, mv0, mv1, mv2, mw0, mw1, mw2};
  reach_params(mm, xy.x, xy.y, opac, p);
  float4 *r = rec + 8 * g;
  r[0] = make_float4(ax, ay, az, xy.x);
  r[1] = make_float4(bx, by, bz, xy.y);
  r[2] = make_float4(cx, cy, cz, opac);
  r[3] = make_float4(D, mw2, c[0], c[1]);
  r[4] = make_float4(c[2], n[0], n[1], n[2]);
  r[5] = make_float4(mw0, mw1, 0.f, 0.f);
  r[6] = make_float4(p[0], p[1], p[2], p[3]);
  r[7] = make_float4(p[4], p[5], p[6], p[7]);
}

// one lane per pair; the pair's tile = the last tile whose first slot is <= the pair's slot (binary search in isect_offsets, 32 KB at 1080p)
__global__ void __launch_bounds__(256)
    raster_mask_kernel(int64_t I, int64_t total_tiles, int64_t n_tiles, int tw, const int32_t *__restrict__ isect_offsets,
                       const int32_t *__restrict__ flatten_ids, const float4 *__restrict__ rec, unsigned long long *__restrict__ masks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= I) return;
  int64_t lo = 0, hi = total_tiles;   // invariant: isect_offsets[lo] <= i, (hi == total_tiles or isect_offsets[hi] > i)
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)isect_offsets[mid] <= i) lo = mid; else hi = mid;
  }
  const int tl = (int)(lo % n_tiles);
  const int ty = tl / tw, tx = tl - ty * tw;
  const int g = flatten_ids[i];
  const float4 a = rec[8 * (int64_t)g + 6], b = rec[8 * (int64_t)g + 7];
  const float p[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  masks[i] = reach_mask2x2(p, (float)(tx * TILE), (float)(ty * TILE));
}

int raster_pack_launch(int64_t M, int64_t I, int64_t total_tiles, int64_t n_tiles, int tw, const float *means2d, const float *ray_transforms,
                       const float *colors, const float *opacities, const float *normals, const int32_t *isect_offsets,
                       const int32_t *flatten_ids, void *ws, hipStream_t stream) {
  if (M <= 0 || I <= 0) return GSDF_OK;
  float4 *rec = (float4 *)ws;
  unsigned long long *masks = const_cast<unsigned long long *>(ws_masks(ws, M));
  raster_pack_kernel<<<(unsigned)((M + 255) / 256), 256, 0, stream>>>(M, means2d, ray_transforms, colors, opacities, normals, rec);
  GSDF_CHECK_LAUNCH("raster_pack_kernel");
  raster_mask_kernel<<<(unsigned)((I + 255) / 256), 256, 0, stream>>>(I, total_tiles, n_tiles, tw, isect_offsets, flatten_ids, rec, masks);
  GSDF_CHECK_LAUNCH("raster_mask_kernel");
  return GSDF_OK;
}

}  // namespace gsdf
