// raster_quad.h — round 6: the compositing kernels on LANE-QUAD lists.
// Reference operator: rasterize_to_pixels_2dgs (call site /root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223, gradients
// consumed at :626-633); semantics SPEC A.4 / A.5.
//
// Round 5 gave every 16-lane DPP row of a wave (a 4x4-pixel sub-block) its own list of the staged splats; 28 % of the evaluated lanes blended,
// because a splat of this workload covers ~10 pixels and its box ~3 sub-blocks.  Here every lane QUAD (2x2 pixels) follows its own list: 16
// lists per wave, 64 per tile, built per batch with ballot + mbcnt from ONE 64-bit reach mask per (tile, splat) pair (reach_mask.h:
// reach_mask2x2, the exact pixel-row intervals of the alpha >= 1/255 region grown by a safety margin).  The work that does not depend on the
// pixel is hoisted out of the tile loop:
//   raster_pack_kernel   once per visible splat: the affine form of the ray-splat cross product (round 3), the blend payload and the reach
//                        parameters as ONE 128-byte record (one cache line; round 5 recomputed them per (tile, splat) pair from five arrays);
//   raster_mask_kernel   once per (tile, splat) pair, every lane busy: the 64-bit reach mask (round 5: ~250 instructions of every staging
//                        pass, with the last pass of a tile mostly idle);
//   forward / backward   stage = copy 80 (88) bytes + the mask; lists; blend.  Both take every skip decision from that mask, so the backward
//                        replays exactly the pairs the forward blended.
// Lane -> pixel: wave w = 8x8 quadrant, quad Q = lane >> 2 = one of its 4x4 blocks of 2x2 pixels (row-major), lane & 3 = pixel of the block.
// Lists are built in two steps per batch and wave: the staged splats that reach the wave's quadrant at all are compacted first (one ballot per
// 64 staged splats), then one ballot per quad runs over the compacted set (usually one chunk of <= 64) instead of over the whole batch.
#pragma once
#include <utility>

#include "raster_common.h"

namespace gsdf {

static constexpr int REC_F = 32;   // floats per packed splat record (128 B)
// record layout (floats): 0-3 q0 = (A.x, A.y, A.z, mean2d.x)   4-7 q1 = (B.x, B.y, B.z, mean2d.y)   8-11 q2 = (C0.x, C0.y, C0.z, opacity)
//                         12-15 q3 = (D, M_w.z, r, g)          16-19 q4 = (b, n.x, n.y, n.z)        20-23 (M_w.x, M_w.y, 0, 0)
//                         24-31 reach parameters (reach_mask.h: cx, cy, 1/S'yy, S'xy/S'yy, kk, R^2, mean2d.x, mean2d.y)

// quad Q = lane >> 2 of wave w owns the 2x2 block (bx, by) = (Q & 3, Q >> 2) of the wave's 8x8 quadrant; lane j = lane & 3 the pixel (j & 1, j >> 1) of it
__device__ __forceinline__ void quad_pixel(int wave, int lane, int &lx, int &ly) {
  const int Q = lane >> 2, j = lane & 3;
  lx = (wave & 1) * 8 + (Q & 3) * 2 + (j & 1);
  ly = (wave >> 1) * 8 + (Q >> 2) * 2 + (j >> 1);
}
// The wave's 16 bits of a 64-bit reach mask (bit 8 by + bx over the tile's 8x8 blocks) as a 16-bit mask with bit Q for quad Q:
// shift by wave_mask_base(w), then squeeze the four 4-bit groups at bits 0, 8, 16, 24 together.
__device__ __forceinline__ int wave_mask_base(int wave) { return 32 * (wave >> 1) + 4 * (wave & 1); }
__device__ __forceinline__ unsigned wave_quad_bits(unsigned long long m64, int mbase) {
  const unsigned m = (unsigned)(m64 >> mbase);
  return (m & 0xFu) | ((m >> 4) & 0xF0u) | ((m >> 8) & 0xF00u) | ((m >> 12) & 0xF000u);
}

// slot of the batch a thread stages / flushes: slots are dealt to the four waves in turn, so that a batch shorter than 256 still keeps every
// wave busy in the staging and flush phases (slot t = list position bstart + t; thread = 64 (t & 3) + (t >> 2))
__device__ __forceinline__ int thread_slot(int tid) { return 4 * (tid & 63) + (tid >> 6); }

// 16-bit mask of the wave's quads that still have a lane with the predicate set (bit Q <=> any of lanes 4Q .. 4Q+3)
__device__ __forceinline__ unsigned quads_any(unsigned long long b) {
  b |= b >> 1;
  b |= b >> 2;
  b &= 0x1111111111111111ull;
  // compress bit 4Q -> bit Q
  b = (b | (b >> 3)) & 0x0303030303030303ull;
  b = (b | (b >> 6)) & 0x000F000F000F000Full;
  b = (b | (b >> 12)) & 0x000000FF000000FFull;
  b = (b | (b >> 24)) & 0xFFFFull;
  return (unsigned)b;
}

// vec[LANE] = sval (a wave-uniform value); hipcc of ROCm 7.2 has no writelane builtin
template <int LANE>
__device__ __forceinline__ void writelane(int &vec, int sval) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(__builtin_amdgcn_readfirstlane(sval)), "n"(LANE));
}

// compile-time loop over the 16 quads of a wave: f(QuadC<Q>{}) for Q = 0 .. 15 (the lane of v_writelane_b32 must be an immediate)
template <int Q>
struct QuadC { static constexpr int value = Q; };
template <class F, int... Qs>
__device__ __forceinline__ void for_quads_impl(F &&f, std::integer_sequence<int, Qs...>) { (f(QuadC<Qs>{}), ...); }
template <class F>
__device__ __forceinline__ void for_quads(F &&f) { for_quads_impl(f, std::make_integer_sequence<int, 16>{}); }

template <int CTRL>
__device__ __forceinline__ int dpp_quad_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// OR / max over the 4 lanes of a quad, result in every lane (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E)
__device__ __forceinline__ unsigned quad_umax(unsigned v) {
  v = max(v, (unsigned)dpp_quad_i<0xB1>((int)v));
  v = max(v, (unsigned)dpp_quad_i<0x4E>((int)v));
  return v;
}
__device__ __forceinline__ int quad_imax(int v) {
  v = max(v, dpp_quad_i<0xB1>(v));
  v = max(v, dpp_quad_i<0x4E>(v));
  return v;
}
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  return v;
}
// Transposing butterfly over a lane quad: 16 per-lane values in, every lane keeps the QUAD sums of four of them, value indices
// f0 .. f0 + 3 with f0 = quad_reduce16_first(lane) (8 + 4 adds, 24 selects).
__device__ __forceinline__ void quad_transpose_reduce16(const float (&v)[16], int lane, float (&out)[4]) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (b0 ? v[k + 8] : v[k]) + dpp_mov<0xB1>(b0 ? v[k] : v[k + 8]);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = (b1 ? a[k + 4] : a[k]) + dpp_mov<0x4E>(b1 ? a[k] : a[k + 4]);
}
__device__ __forceinline__ int quad_reduce16_first(int lane) { return 8 * (lane & 1) + 4 * ((lane >> 1) & 1); }
// the same for 4 values: the lane keeps the quad sum of value index quad_reduce4_index(lane)
__device__ __forceinline__ float quad_transpose_reduce4(const float (&v)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float a[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) a[k] = (b0 ? v[k + 2] : v[k]) + dpp_mov<0xB1>(b0 ? v[k] : v[k + 2]);
  return (b1 ? a[1] : a[0]) + dpp_mov<0x4E>(b1 ? a[0] : a[1]);
}
__device__ __forceinline__ int quad_reduce4_index(int lane) { return 2 * (lane & 1) + ((lane >> 1) & 1); }

// workspace of the forward (the pack + mask passes write it, the backward may reuse it): records [M][32] floats, masks [I] u64
static inline size_t raster_pack_bytes(int64_t M, int64_t I) {
  return align_up((size_t)(M > 0 ? M : 1) * REC_F * sizeof(float), 256) + align_up((size_t)(I > 0 ? I : 1) * sizeof(unsigned long long), 256);
}
static inline const float *ws_records(const void *ws) { return (const float *)ws; }
static inline const unsigned long long *ws_masks(const void *ws, int64_t M) {
  return (const unsigned long long *)((const char *)ws + align_up((size_t)(M > 0 ? M : 1) * REC_F * sizeof(float), 256));
}
// pack + mask passes (raster_pack.hip)
int raster_pack_launch(int64_t M, int64_t I, int64_t total_tiles, int64_t n_tiles, int tw, const float *means2d, const float *ray_transforms,
                       const float *colors, const float *opacities, const float *normals, const int32_t *isect_offsets,
                       const int32_t *flatten_ids, void *ws, hipStream_t stream);

}  // namespace gsdf
