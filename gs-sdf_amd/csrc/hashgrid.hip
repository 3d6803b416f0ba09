// hashgrid.hip — S1 / S1' / S1'': multiresolution hash-grid encoding, forward, backward and
// double backward (the latter is what torch::autograd::grad(create_graph=true) needs for the
// reference's analytic eikonal term).
// Replaces tiny-cuda-nn's GridEncoding behind TCNNEncoding::forward
// (/root/reference/include/neural_net/encoding_map.cpp:15-26 config, :59 call;
//  second order requested at /root/reference/include/neural_net/local_map.cpp:151-172).
// Semantics: SPEC A.7 — Hash grid, 3-D, F=2, Linear interpolation, pos = fma(scale,x,0.5),
// dense index while the stride fits the level's table else prime hash, index % table size.
//
// MI355X mapping: wave64 = one point x 16 levels x 4 lanes, the 4 lanes of a (point, level) being
// (x-corner bit, feature):
//   * every memory instruction of a (y,z) corner pair touches 4 consecutive floats (16 B) whenever the two
//     x-neighbour entries are adjacent (always in the dense levels, for even x0 in the hashed levels);
//   * the 32 output features of a point are 128 contiguous bytes written by one wave (coalesced);
//   * the per-point reductions over levels (d/dx, double-backward d/dx) are DPP adds, no atomics, no LDS;
//   * the only scattered traffic is the corner gathers and the matching fp32 atomics of the table gradient,
//     the algorithm's compulsory random access; the 61 MB fp32 table is resident in the 256 MiB Infinity Cache.
#include "hashgrid_common.h"

#include <cstring>
#include <string>

namespace gsdf {

// sums `v` over the 16 lanes of a DPP row; the total is valid in the row's last lane (lane & 15 == 15)
__device__ __forceinline__ float row_sum_to_lane15(float v) {
  v += dpp_mov<0x111>(v);
  v += dpp_mov<0x112>(v);
  v += dpp_mov<0x114>(v);
  v += dpp_mov<0x118>(v);
  return v;
}

// forward: 4 lanes per (point, level) = (x-corner bit, feature), like the backward kernels below: the 4 lanes of a
// (y,z) corner pair read 4 consecutive floats {entry(x0).f0,.f1, entry(x1).f0,.f1} -> one 16-byte segment per group
// instead of two 8-byte gathers in different instructions; the two x partials are combined with one quad DPP.
// JAC: also stores d feat / d x (jac[b][level*2+f][0..2], 384 B per point) so that the first-order input gradient is
// a dense 32x3 contraction per point later (hashgrid_bwd_jac_kernel) instead of a second pass over the table.
template <bool JAC>
__global__ void __launch_bounds__(HG_THREADS)
    hashgrid_fwd_kernel(int64_t B, int64_t jac_rows, HgLevels lv, const float *__restrict__ x,
                        const float *__restrict__ table, float *__restrict__ feat, float *__restrict__ jac) {
  const int lane = threadIdx.x & 63;
  const int f = lane & 1, xb = (lane >> 1) & 1, level = lane >> 2;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // wave-uniform
  float acc = 0.f, jx = 0.f, jy = 0.f, jz = 0.f;
  if (level < lv.n_levels) {
    Cell c;
    load_cell(lv, level, x, b, table, c);
    const float *tb = table + (int64_t)lv.offset[level] * 2 + f;
    const float wx = xb ? c.fr[0] : 1.f - c.fr[0], sx = xb ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hy = k & 1, hz = k >> 1;
      const uint32_t idx = grid_index(c.hsize, c.res, c.g0[0] + xb, c.g0[1] + hy, c.g0[2] + hz);
      const float wy = hy ? c.fr[1] : 1.f - c.fr[1], wz = hz ? c.fr[2] : 1.f - c.fr[2];
      const float t = tb[2 * (int64_t)idx];
      acc += wx * wy * wz * t;
      if (JAC) {
        jx += sx * wy * wz * t;
        jy += (hy ? 1.f : -1.f) * wx * wz * t;
        jz += (hz ? 1.f : -1.f) * wx * wy * t;
      }
    }
    if (JAC) { jx *= c.scale; jy *= c.scale; jz *= c.scale; }
  }
  acc += dpp_mov<0x4E>(acc);  // quad_perm [2,3,0,1]: add the other x-corner's partial (same feature)
  if (JAC) { jx += dpp_mov<0x4E>(jx); jy += dpp_mov<0x4E>(jy); jz += dpp_mov<0x4E>(jz); }
  if (xb == 0 && level < lv.n_levels) {
    const int64_t o = (b * lv.n_levels + level) * 2 + f;
    feat[o] = acc;
    if (JAC && b < jac_rows) { jac[3 * o] = jx; jac[3 * o + 1] = jy; jac[3 * o + 2] = jz; }
  }
}

// v_x[b] = sum_k v_feat[b][k] * jac[b][k][:]  (k < n_out <= 32): half a wave per point, DPP row reduction
__global__ void __launch_bounds__(256)
    hashgrid_bwd_jac_kernel(int64_t B, int n_out, const float *__restrict__ jac, const float *__restrict__ v_feat,
                            float *__restrict__ v_x) {
  const int lane = threadIdx.x & 63, k = lane & 31;
  const int64_t b = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (b < B && k < n_out) {
    const float vf = v_feat[b * n_out + k];
    const float *j = jac + (b * n_out + k) * 3;
    gx = vf * j[0]; gy = vf * j[1]; gz = vf * j[2];
  }
  // sum over the 32 lanes of a half wave: 16-lane rows, then row_bcast15 into rows 1 and 3 -> lanes 31 and 63
  gx = row_sum_to_lane15(gx); gy = row_sum_to_lane15(gy); gz = row_sum_to_lane15(gz);
  gx += dpp_mov<0x142, 0xA, 0xF, false>(gx); gy += dpp_mov<0x142, 0xA, 0xF, false>(gy); gz += dpp_mov<0x142, 0xA, 0xF, false>(gz);
  if (k == 31 && b < B) { v_x[3 * b] = gx; v_x[3 * b + 1] = gy; v_x[3 * b + 2] = gz; }
}
// the same, scaled and ADDED to row ids[b] (b when ids == NULL) of `out`: the chain rule through the world -> unit-cube map and the
// row selection of the joint iteration's splat samples in the same launch (rows are distinct: plain read-modify-write)
__global__ void __launch_bounds__(256)
    hashgrid_bwd_jac_scatter_kernel(int64_t B, int n_out, const float *__restrict__ jac, const float *__restrict__ v_feat, float scale,
                                    const int64_t *__restrict__ ids, float *__restrict__ out) {
  const int lane = threadIdx.x & 63, k = lane & 31;
  const int64_t b = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (b < B && k < n_out) {
    const float vf = v_feat[b * n_out + k];
    const float *j = jac + (b * n_out + k) * 3;
    gx = vf * j[0]; gy = vf * j[1]; gz = vf * j[2];
  }
  gx = row_sum_to_lane15(gx); gy = row_sum_to_lane15(gy); gz = row_sum_to_lane15(gz);
  gx += dpp_mov<0x142, 0xA, 0xF, false>(gx); gy += dpp_mov<0x142, 0xA, 0xF, false>(gy); gz += dpp_mov<0x142, 0xA, 0xF, false>(gz);
  if (k == 31 && b < B) {
    float *o = out + 3 * (ids != nullptr ? ids[b] : b);
    o[0] += gx * scale; o[1] += gy * scale; o[2] += gz * scale;
  }
}

// XCD-partitioned forward for large batches.  The 14 hashed levels are 4 MiB each, an XCD's L2 is 4 MiB, and a wave of the
// kernel above touches all 16 levels: every L2 thrashes over the whole 58 MiB table (hit rate ~7 %; with a 2 MiB table
// the same kernel runs 2.5x faster).  Workgroups are dealt round-robin over the XCDs of their queue, so workgroup b is
// on XCD b % n_xcd: give each XCD a fixed, contiguous group of 2-3 levels for ALL points.  Its L2 then sees 8-12 MiB
// instead of 58.  A wave = (16 / nl) points x nl levels x 4 lanes; the feature row of a point is assembled by n_xcd
// workgroups writing 16-24 contiguous bytes each (merged in the memory-side cache).
struct XcdLevels {
  int begin[8], count[8];
};
template <bool JAC>
__global__ void __launch_bounds__(HG_THREADS)
    hashgrid_fwd_xcd_kernel(int64_t B, int64_t jac_rows, HgLevels lv, XcdLevels xl, int n_xcd,
                            const float *__restrict__ x, const float *__restrict__ table, float *__restrict__ feat,
                            float *__restrict__ jac) {
  const int xcd = blockIdx.x % n_xcd;
  const int64_t chunk = blockIdx.x / n_xcd;
  const int l0 = xl.begin[xcd], nl = xl.count[xcd];
  const int ppw = 16 / nl;  // points per wave
  const int lane = threadIdx.x & 63;
  const int f = lane & 1, xb = (lane >> 1) & 1, slot = lane >> 2;
  const int pw = slot / nl, level = l0 + slot - pw * nl;
  const int64_t b = (chunk * 4 + (threadIdx.x >> 6)) * ppw + pw;
  const bool live = pw < ppw && b < B;
  float acc = 0.f, jx = 0.f, jy = 0.f, jz = 0.f;
  if (live) {
    Cell c;
    load_cell(lv, level, x, b, table, c);
    const float *tb = table + (int64_t)lv.offset[level] * 2 + f;
    const float wx = xb ? c.fr[0] : 1.f - c.fr[0], sx = xb ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hy = k & 1, hz = k >> 1;
      const uint32_t idx = grid_index(c.hsize, c.res, c.g0[0] + xb, c.g0[1] + hy, c.g0[2] + hz);
      const float wy = hy ? c.fr[1] : 1.f - c.fr[1], wz = hz ? c.fr[2] : 1.f - c.fr[2];
      const float t = tb[2 * (int64_t)idx];
      acc += wx * wy * wz * t;
      if (JAC) {
        jx += sx * wy * wz * t;
        jy += (hy ? 1.f : -1.f) * wx * wz * t;
        jz += (hz ? 1.f : -1.f) * wx * wy * t;
      }
    }
    if (JAC) { jx *= c.scale; jy *= c.scale; jz *= c.scale; }
  }
  acc += dpp_mov<0x4E>(acc);
  if (JAC) { jx += dpp_mov<0x4E>(jx); jy += dpp_mov<0x4E>(jy); jz += dpp_mov<0x4E>(jz); }
  if (xb == 0 && live) {
    const int64_t o = (b * lv.n_levels + level) * 2 + f;
    feat[o] = acc;
    if (JAC && b < jac_rows) { jac[3 * o] = jx; jac[3 * o + 1] = jy; jac[3 * o + 2] = jz; }
  }
}

// levels -> XCD groups: contiguous, the cheap dense levels go to the groups that get one level more
static bool make_xcd_levels(int n_levels, int n_xcd, XcdLevels *xl, int *max_nl) {
  if (n_xcd < 2 || n_levels < n_xcd || (n_levels + n_xcd - 1) / n_xcd > 4) return false;
  const int base = n_levels / n_xcd, extra = n_levels % n_xcd;
  int l = 0;
  *max_nl = 0;
  for (int k = 0; k < 8; ++k) { xl->begin[k] = 0; xl->count[k] = 0; }
  for (int k = 0; k < n_xcd; ++k) {
    xl->begin[k] = l;
    xl->count[k] = base + (k < extra ? 1 : 0);
    l += xl->count[k];
    *max_nl = xl->count[k] > *max_nl ? xl->count[k] : *max_nl;
  }
  return true;
}

// Level slots of an XCD for the stencil kernel: slot j of XCD k = level lvl[k][j], walked for every chunk (mode 0) or for the chunks
// with chunk % m == r (mode = m << 4 | r).  A hashed level costs the same for every point, a coarse one much less (shared cells, dense
// table), and workgroups are dealt to the XCDs strictly in turn, so the XCD with the most expensive slots sets the pace of all:
// two XCDs can SHARE a level (one takes the even chunks, the other the odd ones) to even the load out.
struct XcdSlots {
  unsigned char lvl[8][16], mode[8][16];
  int count[8];
};

// Stencil batches (n base rows followed by 6 blocks of n central-difference rows, LocalMap.query_points layout): the
// 4 lanes of a (group, level) walk the group's 7 rows back to back instead of 7 workgroups far apart in time.
//   * at the coarse levels the +-delta points fall in the base point's cell: its 4 gathered values are reused from
//     registers (the loads of such a row are not issued at all);
//   * at the fine levels the 7 cells are neighbours and share most of their 64-byte segments: the gathers of rows 1..6 hit
//     the L1/L2 lines row 0 has just brought in, where the row-major kernel fetched them from the fabric 7 times.
// The arithmetic of a row is the one of hashgrid_fwd_xcd_kernel (same operations in the same order): bit-identical features.
// one chunk (4 ppw groups) of one level slot: the work of a (lane quad, chunk)
// IDX: the integer type of the row / feature offsets.  int32_t when 7 n rows x 16 levels x 2 features x 3 (the Jacobian's floats) stay below
// 2^31 (any batch of the reference's iteration): the 7 rows' addresses are then one 32-bit offset each against a uniform base instead of
// 64-bit pairs the compiler hoists out of the resident grid's chunk loop (round 5: 128 registers with 4 of them spilled).
// GEN (gsdf_hashgrid_fwd_stencil_points): the 7 rows of a group are MADE here from the group's world point instead of being read from a [7 n, 3]
// buffer that a launch in front of this one wrote (gsdf_sdf_query_points2): base row = rows [0, n_a) of `a`, then rows ids[j] (or j) of `b`, mapped to
// the unit cube; rows 1..6 = the point moved by +-delta along x, y, z IN WORLD UNITS and mapped again.  The operations are those of
// sdf_query_points2_kernel as hipcc compiles it (d = x - p; (d + d) * inv; fma(., 0.5, 0.5)), spelled out and kept from contracting: the same bits.
// The quad of level 0 writes the rows to x_out (the table scatter of the backward reads the base rows).  One launch and 18 of a lane's 21 position
// loads less; the level slots of a point sit on different XCDs, so every slot derives the rows again (9 map evaluations).
struct StencilGen {
  const float *a, *b;
  const int64_t *ids;
  int64_t n_a;
  float delta, px, py, pz, inv;
  float *x_out;
};
__device__ __forceinline__ float unit_coord(float x, float p, float inv) {
#pragma clang fp contract(off)
  const float d = x - p;
  const float m = (d + d) * inv;
  return __builtin_fmaf(m, 0.5f, 0.5f);
}
template <typename IDX>
__device__ __forceinline__ void stencil_rows(const StencilGen &sg, IDX n, IDX g, bool write, float (&xr)[7][3]) {
#pragma clang fp contract(off)
  const int64_t j = (int64_t)g;
  const float *src = j < sg.n_a ? sg.a + 3 * j : sg.b + 3 * (sg.ids != nullptr ? sg.ids[j - sg.n_a] : j - sg.n_a);
  const float w[3] = {src[0], src[1], src[2]}, p[3] = {sg.px, sg.py, sg.pz};
#pragma unroll
  for (int d = 0; d < 3; ++d) xr[0][d] = unit_coord(w[d], p[d], sg.inv);
#pragma unroll
  for (int r = 1; r < 7; ++r) {
    const int axis = (r - 1) >> 1;
    const float moved = w[axis] + (((r - 1) & 1) ? -sg.delta : sg.delta);
#pragma unroll
    for (int d = 0; d < 3; ++d) xr[r][d] = d == axis ? unit_coord(moved, p[d], sg.inv) : xr[0][d];
  }
  if (write) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      float *o = sg.x_out + 3 * ((int64_t)g + (int64_t)r * (int64_t)n);
      o[0] = xr[r][0]; o[1] = xr[r][1]; o[2] = xr[r][2];
    }
  }
}

template <bool JAC, typename IDX, bool GEN = false>
__device__ __forceinline__ void stencil_chunk(IDX n, const HgLevels &lv, int level, int f, int xb, IDX g, const float *__restrict__ x,
                                              const float *__restrict__ table, float *__restrict__ feat, float *__restrict__ jac,
                                              const StencilGen &sg) {
  const float scale = lv.scale[level];
  const uint32_t res = lv.res[level], hsize = lv.hsize[level];
  const float *tb = table + (int64_t)lv.offset[level] * 2 + f;
  uint32_t cg[7][3];
  float fr[7][3];
  float xr[7][3];
  if (GEN) stencil_rows<IDX>(sg, n, g, level == 0 && xb == 0 && f == 0, xr);
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const IDX b = g + (IDX)r * n;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float pos = fmaf(scale, GEN ? xr[r][d] : x[3 * b + d], 0.5f);
      const float fl = floorf(pos);
      cg[r][d] = (uint32_t)(int32_t)fl;
      fr[r][d] = pos - fl;
    }
  }
  float t[7][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[0][k] = tb[2 * grid_index(hsize, res, cg[0][0] + xb, cg[0][1] + (k & 1), cg[0][2] + (k >> 1))];
#pragma unroll
  for (int r = 1; r < 7; ++r) {
    const bool same = cg[r][0] == cg[0][0] && cg[r][1] == cg[0][1] && cg[r][2] == cg[0][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[r][k] = t[0][k];
    if (!same) {
#pragma unroll
      for (int k = 0; k < 4; ++k) t[r][k] = tb[2 * grid_index(hsize, res, cg[r][0] + xb, cg[r][1] + (k & 1), cg[r][2] + (k >> 1))];
    }
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const float wx = xb ? fr[r][0] : 1.f - fr[r][0], sx = xb ? 1.f : -1.f;
    float acc = 0.f, jx = 0.f, jy = 0.f, jz = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hy = k & 1, hz = k >> 1;
      const float wy = hy ? fr[r][1] : 1.f - fr[r][1], wz = hz ? fr[r][2] : 1.f - fr[r][2];
      acc += wx * wy * wz * t[r][k];
      if (JAC && r == 0) {
        jx += sx * wy * wz * t[r][k];
        jy += (hy ? 1.f : -1.f) * wx * wz * t[r][k];
        jz += (hz ? 1.f : -1.f) * wx * wy * t[r][k];
      }
    }
    acc += dpp_mov<0x4E>(acc);
    const IDX o = ((g + (IDX)r * n) * (IDX)lv.n_levels + (IDX)level) * 2 + (IDX)f;
    if (xb == 0) feat[o] = acc;
    if (JAC && r == 0) {
      jx *= scale; jy *= scale; jz *= scale;
      jx += dpp_mov<0x4E>(jx); jy += dpp_mov<0x4E>(jy); jz += dpp_mov<0x4E>(jz);
      if (xb == 0) { jac[3 * o] = jx; jac[3 * o + 1] = jy; jac[3 * o + 2] = jz; }
    }
  }
}

template <bool JAC, bool RESIDENT, typename IDX, bool GEN = false>
__global__ void __launch_bounds__(HG_THREADS, 4)   // <= 128 registers: three resident waves per SIMD leave room for a 96-register wave of another kernel
    hashgrid_fwd_stencil_kernel(int64_t n_, HgLevels lv, XcdSlots xl, int n_xcd, int ppw, int64_t chunks_, int64_t chunk_stride_,
                                const float *__restrict__ x, const float *__restrict__ table, float *__restrict__ feat, float *__restrict__ jac,
                                StencilGen sg) {
  const IDX n = (IDX)n_, chunks = (IDX)chunks_, chunk_stride = (IDX)chunk_stride_;
  const int xcd = blockIdx.x % n_xcd;
  const int nl = xl.count[xcd];
  const int lane = threadIdx.x & 63;
  const int f = lane & 1, xb = (lane >> 1) & 1, slot = lane >> 2;
  // ppw groups per wave (the same for every XCD: a chunk is the same 4 ppw groups everywhere), each against the XCD's nl <= 16 / ppw slots
  const int nls = 16 / ppw, pw = slot / nls, ls = slot - pw * nls;   // a group's level slots on neighbouring quads: its features leave in 8 nls byte runs
  if (!(pw < ppw && ls < nl)) return;  // whole quads leave together: the quad DPP of stencil_chunk stays among live lanes
  const int level = xl.lvl[xcd][ls], mode = xl.mode[xcd][ls];
  if (!RESIDENT) {   // one chunk per workgroup
    const IDX chunk = (IDX)(blockIdx.x / n_xcd);
    const IDX g = (chunk * 4 + (IDX)(threadIdx.x >> 6)) * ppw + pw;
    if (g >= n) return;
    if (mode != 0 && (int)(chunk % (mode >> 4)) != (mode & 15)) return;
    stencil_chunk<JAC, IDX, GEN>(n, lv, level, f, xb, g, x, table, feat, jac, sg);
  } else {           // a grid of chunk_stride workgroups per XCD that walks the chunks (see launch_fwd_stencil)
    for (IDX chunk = (IDX)(blockIdx.x / n_xcd); chunk < chunks; chunk += chunk_stride) {
      const IDX g = (chunk * 4 + (IDX)(threadIdx.x >> 6)) * ppw + pw;
      if (g >= n) break;
      if (mode != 0 && (int)(chunk % (mode >> 4)) != (mode & 15)) continue;
      stencil_chunk<JAC, IDX, GEN>(n, lv, level, f, xb, g, x, table, feat, jac, sg);
    }
  }
}

template <bool JAC>
static void launch_fwd(int64_t B, int64_t jac_rows, const HgLevels &lv, int n_levels, const float *x, const float *table,
                       float *feat, float *jac, hipStream_t stream) {
  XcdLevels xl;
  int max_nl = 0;
  const int n_xcd = xcd_count(stream);
  // measured: 17 % faster on a full-chip queue (2 levels = 8 MiB per XCD); no gain on a 6-XCD queue (3 levels = 12 MiB per
  // XCD, uneven groups) and none on 2 XCDs, so only full-chip queues take it
  // (measured and rejected: two launches with ONE level = 4 MiB per XCD and launch, to fit the 4 MiB L2: 1.58 against 1.66 ms
  // at 2.7 M points — the streaming x / feature traffic shares the L2 and the table still does not stay resident)
  if (n_xcd == 8 && B >= 65536 && make_xcd_levels(n_levels, n_xcd, &xl, &max_nl)) {
    const int ppw_min = 16 / max_nl;
    const int64_t chunks = (B + 4 * ppw_min - 1) / (4 * ppw_min);
    hashgrid_fwd_xcd_kernel<JAC><<<(unsigned)(chunks * n_xcd), HG_THREADS, 0, stream>>>(B, jac_rows, lv, xl, n_xcd, x, table, feat, jac);
  } else {
    hashgrid_fwd_kernel<JAC><<<(unsigned)((B + 3) / 4), HG_THREADS, 0, stream>>>(B, jac_rows, lv, x, table, feat, jac);
  }
}

// slots of the stencil kernel.  A map is written as XCDs separated by ';', slots by ',', a slot = level or level%m=r (only the chunks with
// chunk % m == r), e.g. "0,1;2,3;4,5%2=0;5%2=1,6;...".  16 levels on 8 XCDs take the measured map below, anything else contiguous groups.
static bool make_stencil_slots(int n_levels, int n_xcd, XcdSlots *xs, int *ppw) {
  *xs = XcdSlots{};
  int max_nl = 0;
  // 16 levels (base 32, x2: the reference's grid, config/base.yaml) on 8 XCDs, measured (tools/exp_hgmaps.sh, 494 k points): the four
  // coarse levels together cost two hashed ones (their +-delta rows share the base row's cell, tables of 0.3-4 MiB), level 4 three
  // quarters of one; the ten levels 6..15 are dealt in thirds, five thirds per XCD: 1.77 ms against 1.97 ms for contiguous pairs
  // (0.93 / 1.00 at 250 k, 2.81 / 3.27 at 800 k).  Quarters (3-4 tables per L2) and halves with 3 slots measured slower.
  static const std::string tuned16 = "0,1,2,3;4,5;6,7%3=0,7%3=1;7%3=2,8,9%3=0;9%3=1,9%3=2,10;11,12%3=0,12%3=1;12%3=2,13,14%3=0;14%3=1,14%3=2,15";
  static const std::string none;
  const std::string &env = (n_levels == 16 && n_xcd == 8) ? tuned16 : none;
  if (!env.empty()) {
    int k = 0;
    size_t i = 0;
    while (k < 8) {
      size_t j = env.find(';', i);
      std::string part = env.substr(i, j == std::string::npos ? std::string::npos : j - i);
      size_t a = 0;
      while (a < part.size()) {
        size_t c = part.find(',', a);
        std::string tok = part.substr(a, c == std::string::npos ? std::string::npos : c - a);
        if (!tok.empty() && xs->count[k] < 16) {
          const int lvl = atoi(tok.c_str());
          if (lvl < 0 || lvl >= n_levels) return false;
          xs->lvl[k][xs->count[k]] = (unsigned char)lvl;
          const size_t pc = tok.find('%');   // "level%m=r": only the chunks with chunk % m == r
          int m = 0, r = 0;
          if (pc != std::string::npos) { m = atoi(tok.c_str() + pc + 1); const size_t eq = tok.find('=', pc); r = eq == std::string::npos ? 0 : atoi(tok.c_str() + eq + 1); }
          if (m < 0 || m > 15 || r < 0 || r > 15 || (m > 0 && r >= m)) return false;
          xs->mode[k][xs->count[k]] = (unsigned char)(m > 1 ? (m << 4) | r : 0);
          xs->count[k]++;
        }
        if (c == std::string::npos) break;
        a = c + 1;
      }
      ++k;
      if (j == std::string::npos) break;
      i = j + 1;
    }
    for (int q = 0; q < 8; ++q) max_nl = xs->count[q] > max_nl ? xs->count[q] : max_nl;
    if (max_nl == 0) return false;
  } else {
    XcdLevels xl;
    if (!make_xcd_levels(n_levels, n_xcd, &xl, &max_nl)) return false;
    for (int k = 0; k < n_xcd; ++k) {
      xs->count[k] = xl.count[k];
      for (int j = 0; j < xl.count[k]; ++j) xs->lvl[k][j] = (unsigned char)(xl.begin[k] + j);
    }
  }
  *ppw = 16 / max_nl;
  return *ppw >= 1;
}

static thread_local int tl_stencil_resident = -1;   // <= 0: the full grid

template <bool JAC, bool GEN = false>
static void launch_fwd_stencil(int64_t n, const HgLevels &lv, int n_levels, const float *x, const float *table, float *feat,
                               float *jac, hipStream_t stream, const StencilGen &sg = StencilGen{}) {
  XcdSlots xs;
  int ppw = 1, n_xcd = xcd_count(stream);
  if (n_xcd != 8 || 7 * n < 65536 || !make_stencil_slots(n_levels, n_xcd, &xs, &ppw)) {
    xs = XcdSlots{};
    xs.count[0] = n_levels;
    for (int j = 0; j < n_levels; ++j) xs.lvl[0][j] = (unsigned char)j;
    n_xcd = 1;
    ppw = 1;
  }
  if (GEN) {   // the quad of level 0 writes the rows a point was made into: the map must give level 0 to one XCD for EVERY chunk (all built-in maps do)
    bool whole = false;
    for (int k = 0; k < n_xcd; ++k)
      for (int j = 0; j < xs.count[k]; ++j) whole = whole || (xs.lvl[k][j] == 0 && xs.mode[k][j] == 0);
    if (!whole) {
      xs = XcdSlots{};
      xs.count[0] = n_levels;
      for (int j = 0; j < n_levels; ++j) xs.lvl[0][j] = (unsigned char)j;
      n_xcd = 1;
      ppw = 1;
    }
  }
  const int64_t chunks = (n + 4 * ppw - 1) / (4 * ppw);
  // Resident grid (gsdf_hashgrid_fwd_stencil_resident(w) on this thread): w workgroups per CU (w * 32 per XCD)
  // that walk the chunks, instead of one workgroup per chunk.  The gathers are bound by the L1's miss queue, which two waves per SIMD keep
  // nearly as full as four; the wave slots and registers a full-occupancy grid would hold stay free for the kernels of another stream
  // (the resident kernel is held to 128 registers so that two of its waves leave room for two 128-register waves per SIMD).
  const int resident = tl_stencil_resident > 0 ? tl_stencil_resident : 0;   // gsdf_hashgrid_fwd_stencil_resident(): the caller's hint
  int64_t stride = chunks;
  if (resident > 0 && n_xcd == 8 && chunks > (int64_t)resident * 32) {
    stride = (int64_t)resident * 32;
    if (stride % 3 == 0) ++stride;   // slots dealt by chunk % 3: every workgroup sees all three residues in turn
  }
  const bool idx32 = 7 * n * (int64_t)n_levels * 6 < (int64_t)1 << 31;
#define STENCIL(RES, IDX, grid) hashgrid_fwd_stencil_kernel<JAC, RES, IDX, GEN><<<(unsigned)((grid) * n_xcd), HG_THREADS, 0, stream>>>(n, lv, xs, n_xcd, ppw, chunks, stride, x, table, feat, jac, sg)
  if (stride < chunks) { if (idx32) STENCIL(true, int32_t, stride); else STENCIL(true, int64_t, stride); }
  else { if (idx32) STENCIL(false, int32_t, chunks); else STENCIL(false, int64_t, chunks); }
#undef STENCIL
}

// ---- backward kernels: 4 lanes per (point, level) -------------------------------------------------------
// Measured on MI355X (tools/ubench/atomic_*.hip): fp32 atomics retire ~21 G 64-byte-LINE requests/s
// chip-wide regardless of footprint; lanes of one instruction that fall in the same line are merged.
// One lane per (point, level) issuing 16 scalar atomics costs 16 line requests.  Here the 4 lanes
// sub = (x-corner bit, feature) of a (point, level) issue ONE instruction per (y,z) corner pair whose
// addresses are the 4 consecutive floats {entry(x0).f0, .f1, entry(x1).f0, .f1}: entries x0/x1 are
// adjacent in the dense levels and, thanks to the hash's unit x-prime, for every even x0 in the hashed
// levels -> 4 (dense) to ~6 (hashed) line requests instead of 16.  wave64 = one point x 16 levels.
static constexpr int HGB_THREADS = 256;  // 4 points per workgroup

template <bool WANT_TABLE, bool WANT_X>
__global__ void __launch_bounds__(HGB_THREADS)
    hashgrid_bwd_kernel(int64_t B, HgLevels lv, const float *__restrict__ x, const float *__restrict__ table,
                        const float *__restrict__ v_feat, float *__restrict__ v_table, float *__restrict__ v_x) {
  const int lane = threadIdx.x & 63;
  const int f = lane & 1, xb = (lane >> 1) & 1, level = lane >> 2;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // wave-uniform
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (level < lv.n_levels) {
    Cell c;
    load_cell(lv, level, x, b, table, c);
    const float vf = v_feat[(b * lv.n_levels + level) * 2 + f];
    float *vt = v_table + (int64_t)lv.offset[level] * 2 + f;
    const float *tb = table + (int64_t)lv.offset[level] * 2 + f;
    const float wx = xb ? c.fr[0] : 1.f - c.fr[0], sx = xb ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hy = k & 1, hz = k >> 1;
      const uint32_t idx = grid_index(c.hsize, c.res, c.g0[0] + xb, c.g0[1] + hy, c.g0[2] + hz);
      const float wy = hy ? c.fr[1] : 1.f - c.fr[1], wz = hz ? c.fr[2] : 1.f - c.fr[2];
      if (WANT_TABLE) atomicAdd(vt + 2 * (int64_t)idx, wx * wy * wz * vf);
      if (WANT_X) {
        const float t = vf * tb[2 * (int64_t)idx];
        gx += sx * wy * wz * t;
        gy += (hy ? 1.f : -1.f) * wx * wz * t;
        gz += (hz ? 1.f : -1.f) * wx * wy * t;
      }
    }
    gx *= c.scale; gy *= c.scale; gz *= c.scale;
  }
  if (WANT_X) {
    gx = wave_sum_to_lane63(gx); gy = wave_sum_to_lane63(gy); gz = wave_sum_to_lane63(gz);
    if (lane == 63) { v_x[3 * b] = gx; v_x[3 * b + 1] = gy; v_x[3 * b + 2] = gz; }
  }
}

// double backward of v_x = J(x,table)^T v_feat: inputs vv_x (the gradient arriving at v_x)
template <bool WANT_VFEAT, bool WANT_TABLE, bool WANT_X>
__global__ void __launch_bounds__(HGB_THREADS)
    hashgrid_bwd_bwd_kernel(int64_t B, HgLevels lv, const float *__restrict__ x, const float *__restrict__ table,
                            const float *__restrict__ v_feat, const float *__restrict__ vv_x,
                            float *__restrict__ g_vfeat, float *__restrict__ g_table, float *__restrict__ g_x) {
  const int lane = threadIdx.x & 63;
  const int f = lane & 1, xb = (lane >> 1) & 1, level = lane >> 2;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // wave-uniform
  float ox = 0.f, oy = 0.f, oz = 0.f, gv = 0.f;
  if (level < lv.n_levels) {
    Cell c;
    load_cell(lv, level, x, b, table, c);
    const float vf = v_feat[(b * lv.n_levels + level) * 2 + f];
    const float vx = vv_x[3 * b], vy = vv_x[3 * b + 1], vz = vv_x[3 * b + 2];
    float *gt = g_table + (int64_t)lv.offset[level] * 2 + f;
    const float *tb = table + (int64_t)lv.offset[level] * 2 + f;
    const float wx = xb ? c.fr[0] : 1.f - c.fr[0], sx = xb ? 1.f : -1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hy = k & 1, hz = k >> 1;
      const uint32_t idx = grid_index(c.hsize, c.res, c.g0[0] + xb, c.g0[1] + hy, c.g0[2] + hz);
      const float wy = hy ? c.fr[1] : 1.f - c.fr[1], wz = hz ? c.fr[2] : 1.f - c.fr[2];
      const float sy = hy ? 1.f : -1.f, sz = hz ? 1.f : -1.f;
      const float t = c.scale * (vx * sx * wy * wz + vy * sy * wx * wz + vz * sz * wx * wy);
      const float th = tb[2 * (int64_t)idx];
      if (WANT_VFEAT) gv += t * th;
      if (WANT_TABLE) atomicAdd(gt + 2 * (int64_t)idx, t * vf);
      if (WANT_X) {
        const float q = vf * th;
        ox += (vy * sy * sx * wz + vz * sz * sx * wy) * q;
        oy += (vx * sx * sy * wz + vz * sz * sy * wx) * q;
        oz += (vx * sx * sz * wy + vy * sy * sz * wx) * q;
      }
    }
    const float s2 = c.scale * c.scale;
    ox *= s2; oy *= s2; oz *= s2;
  }
  if (WANT_VFEAT) {
    gv += dpp_mov<0x4E>(gv);  // quad_perm [2,3,0,1]: add the other x-corner's partial (same feature)
    if (xb == 0 && level < lv.n_levels) g_vfeat[(b * lv.n_levels + level) * 2 + f] = gv;
  }
  if (WANT_X) {
    ox = wave_sum_to_lane63(ox); oy = wave_sum_to_lane63(oy); oz = wave_sum_to_lane63(oz);
    if (lane == 63) { g_x[3 * b] = ox; g_x[3 * b + 1] = oy; g_x[3 * b + 2] = oz; }
  }
}

// third order: the backward of the double backward above, for lam_x (the gradient arriving at g_x) and mu (arriving at g_vfeat; NULL = zero).
// What a loss on the analytic Hessian needs (LocalMap::get_gradient(hessian = true, numerical_grad = 0) + curvate_loss,
// /root/reference/include/neural_net/local_map.cpp:151-168, include/neural_mapping/neural_mapping.cpp:117-121).  Trilinear weights: the second
// derivatives of a corner weight are the mixed ones, s_d s_e scale^2 w_r (r the remaining axis); the third is s_x s_y s_z scale^3.
// Off the hot path (curvate_weight is 0 in every shipped configuration): the 4-lanes-per-(point, level) form of the kernels above, atomics on the table.
template <bool WANT_TABLE>
__global__ void __launch_bounds__(HGB_THREADS)
    hashgrid_bwd3_kernel(int64_t B, HgLevels lv, const float *__restrict__ x, const float *__restrict__ table,
                         const float *__restrict__ v_feat, const float *__restrict__ vv_x, const float *__restrict__ lam_x,
                         const float *__restrict__ mu, float *__restrict__ t_vfeat, float *__restrict__ t_table,
                         float *__restrict__ t_vv, float *__restrict__ t_x) {
  const int lane = threadIdx.x & 63;
  const int f = lane & 1, xb = (lane >> 1) & 1, level = lane >> 2;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // wave-uniform
  float tvx = 0.f, tvy = 0.f, tvz = 0.f, txx = 0.f, txy = 0.f, txz = 0.f, tvf = 0.f;
  if (level < lv.n_levels) {
    Cell c;
    load_cell(lv, level, x, b, table, c);
    const float vf = v_feat[(b * lv.n_levels + level) * 2 + f];
    const float m = mu != nullptr ? mu[(b * lv.n_levels + level) * 2 + f] : 0.f;
    const float ax = vv_x[3 * b], ay = vv_x[3 * b + 1], az = vv_x[3 * b + 2];
    const float lx = lam_x[3 * b], ly = lam_x[3 * b + 1], lz = lam_x[3 * b + 2];
    float *tt = t_table + (int64_t)lv.offset[level] * 2 + f;
    const float *tb = table + (int64_t)lv.offset[level] * 2 + f;
    const float wx = xb ? c.fr[0] : 1.f - c.fr[0], sx = xb ? 1.f : -1.f;
    const float s1 = c.scale, s2 = c.scale * c.scale, s3 = s2 * c.scale;
    // symmetric pair sums of (vv, lam)
    const float pxy = ax * ly + ay * lx, pxz = ax * lz + az * lx, pyz = ay * lz + az * ly;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hy = k & 1, hz = k >> 1;
      const uint32_t idx = grid_index(c.hsize, c.res, c.g0[0] + xb, c.g0[1] + hy, c.g0[2] + hz);
      const float wy = hy ? c.fr[1] : 1.f - c.fr[1], wz = hz ? c.fr[2] : 1.f - c.fr[2];
      const float sy = hy ? 1.f : -1.f, sz = hz ? 1.f : -1.f;
      const float dx = sx * wy * wz, dy = sy * wx * wz, dz = sz * wx * wy;          // d w / d pos
      const float dxy = sx * sy * wz, dxz = sx * sz * wy, dyz = sy * sz * wx;      // mixed second derivatives
      const float A = s1 * (ax * dx + ay * dy + az * dz);
      const float Bk = s2 * (pxy * dxy + pxz * dxz + pyz * dyz);
      const float th = tb[2 * (int64_t)idx];
      if (WANT_TABLE) atomicAdd(tt + 2 * (int64_t)idx, vf * Bk + m * A);
      tvf += Bk * th;
      const float q = vf * th, r = m * th;
      tvx += q * s2 * (ly * dxy + lz * dxz) + r * s1 * dx;
      tvy += q * s2 * (lx * dxy + lz * dyz) + r * s1 * dy;
      tvz += q * s2 * (lx * dxz + ly * dyz) + r * s1 * dz;
      const float d3 = s3 * sx * sy * sz;
      txx += q * d3 * pyz + r * s2 * (ay * dxy + az * dxz);
      txy += q * d3 * pxz + r * s2 * (ax * dxy + az * dyz);
      txz += q * d3 * pxy + r * s2 * (ax * dxz + ay * dyz);
    }
  }
  if (t_vfeat != nullptr) {
    tvf += dpp_mov<0x4E>(tvf);  // quad_perm [2,3,0,1]: add the other x-corner's partial (same feature)
    if (xb == 0 && level < lv.n_levels) t_vfeat[(b * lv.n_levels + level) * 2 + f] = tvf;
  }
  if (t_vv != nullptr) {
    tvx = wave_sum_to_lane63(tvx); tvy = wave_sum_to_lane63(tvy); tvz = wave_sum_to_lane63(tvz);
    if (lane == 63) { t_vv[3 * b] = tvx; t_vv[3 * b + 1] = tvy; t_vv[3 * b + 2] = tvz; }
  }
  if (t_x != nullptr) {
    txx = wave_sum_to_lane63(txx); txy = wave_sum_to_lane63(txy); txz = wave_sum_to_lane63(txz);
    if (lane == 63) { t_x[3 * b] = txx; t_x[3 * b + 1] = txy; t_x[3 * b + 2] = txz; }
  }
}

}  // namespace gsdf

using namespace gsdf;

static int check_cfg(int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale, const char *who) {
  GSDF_REQUIRE(n_levels >= 1 && n_levels <= HG_MAX_LEVELS, "%s: n_levels %d not in [1,16]", who, n_levels);
  GSDF_REQUIRE(n_feat == 2, "%s: n_features_per_level %d unsupported (2 only, as the reference configures)", who, n_feat);
  GSDF_REQUIRE(log2_hashmap >= 3 && log2_hashmap <= 30 && base_res >= 1 && per_level_scale >= 1.0f,
               "%s: bad grid configuration", who);
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_fwd_stencil_resident(int wgs_per_cu) {
  const int before = tl_stencil_resident;
  tl_stencil_resident = wgs_per_cu < 0 ? -1 : (wgs_per_cu > 8 ? 8 : wgs_per_cu);
  return before;
}

extern "C" int64_t gsdf_hashgrid_offsets(int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                                         int64_t *offsets_host) {
  if (check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_offsets") != GSDF_OK) return -1;
  return build_levels(n_levels, log2_hashmap, base_res, per_level_scale, nullptr, offsets_host);
}

extern "C" int gsdf_hashgrid_fwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                 float per_level_scale, const float *x, const float *table, float *feat,
                                 gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_fwd");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_fwd");
  if (rc) return rc;
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(x && table && feat, "hashgrid_fwd: null buffer");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  launch_fwd<false>(B, 0, lv, n_levels, x, table, feat, nullptr, stream);
  GSDF_CHECK_LAUNCH("hashgrid_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_fwd_jac_rows(int64_t B, int64_t jac_rows, int n_levels, int n_feat, int log2_hashmap,
                                          int base_res, float per_level_scale, const float *x, const float *table,
                                          float *feat, float *jac, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_fwd_jac_rows");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_fwd_jac");
  if (rc) return rc;
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(jac_rows >= 0 && jac_rows <= B, "hashgrid_fwd_jac: jac_rows must be in [0, B]");
  GSDF_REQUIRE(x && table && feat && (jac || jac_rows == 0), "hashgrid_fwd_jac: null buffer");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  if (jac_rows > 0) launch_fwd<true>(B, jac_rows, lv, n_levels, x, table, feat, jac, stream);
  else launch_fwd<false>(B, 0, lv, n_levels, x, table, feat, nullptr, stream);
  GSDF_CHECK_LAUNCH("hashgrid_fwd_kernel<jac>");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_fwd_stencil(int64_t B, int64_t stencil_n, int64_t jac_rows, int n_levels, int n_feat,
                                         int log2_hashmap, int base_res, float per_level_scale, const float *x,
                                         const float *table, float *feat, float *jac, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_fwd_stencil");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_fwd_stencil");
  if (rc) return rc;
  GSDF_REQUIRE(stencil_n >= 0 && B == 7 * stencil_n, "hashgrid_fwd_stencil: a stencil batch has 7 * stencil_n rows");
  GSDF_REQUIRE(jac_rows == 0 || jac_rows == stencil_n, "hashgrid_fwd_stencil: jac_rows must be 0 or stencil_n (the base rows)");
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(x && table && feat && (jac || jac_rows == 0), "hashgrid_fwd_stencil: null buffer");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  if (jac_rows > 0) launch_fwd_stencil<true>(stencil_n, lv, n_levels, x, table, feat, jac, stream);
  else launch_fwd_stencil<false>(stencil_n, lv, n_levels, x, table, feat, nullptr, stream);
  GSDF_CHECK_LAUNCH("hashgrid_fwd_stencil_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_fwd_stencil_points(int64_t n_a, const float *xyz_a, int64_t n_b, const float *xyz_b, const int64_t *ids_b, float delta,
                                                const float *origin_host, float map_size_inv, int want_jac, int n_levels, int n_feat, int log2_hashmap,
                                                int base_res, float per_level_scale, const float *table, float *x_out, float *feat, float *jac,
                                                gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_fwd_stencil");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_fwd_stencil_points");
  if (rc) return rc;
  GSDF_REQUIRE(n_a >= 0 && n_b >= 0 && origin_host, "hashgrid_fwd_stencil_points: bad arguments");
  const int64_t n = n_a + n_b;
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE((n_a == 0 || xyz_a) && (n_b == 0 || xyz_b) && table && x_out && feat && (jac || !want_jac), "hashgrid_fwd_stencil_points: null buffer");
  GSDF_REQUIRE(delta > 0.f, "hashgrid_fwd_stencil_points: delta must be positive");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  const StencilGen sg{xyz_a, xyz_b, ids_b, n_a, delta, origin_host[0], origin_host[1], origin_host[2], map_size_inv, x_out};
  if (want_jac) launch_fwd_stencil<true, true>(n, lv, n_levels, nullptr, table, feat, jac, stream, sg);
  else launch_fwd_stencil<false, true>(n, lv, n_levels, nullptr, table, feat, nullptr, stream, sg);
  GSDF_CHECK_LAUNCH("hashgrid_fwd_stencil_kernel<points>");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_fwd_jac(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                     float per_level_scale, const float *x, const float *table, float *feat, float *jac,
                                     gsdf_stream_t stream) {
  return gsdf_hashgrid_fwd_jac_rows(B, B, n_levels, n_feat, log2_hashmap, base_res, per_level_scale, x, table, feat, jac, stream);
}

extern "C" int gsdf_hashgrid_bwd_jac(int64_t B, int n_levels, int n_feat, const float *jac, const float *v_feat,
                                     float *v_x, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd_jac");
  GSDF_REQUIRE(n_levels >= 1 && n_feat >= 1 && n_levels * n_feat <= 32, "hashgrid_bwd_jac: n_levels*n_feat must be <= 32");
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(jac && v_feat && v_x, "hashgrid_bwd_jac: null buffer");
  hashgrid_bwd_jac_kernel<<<(unsigned)((B + 7) / 8), 256, 0, stream>>>(B, n_levels * n_feat, jac, v_feat, v_x);
  GSDF_CHECK_LAUNCH("hashgrid_bwd_jac_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_bwd_jac_scatter(int64_t B, int n_levels, int n_feat, const float *jac, const float *v_feat, float scale,
                                             const int64_t *ids, float *out, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd_jac");
  GSDF_REQUIRE(B >= 0 && n_levels >= 1 && n_levels * n_feat <= 32, "hashgrid_bwd_jac_scatter: bad arguments");
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(jac && v_feat && out, "hashgrid_bwd_jac_scatter: null buffer");
  hashgrid_bwd_jac_scatter_kernel<<<(unsigned)((B + 7) / 8), 256, 0, stream>>>(B, n_levels * n_feat, jac, v_feat, scale, ids, out);
  GSDF_CHECK_LAUNCH("hashgrid_bwd_jac_scatter_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                 float per_level_scale, const float *x, const float *table, const float *v_feat,
                                 float *v_table, float *v_x, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_bwd");
  if (rc) return rc;
  if (B == 0 || (!v_table && !v_x)) return GSDF_OK;
  GSDF_REQUIRE(x && table && v_feat, "hashgrid_bwd: null buffer");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  const unsigned nb = (unsigned)((B + 3) / 4);
  if (v_table && v_x) hashgrid_bwd_kernel<true, true><<<nb, HG_THREADS, 0, stream>>>(B, lv, x, table, v_feat, v_table, v_x);
  else if (v_table)   hashgrid_bwd_kernel<true, false><<<nb, HG_THREADS, 0, stream>>>(B, lv, x, table, v_feat, v_table, v_x);
  else                hashgrid_bwd_kernel<false, true><<<nb, HG_THREADS, 0, stream>>>(B, lv, x, table, v_feat, v_table, v_x);
  GSDF_CHECK_LAUNCH("hashgrid_bwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_bwd_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                     float per_level_scale, const float *x, const float *table, const float *v_feat,
                                     const float *vv_x, float *g_vfeat, float *g_table, float *g_x,
                                     gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd_bwd");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_bwd_bwd");
  if (rc) return rc;
  if (B == 0 || (!g_vfeat && !g_table && !g_x)) return GSDF_OK;
  GSDF_REQUIRE(x && table && v_feat && vv_x, "hashgrid_bwd_bwd: null buffer");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  const unsigned nb = (unsigned)((B + 3) / 4);
#define L(A, Bq, Cq) hashgrid_bwd_bwd_kernel<A, Bq, Cq><<<nb, HG_THREADS, 0, stream>>>(B, lv, x, table, v_feat, vv_x, g_vfeat, g_table, g_x)
  const int sel = (g_vfeat ? 4 : 0) | (g_table ? 2 : 0) | (g_x ? 1 : 0);
  switch (sel) {
    case 1: L(false, false, true); break;  case 2: L(false, true, false); break; case 3: L(false, true, true); break;
    case 4: L(true, false, false); break;  case 5: L(true, false, true); break;  case 6: L(true, true, false); break;
    default: L(true, true, true);
  }
#undef L
  GSDF_CHECK_LAUNCH("hashgrid_bwd_bwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_hashgrid_bwd_bwd_bwd(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                                         const float *x, const float *table, const float *v_feat, const float *vv_x, const float *lam_x,
                                         const float *mu_vfeat, float *t_vfeat, float *t_table, float *t_vv, float *t_x, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd_bwd_bwd");
  int rc = check_cfg(n_levels, n_feat, log2_hashmap, base_res, per_level_scale, "hashgrid_bwd_bwd_bwd");
  if (rc) return rc;
  if (B == 0 || (!t_vfeat && !t_table && !t_vv && !t_x)) return GSDF_OK;
  GSDF_REQUIRE(x && table && v_feat && vv_x && lam_x, "hashgrid_bwd_bwd_bwd: null buffer");
  HgLevels lv;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, &lv, nullptr);
  const unsigned nb = (unsigned)((B + 3) / 4);
  if (t_table) hashgrid_bwd3_kernel<true><<<nb, HG_THREADS, 0, stream>>>(B, lv, x, table, v_feat, vv_x, lam_x, mu_vfeat, t_vfeat, t_table, t_vv, t_x);
  else hashgrid_bwd3_kernel<false><<<nb, HG_THREADS, 0, stream>>>(B, lv, x, table, v_feat, vv_x, lam_x, mu_vfeat, t_vfeat, t_table, t_vv, t_x);
  GSDF_CHECK_LAUNCH("hashgrid_bwd3_kernel");
  return GSDF_OK;
}
