// occupancy.hip — A1: occupancy acceleration structure (query + voxel ray march) for the SDF ray sampler.
// Replaces the OctreeAS of the reference's absent kaolin_wisp_cpp submodule as its call sites use it:
//   SubMap::update_octree_as  /root/reference/include/neural_net/sub_map.cpp:22-35   (quantize, 27-neighbour dilation, build)
//   SubMap::get_valid_mask    sub_map.cpp:76-80;  LocalMap::filter_sample local_map.cpp:511-516   (query(...).pidx > -1)
//   LocalMap::sample          local_map.cpp:467-476                                               (raymarch "voxel")
//   get_quantized_points      /root/reference/include/neural_mapping/neural_mapping.cpp:755-758
// MI355X design: no pointer octree.  A dense BIT PYRAMID in HBM — level l holds 2^(3l) bits, x-fastest, packed in
// uint32 words, levels 0..L back to back (L = 9: 19 MB, L = 11: 1.2 GB of the 288 GB) — so that a query is one bit
// test and a ray skips empty space by testing levels L-6 and L-3 before L.  Semantics: DESIGN.md SPEC A.9; the CPU
// restatement oracle/occ_oracle.c mirrors the arithmetic op for op (this file is compiled with -ffp-contract=off).
#include "common.h"

namespace gsdf {

__host__ __device__ static inline int64_t occ_level_words(int l) {
  const int64_t bits = (int64_t)1 << (3 * l);
  return bits < 32 ? 1 : bits / 32;
}
__host__ __device__ static inline int64_t occ_level_offset(int l) {
  int64_t o = 0;
  for (int k = 0; k < l; ++k) o += occ_level_words(k);
  return o;
}

struct OccLevels {  // word offset of each pyramid level (by value to the kernels)
  int64_t off[22];
};
static OccLevels make_levels(int L) {
  OccLevels lv;
  for (int l = 0; l <= L + 1 && l < 22; ++l) lv.off[l] = occ_level_offset(l);
  return lv;
}

__device__ __forceinline__ int64_t cell_index(int l, int x, int y, int z) { return ((((int64_t)z << l) + y) << l) + x; }
__device__ __forceinline__ unsigned occ_bit(const uint32_t *grid, const OccLevels &lv, int l, int x, int y, int z) {
  const int64_t idx = cell_index(l, x, y, z);
  return (grid[lv.off[l] + (idx >> 5)] >> (idx & 31)) & 1u;
}
__device__ __forceinline__ int occ_quantize(float x, int res) {
  float q = floorf((float)res * (x + 1.0f) / 2.0f);
  q = q < 0.f ? 0.f : q;
  q = q > (float)(res - 1) ? (float)(res - 1) : q;
  return (int)q;
}

__global__ void __launch_bounds__(256)
    occ_set_kernel(int L, OccLevels lv, int64_t n, int K, const float *__restrict__ xyz, uint32_t *__restrict__ grid) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * K) return;
  const int64_t i = t / K;
  const int k = (int)(t - i * K);
  const int res = 1 << L;
  int x = occ_quantize(xyz[3 * i], res), y = occ_quantize(xyz[3 * i + 1], res), z = occ_quantize(xyz[3 * i + 2], res);
  if (K == 27) {
    x = min(max(x + k % 3 - 1, 0), res - 1);
    y = min(max(y + (k / 3) % 3 - 1, 0), res - 1);
    z = min(max(z + k / 9 - 1, 0), res - 1);
  }
  const int64_t idx = cell_index(L, x, y, z);
  atomicOr(grid + lv.off[L] + (idx >> 5), 1u << (idx & 31));
}

// parent level l from child level l+1.  One thread per parent WORD when the parent row is >= 32 cells wide
// (8 child words -> 1), else one thread per parent cell.
__device__ __forceinline__ uint32_t compact_pairs(uint32_t w) {  // bit i of the result = w[2i] | w[2i+1], i < 16
  w = (w | (w >> 1)) & 0x55555555u;
  w = (w | (w >> 1)) & 0x33333333u;
  w = (w | (w >> 2)) & 0x0F0F0F0Fu;
  w = (w | (w >> 4)) & 0x00FF00FFu;
  w = (w | (w >> 8)) & 0x0000FFFFu;
  return w;
}
__global__ void __launch_bounds__(256) occ_reduce_kernel(int l, OccLevels lv, uint32_t *__restrict__ grid) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int r = 1 << l;
  const uint32_t *child = grid + lv.off[l + 1];
  uint32_t *parent = grid + lv.off[l];
  if (r >= 32) {
    const int wpr = r / 32;  // parent words per row
    if (t >= (int64_t)wpr * r * r) return;
    const int wx = (int)(t % wpr);
    const int64_t yz = t / wpr;
    const int y = (int)(yz % r), z = (int)(yz / r);
    const int cw = 2 * r / 32;  // child words per row
    uint32_t lo = 0, hi = 0;
    for (int c = 0; c < 4; ++c) {
      const int64_t row = ((int64_t)(2 * z + (c >> 1)) * (2 * r) + (2 * y + (c & 1))) * cw + 2 * wx;
      lo |= child[row];
      hi |= child[row + 1];
    }
    parent[t] = compact_pairs(lo) | (compact_pairs(hi) << 16);
  } else {
    if (t >= (int64_t)r * r * r) return;
    const int x = (int)(t % r), y = (int)((t / r) % r), z = (int)(t / ((int64_t)r * r));
    unsigned any = 0;
    for (int c = 0; c < 8; ++c) any |= occ_bit(grid, lv, l + 1, 2 * x + (c & 1), 2 * y + ((c >> 1) & 1), 2 * z + (c >> 2));
    if (any) atomicOr(parent + (t >> 5), 1u << (t & 31));
  }
}

// AFFINE: xyz are world coordinates, mapped to the cube frame as SubMap::xyz_to_m1p1_pts does, ((x - origin) * 2) * inv
template <bool AFFINE>
__global__ void __launch_bounds__(256)
    occ_query_kernel(int l, OccLevels lv, int64_t n, const float *__restrict__ xyz, float ox, float oy, float oz, float inv,
                     const uint32_t *__restrict__ grid, uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int res = 1 << l;
  float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  if (AFFINE) { x = ((x - ox) * 2.0f) * inv; y = ((y - oy) * 2.0f) * inv; z = ((z - oz) * 2.0f) * inv; }
  const bool in = x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f && z >= -1.0f && z <= 1.0f;
  mask[i] = (uint8_t)(in && occ_bit(grid, lv, l, occ_quantize(x, res), occ_quantize(y, res), occ_quantize(z, res)));
}

// ---- the visible, occupancy-valid set of the joint iteration (neural_mapping.cpp:423-437) ---------------------------------------
// flag(i) = visibilities[i] > thr  &&  sample i inside an occupied level-l voxel;  w_all[i] = samples_weights[i] * visibilities[i];
// ids = the flagged rows in increasing order (what nonzero() returns).  Three launches: flags + weights + per-workgroup counts,
// one-workgroup scan of the counts (+ total), ordered write with ballot ranks.  Replaces mul, compare, and, nonzero (a hipcub
// reduction, a read-back, a rocPRIM partition) = 10 libtorch launches on the chain between the compositing forward and the SDF leg.
static constexpr int VS_ROWS = 1024;   // rows per workgroup (256 lanes x 4)
__global__ void __launch_bounds__(256)
    visible_flags_kernel(int l, OccLevels lv, int64_t n, const float *__restrict__ xyz, float ox, float oy, float oz, float inv,
                         const uint32_t *__restrict__ grid, const float *__restrict__ vis, const float *__restrict__ sw, float thr,
                         float *__restrict__ w_all, uint8_t *__restrict__ flags, int32_t *__restrict__ counts) {
  __shared__ int s_cnt[4];
  const int res = 1 << l;
  int c = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t i = (int64_t)blockIdx.x * VS_ROWS + q * 256 + threadIdx.x;
    if (i >= n) continue;
    const float v = vis[i];
    w_all[i] = sw[i] * v;
    const float x = ((xyz[3 * i] - ox) * 2.0f) * inv, y = ((xyz[3 * i + 1] - oy) * 2.0f) * inv, z = ((xyz[3 * i + 2] - oz) * 2.0f) * inv;
    const bool in = x >= -1.0f && x <= 1.0f && y >= -1.0f && y <= 1.0f && z >= -1.0f && z <= 1.0f;
    const bool f = v > thr && in && occ_bit(grid, lv, l, occ_quantize(x, res), occ_quantize(y, res), occ_quantize(z, res));
    flags[i] = (uint8_t)f;
    c += f ? 1 : 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}
// exclusive scan of nblk counts in place, total -> *total (one workgroup of 1024 lanes, any nblk)
__global__ void __launch_bounds__(1024) visible_scan_kernel(int nblk, int32_t *__restrict__ counts, int64_t *__restrict__ total) {
  __shared__ int s_w[16];
  __shared__ int s_carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wave; ++w) before += s_w[w];
    if (i < nblk) counts[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) store_host_visible(total, s_carry);   // (the count may be a host-visible word: gsdf_host_words_alloc)
}
__global__ void __launch_bounds__(256)
    visible_write_kernel(int64_t n, const uint8_t *__restrict__ flags, const int32_t *__restrict__ offsets, int64_t *__restrict__ ids) {
  __shared__ int s_cnt[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool f[4];
  int rank[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {   // rows in increasing order: pass q covers rows [256 q, 256 q + 256) of the block, wave w its 64-row slice
    const int64_t i = (int64_t)blockIdx.x * VS_ROWS + q * 256 + threadIdx.x;
    f[q] = i < n && flags[i];
    const unsigned long long b = __ballot(f[q]);
    rank[q] = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[q][wave] = __popcll(b);
  }
  __syncthreads();
  int base = offsets[blockIdx.x];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) before += w < wave ? s_cnt[q][w] : 0;
    if (f[q]) ids[base + before + rank[q]] = (int64_t)blockIdx.x * VS_ROWS + q * 256 + threadIdx.x;
    base += s_cnt[q][0] + s_cnt[q][1] + s_cnt[q][2] + s_cnt[q][3];
  }
}

// occupied level-L voxels: popcount per word, then (with the exclusive scan of the counts) their coordinates
__global__ void __launch_bounds__(256)
    occ_popc_kernel(int64_t n_words, const uint32_t *__restrict__ words, int32_t *__restrict__ counts) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w < n_words) counts[w] = __popc(words[w]);
}
__global__ void __launch_bounds__(256)
    occ_list_kernel(int L, int64_t n_words, const uint32_t *__restrict__ words, const int64_t *__restrict__ offsets,
                    int16_t *__restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  uint32_t bits = words[w];
  int64_t j = offsets[w];
  const int64_t mask = ((int64_t)1 << L) - 1;
  while (bits) {
    const int b = __builtin_ctz(bits);
    bits &= bits - 1;
    const int64_t idx = w * 32 + b;
    out[3 * j] = (int16_t)(idx & mask);
    out[3 * j + 1] = (int16_t)((idx >> L) & mask);
    out[3 * j + 2] = (int16_t)(idx >> (2 * L));
    ++j;
  }
}

// ---- voxel ray march: one WAVE per ray, one major-axis SLAB per lane (DESIGN.md SPEC A.9, round 5) ---------------------------------
// Grid frame g(s) = go + u s, |u| = 1, s in [s0, s1] = the ray inside the cube.  m = the axis with the largest |u|.  Slab k (cells with
// c[m] = k) is crossed for s in [sa, sb]; inside it the ray advances at most one cell along either minor axis, so a slab holds at most
// three cells, separated by at most one integer crossing per minor axis.  Slabs do not depend on each other: lane i of the wave takes slab
// base + i, tests its <= 3 cells against the finest level's bits (three independent loads) and the wave orders the hits with ballots.
// A ray through a 256^3 grid is 4 wave iterations instead of ~300 dependent iterations of a single lane (round 4: one lane per ray walked
// three pyramid levels cell by cell, 0.47 ms per launch for 300 rays).  The arithmetic mirrors oracle/occ_oracle.c: march() operation for
// operation (fp32, this file is compiled with -ffp-contract=off) -> bit-identical samples.
struct RayFrame {
  float go[3], u[3], len, s0, s1;
  int m, a0, a1, k0, dir, nsl;
};
__device__ __forceinline__ int occ_clampi(float v, int res) { const int c = (int)v; return c < 0 ? 0 : (c > res - 1 ? res - 1 : c); }
__device__ __forceinline__ float pick3(const float v[3], int a) { return a == 0 ? v[0] : (a == 1 ? v[1] : v[2]); }

// -> false when the ray misses the cube (or has no direction)
__device__ __forceinline__ bool ray_frame(int res, const float o[3], const float d[3], RayFrame &f) {
  const float half = 0.5f * (float)res;
  float gd[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { f.go[a] = (o[a] + 1.0f) * half; gd[a] = d[a] * half; }
  f.len = sqrtf(gd[0] * gd[0] + gd[1] * gd[1] + gd[2] * gd[2]);
  if (!(f.len > 0.f)) return false;
  float s0 = 0.f, s1 = INFINITY;
  bool live = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    f.u[a] = gd[a] / f.len;
    if (f.u[a] != 0.f) {
      float a0 = (0.f - f.go[a]) / f.u[a], a1 = ((float)res - f.go[a]) / f.u[a];
      if (a0 > a1) { const float t = a0; a0 = a1; a1 = t; }
      s0 = a0 > s0 ? a0 : s0;
      s1 = a1 < s1 ? a1 : s1;
    } else if (f.go[a] < 0.f || f.go[a] >= (float)res) {
      live = false;
    }
  }
  if (!live || !(s0 < s1)) return false;
  f.s0 = s0; f.s1 = s1;
  int m = 0;
  if (fabsf(f.u[1]) > fabsf(f.u[0])) m = 1;
  if (fabsf(f.u[2]) > fabsf(pick3(f.u, m))) m = 2;
  f.m = m; f.a0 = (m + 1) % 3; f.a1 = (m + 2) % 3;
  const float um = pick3(f.u, m), gm = pick3(f.go, m);
  f.k0 = occ_clampi(floorf(gm + um * s0), res);
  const int k1 = occ_clampi(floorf(gm + um * s1), res);
  f.dir = um > 0.f ? 1 : -1;
  f.nsl = (k1 - f.k0) * f.dir + 1;
  return true;
}

// slab i of the ray -> bit j of the result set when sub-interval j is an occupied voxel, crossed for s in [in[j], e[j]]
__device__ __forceinline__ unsigned ray_slab(int L, int res, const OccLevels &lv, const uint32_t *__restrict__ grid, const RayFrame &f, int i,
                                             float in[3], float e[3]) {
  const int k = f.k0 + f.dir * i;
  const float um = pick3(f.u, f.m), gm = pick3(f.go, f.m);
  const float lo = (float)k, hi = lo + 1.0f;
  float sa = ((um > 0.f ? lo : hi) - gm) / um, sb = ((um > 0.f ? hi : lo) - gm) / um;
  if (sa < f.s0) sa = f.s0;
  if (sb > f.s1) sb = f.s1;
  if (!(sb > sa)) return 0u;
  const float ua[2] = {pick3(f.u, f.a0), pick3(f.u, f.a1)}, ga[2] = {pick3(f.go, f.a0), pick3(f.go, f.a1)};
  float tb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    tb[q] = sb;
    if (ua[q] != 0.f) {
      const float ia = floorf(ga[q] + ua[q] * sa), ib = floorf(ga[q] + ua[q] * sb);
      if (ia != ib) {
        float tt = ((ua[q] > 0.f ? ia + 1.0f : ia) - ga[q]) / ua[q];
        if (tt < sa) tt = sa;
        if (tt > sb) tt = sb;
        tb[q] = tt;
      }
    }
  }
  const float b[4] = {sa, tb[0] < tb[1] ? tb[0] : tb[1], tb[0] < tb[1] ? tb[1] : tb[0], sb};
  int c1[3], c2[3];
  bool valid[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    in[j] = b[j]; e[j] = b[j + 1];
    valid[j] = e[j] > in[j];
    const float mid = (in[j] + e[j]) * 0.5f;
    c1[j] = occ_clampi(floorf(ga[0] + ua[0] * mid), res);
    c2[j] = occ_clampi(floorf(ga[1] + ua[1] * mid), res);
  }
  if (valid[0] && valid[1] && c1[0] == c1[1] && c2[0] == c2[1]) { in[1] = in[0]; valid[0] = false; }
  {
    const bool p1 = valid[1];
    const int pc1 = p1 ? c1[1] : c1[0], pc2 = p1 ? c2[1] : c2[0];
    const bool pv = p1 ? valid[1] : valid[0];
    if (valid[2] && pv && pc1 == c1[2] && pc2 == c2[2]) {
      in[2] = p1 ? in[1] : in[0];
      if (p1) valid[1] = false; else valid[0] = false;
    }
  }
  unsigned hit = 0u;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (!valid[j]) continue;
    // (no dynamically indexed array: everything stays in registers)
    const int cx = f.m == 0 ? k : (f.a0 == 0 ? c1[j] : c2[j]);
    const int cy = f.m == 1 ? k : (f.a0 == 1 ? c1[j] : c2[j]);
    const int cz = f.m == 2 ? k : (f.a0 == 2 ? c1[j] : c2[j]);
    if (occ_bit(grid, lv, L, cx, cy, cz)) hit |= 1u << j;
  }
  return hit;
}

// One wave per ray (4 rays per workgroup).  FILL = false: counts[r] = occupied voxels crossed.  FILL = true: writes num_samples
// stratified midpoint samples per crossed voxel at offsets[r] (exclusive scan of the counts, in voxels).
template <bool FILL>
__global__ void __launch_bounds__(256)
    occ_raymarch_kernel(int L, OccLevels lv, int64_t n_rays, const float *__restrict__ origins,
                        const float *__restrict__ dirs, const uint32_t *__restrict__ grid, int32_t *__restrict__ counts,
                        const int64_t *__restrict__ offsets, int num_samples, int32_t *__restrict__ ridx,
                        float *__restrict__ samples, float *__restrict__ depth) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rays) return;   // wave-uniform
  const int lane = threadIdx.x & 63;
  const int res = 1 << L;
  const float o[3] = {origins[3 * r], origins[3 * r + 1], origins[3 * r + 2]};
  const float d[3] = {dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]};
  RayFrame f;
  int n = 0;
  if (ray_frame(res, o, d, f)) {
    const int64_t base = FILL ? offsets[r] : 0;
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int i0 = 0; i0 < f.nsl; i0 += 64) {
      float in[3], e[3];
      const unsigned hit = (i0 + lane < f.nsl) ? ray_slab(L, res, lv, grid, f, i0 + lane, in, e) : 0u;
      const unsigned long long b0 = __ballot(hit & 1u), b1 = __ballot(hit & 2u), b2 = __ballot(hit & 4u);
      if (FILL && hit) {
        int pos = n + __popcll(b0 & lower) + __popcll(b1 & lower) + __popcll(b2 & lower);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (!(hit & (1u << j))) continue;
          const float t_in = in[j] / f.len, t_out = e[j] / f.len;
          for (int q = 0; q < num_samples; ++q) {
            const int64_t row = (base + pos) * num_samples + q;
            const float t = t_in + (t_out - t_in) * (((float)q + 0.5f) / (float)num_samples);
            ridx[row] = (int32_t)r;
            depth[row] = t;
            samples[3 * row] = o[0] + d[0] * t;
            samples[3 * row + 1] = o[1] + d[1] * t;
            samples[3 * row + 2] = o[2] + d[2] * t;
          }
          ++pos;
        }
      }
      n += __popcll(b0) + __popcll(b1) + __popcll(b2);
    }
  }
  if (!FILL && lane == 0) counts[r] = n;
}

// ---- the per-ray SDF batch of the reference in two passes over the rays --------------------------------------------
// NeuralSLAM::sample (/root/reference/include/neural_mapping/neural_mapping.cpp:73-104) = LocalMap::sample (include/neural_net/
// local_map.cpp:449-509: one sample per occupied voxel a ray crosses + free_sample_num stratified free-space samples, those in front of the
// surface kept) + utils::sample_surface_pts (include/utils/utils.cpp:336-364) + truncation of the targets + the rays' end points + the
// in-range filter (sub_map.cpp:37-45).  The reference runs it as ~60 libtorch launches; here: count (per ray and segment) -> one-workgroup
// scan -> fill.  The random draws stay torch's (rand [n,F], randn [n,S]: the same generator state as the reference's op chain), every
// elementwise operation is the reference's in the same order in fp32, and the output rows come in the reference's order:
// [voxel samples, rays in order, front to back | free samples, ray-major | surface samples, ray-major | end points], each filtered.
struct SamplerArgs {
  int L;
  OccLevels lv;
  int64_t n;
  const float *origin, *direction, *depth, *end_xyz;   // world frame [n,3] / [n,1]
  const uint32_t *grid;
  const float *rand_free, *randn_surf;                 // [n,F] U[0,1), [n,S] N(0,1)
  int F, S;
  float pos[3], map_size_inv, map_half;                // m1p1 <-> world: ((x - pos) * 2) * map_size_inv;  (t * 0.5) * map_half + pos
  float lo[3], hi[3];                                  // in-range test xyz > lo && xyz < hi (get_inrange_mask's own bounds)
  float inv_F, sample_std, trunc;
};
struct SamplerOut {
  float *xyz, *ray_sdf, *origin, *direction, *depth;
  int64_t *ridx;
};

__device__ __forceinline__ bool in_range(const SamplerArgs &a, float x, float y, float z) {
  return x < a.hi[0] && x > a.lo[0] && y < a.hi[1] && y > a.lo[1] && z < a.hi[2] && z > a.lo[2];
}
// where(|s| > trunc, sign(s) * trunc, s)
__device__ __forceinline__ float truncated(float s, float trunc) { return fabsf(s) > trunc ? (s > 0.f ? trunc : -trunc) : s; }
__device__ __forceinline__ void put_row(const SamplerOut &o, int64_t row, int64_t r, const float org[3], const float dir[3], float x, float y, float z,
                                        float sdf, float dep) {
  o.xyz[3 * row] = x; o.xyz[3 * row + 1] = y; o.xyz[3 * row + 2] = z;
  o.ray_sdf[row] = sdf;
  o.depth[row] = dep;
  o.ridx[row] = r;
  o.origin[3 * row] = org[0]; o.origin[3 * row + 1] = org[1]; o.origin[3 * row + 2] = org[2];
  o.direction[3 * row] = dir[0]; o.direction[3 * row + 1] = dir[1]; o.direction[3 * row + 2] = dir[2];
}

// counts / offsets are segment-major: [4][n] (voxel, free, surface, end).  FILL: offs_incl = inclusive scan of counts over all 4 n entries.
template <bool FILL>
__global__ void __launch_bounds__(256)
    ray_sampler_kernel(SamplerArgs a, int32_t *__restrict__ counts, const int64_t *__restrict__ offs_incl, SamplerOut out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.n) return;   // wave-uniform
  const int lane = threadIdx.x & 63;
  const unsigned long long lower = (1ull << lane) - 1ull;
  const int res = 1 << a.L;
  const float org[3] = {a.origin[3 * r], a.origin[3 * r + 1], a.origin[3 * r + 2]};
  const float dir[3] = {a.direction[3 * r], a.direction[3 * r + 1], a.direction[3 * r + 2]};
  const float dep = a.depth[r];
  float o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) o[k] = ((org[k] - a.pos[k]) * 2.0f) * a.map_size_inv;      // SubMap::xyz_to_m1p1_pts
  int64_t base[4] = {0, 0, 0, 0};
  if (FILL) {
#pragma unroll
    for (int sgm = 0; sgm < 4; ++sgm) base[sgm] = offs_incl[sgm * a.n + r] - counts[sgm * a.n + r];
  }
  // segment 0: one (or ns) sample(s) in every occupied voxel crossed, kept when in front of the surface and in range
  int n = 0;
  RayFrame f;
  if (ray_frame(res, o, dir, f)) {
    for (int i0 = 0; i0 < f.nsl; i0 += 64) {
      float in[3], e[3];
      const unsigned hit = (i0 + lane < f.nsl) ? ray_slab(a.L, res, a.lv, a.grid, f, i0 + lane, in, e) : 0u;
      // rows in slab order, then sub-interval order: all three keep flags first, then one ballot each
      bool keep[3];
      float px[3], py[3], pz[3], psdf[3], pdd[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        keep[j] = false;
        px[j] = py[j] = pz[j] = psdf[j] = pdd[j] = 0.f;
        if (hit & (1u << j)) {
          const float t_in = in[j] / f.len, t_out = e[j] / f.len;
          const float t = t_in + (t_out - t_in) * ((0.0f + 0.5f) / 1.0f);                  // voxel_sample_num = 1 (local_map.cpp:467)
          px[j] = ((o[0] + dir[0] * t) * 0.5f) * a.map_half + a.pos[0];                  // m1p1_pts_to_xyz(o + d t)
          py[j] = ((o[1] + dir[1] * t) * 0.5f) * a.map_half + a.pos[1];
          pz[j] = ((o[2] + dir[2] * t) * 0.5f) * a.map_half + a.pos[2];
          pdd[j] = (t * 0.5f) * a.map_half;                                              // scale_from_m1p1(depth_samples)
          psdf[j] = dep - pdd[j];
          keep[j] = psdf[j] > 0.f && in_range(a, px[j], py[j], pz[j]);
        }
      }
      const unsigned long long b0 = __ballot(keep[0]), b1 = __ballot(keep[1]), b2 = __ballot(keep[2]);
      if (FILL) {
        int64_t pos = base[0] + n + __popcll(b0 & lower) + __popcll(b1 & lower) + __popcll(b2 & lower);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (keep[j]) put_row(out, pos++, r, org, dir, px[j], py[j], pz[j], truncated(psdf[j], a.trunc), pdd[j]);
      }
      n += __popcll(b0) + __popcll(b1) + __popcll(b2);
    }
  }
  // segment 1: free-space samples, lane k < F
  {
    bool keep = false;
    float x = 0.f, y = 0.f, z = 0.f, sdf = 0.f, dd = 0.f;
    if (lane < a.F) {
      const float steps = ((float)lane + a.rand_free[r * a.F + lane]) * a.inv_F;
      dd = dep * steps;
      sdf = dep - dd;
      x = org[0] + dir[0] * dd; y = org[1] + dir[1] * dd; z = org[2] + dir[2] * dd;
      keep = sdf > 0.f && in_range(a, x, y, z);
    }
    const unsigned long long b = __ballot(keep);
    if (FILL && keep) put_row(out, base[1] + __popcll(b & lower), r, org, dir, x, y, z, truncated(sdf, a.trunc), dd);
    if (!FILL && lane == 0) counts[a.n + r] = __popcll(b);
  }
  // segment 2: near-surface samples at signed distance N(0, std) from the end point, lane k < S
  const float end[3] = {a.end_xyz[3 * r], a.end_xyz[3 * r + 1], a.end_xyz[3 * r + 2]};
  {
    bool keep = false;
    float x = 0.f, y = 0.f, z = 0.f, sdf = 0.f;
    if (lane < a.S) {
      sdf = a.randn_surf[r * a.S + lane] * a.sample_std;
      x = end[0] - dir[0] * sdf; y = end[1] - dir[1] * sdf; z = end[2] - dir[2] * sdf;
      keep = in_range(a, x, y, z);
    }
    const unsigned long long b = __ballot(keep);
    if (FILL && keep) put_row(out, base[2] + __popcll(b & lower), r, org, dir, x, y, z, truncated(sdf, a.trunc), dep);
    if (!FILL && lane == 0) counts[2 * a.n + r] = __popcll(b);
  }
  // segment 3: the ray's end point (target 0)
  {
    const bool keep = in_range(a, end[0], end[1], end[2]);
    if (FILL && keep && lane == 0) put_row(out, base[3], r, org, dir, end[0], end[1], end[2], 0.f, dep);
    if (!FILL && lane == 0) { counts[3 * a.n + r] = keep ? 1 : 0; counts[r] = n; }
  }
}

// inclusive scan of m int32 counts by ONE workgroup (m = 4 n <= a few 100 k: cheaper than the two-launch scan of scan.hip) + the total
__global__ void __launch_bounds__(1024)
    sampler_scan_kernel(int64_t m, const int32_t *__restrict__ counts, int64_t *__restrict__ incl, int64_t *__restrict__ total) {
  __shared__ int64_t s_w[16];
  const int64_t per = (m + 1023) / 1024;
  const int64_t b = (int64_t)threadIdx.x * per, e = b + per < m ? b + per : m;
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += counts[i];
  // block-wide exclusive prefix of the per-thread sums
  int64_t v = s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const int64_t t = __shfl_up(v, dlt, 64);
    if (lane >= dlt) v += t;
  }
  if (lane == 63) s_w[wave] = v;
  __syncthreads();
  int64_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { before += w < wave ? s_w[w] : 0; all += s_w[w]; }
  int64_t run = before + v - s;
  for (int64_t i = b; i < e; ++i) { run += counts[i]; incl[i] = run; }
  if (threadIdx.x == 0) store_host_visible(total, all);
}

}  // namespace gsdf

using namespace gsdf;

static int check_level(int L, const char *who) {
  GSDF_REQUIRE(L >= 1 && L <= 12, "%s: level %d outside [1,12]", who, L);
  return GSDF_OK;
}

extern "C" size_t gsdf_occ_bytes(int level) {
  if (level < 1 || level > 12) return 0;
  return (size_t)occ_level_offset(level + 1) * 4;
}

extern "C" int gsdf_occ_build(int level, int64_t n_points, const float *xyz_m1p1, int dilate27, void *grid,
                              gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_build");
  if (int rc = check_level(level, "occ_build")) return rc;
  GSDF_REQUIRE(grid && (n_points == 0 || xyz_m1p1) && n_points >= 0, "occ_build: bad arguments");
  const OccLevels lv = make_levels(level);
  GSDF_HIP(hipMemsetAsync(grid, 0, gsdf_occ_bytes(level), stream), "occ_build memset");
  if (n_points == 0) return GSDF_OK;
  const int K = dilate27 ? 27 : 1;
  occ_set_kernel<<<(unsigned)((n_points * K + 255) / 256), 256, 0, stream>>>(level, lv, n_points, K, xyz_m1p1,
                                                                              (uint32_t *)grid);
  GSDF_CHECK_LAUNCH("occ_set_kernel");
  for (int l = level - 1; l >= 0; --l) {
    const int64_t r = (int64_t)1 << l;
    const int64_t threads = r >= 32 ? (r / 32) * r * r : r * r * r;
    occ_reduce_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(l, lv, (uint32_t *)grid);
    GSDF_CHECK_LAUNCH("occ_reduce_kernel");
  }
  return GSDF_OK;
}

extern "C" int gsdf_occ_query(int level, int query_level, int64_t n, const float *xyz_m1p1, const void *grid,
                              uint8_t *mask, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_query");
  if (int rc = check_level(level, "occ_query")) return rc;
  const int l = query_level < 0 ? level : query_level;
  GSDF_REQUIRE(l >= 0 && l <= level, "occ_query: query level %d outside [0,%d]", l, level);
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(n > 0 && xyz_m1p1 && grid && mask, "occ_query: bad arguments");
  occ_query_kernel<false><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(l, make_levels(level), n, xyz_m1p1, 0.f, 0.f, 0.f,
                                                                           1.f, (const uint32_t *)grid, mask);
  GSDF_CHECK_LAUNCH("occ_query_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_occ_query_world(int level, int query_level, int64_t n, const float *xyz_world, const float *origin_host,
                                    float map_size_inv, const void *grid, uint8_t *mask, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_query_world");
  if (int rc = check_level(level, "occ_query_world")) return rc;
  const int l = query_level < 0 ? level : query_level;
  GSDF_REQUIRE(l >= 0 && l <= level, "occ_query_world: query level %d outside [0,%d]", l, level);
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(n > 0 && xyz_world && origin_host && grid && mask, "occ_query_world: bad arguments");
  occ_query_kernel<true><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(l, make_levels(level), n, xyz_world, origin_host[0],
                                                                          origin_host[1], origin_host[2], map_size_inv,
                                                                          (const uint32_t *)grid, mask);
  GSDF_CHECK_LAUNCH("occ_query_kernel<world>");
  return GSDF_OK;
}

extern "C" size_t gsdf_visible_set_ws_bytes(int64_t n) { return (size_t)n + 4 * (size_t)((n + VS_ROWS - 1) / VS_ROWS) + 512; }

extern "C" int gsdf_visible_set(int level, int query_level, int64_t n, const float *xyz_world, const float *origin_host, float map_size_inv,
                                const void *grid, const float *visibilities, const float *samples_weights, float vis_thresh, float *w_all,
                                int64_t *ids, int64_t *count, void *ws, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_visible_set");
  if (int rc = check_level(level, "visible_set")) return rc;
  const int l = query_level < 0 ? level : query_level;
  GSDF_REQUIRE(l >= 0 && l <= level, "visible_set: query level %d outside [0,%d]", l, level);
  GSDF_REQUIRE(n >= 0 && count, "visible_set: bad arguments");
  if (n == 0) { GSDF_HIP(hipMemsetAsync(count, 0, sizeof(int64_t), stream), "visible_set memset"); return GSDF_OK; }
  GSDF_REQUIRE(xyz_world && origin_host && grid && visibilities && samples_weights && w_all && ids && ws, "visible_set: null buffer");
  GSDF_REQUIRE(n < ((int64_t)1 << 31), "visible_set: too many rows");
  const int nblk = (int)((n + VS_ROWS - 1) / VS_ROWS);
  int32_t *counts = (int32_t *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  uint8_t *flags = (uint8_t *)(counts + nblk);
  visible_flags_kernel<<<nblk, 256, 0, stream>>>(l, make_levels(level), n, xyz_world, origin_host[0], origin_host[1], origin_host[2], map_size_inv,
                                                 (const uint32_t *)grid, visibilities, samples_weights, vis_thresh, w_all, flags, counts);
  GSDF_CHECK_LAUNCH("visible_flags_kernel");
  visible_scan_kernel<<<1, 1024, 0, stream>>>(nblk, counts, count);
  GSDF_CHECK_LAUNCH("visible_scan_kernel");
  visible_write_kernel<<<nblk, 256, 0, stream>>>(n, flags, counts, ids);
  GSDF_CHECK_LAUNCH("visible_write_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_occ_voxel_counts(int level, const void *grid, int32_t *word_counts, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_voxel_counts");
  if (int rc = check_level(level, "occ_voxel_counts")) return rc;
  GSDF_REQUIRE(grid && word_counts, "occ_voxel_counts: null buffer");
  const int64_t nw = occ_level_words(level);
  occ_popc_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, stream>>>(nw, (const uint32_t *)grid + occ_level_offset(level),
                                                                    word_counts);
  GSDF_CHECK_LAUNCH("occ_popc_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_occ_voxel_list(int level, const void *grid, const int64_t *word_offsets, int16_t *voxels,
                                   gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_voxel_list");
  if (int rc = check_level(level, "occ_voxel_list")) return rc;
  GSDF_REQUIRE(grid && word_offsets && voxels, "occ_voxel_list: null buffer");
  const int64_t nw = occ_level_words(level);
  occ_list_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, stream>>>(level, nw,
                                                                    (const uint32_t *)grid + occ_level_offset(level),
                                                                    word_offsets, voxels);
  GSDF_CHECK_LAUNCH("occ_list_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_occ_raymarch_count(int level, int64_t n_rays, const float *origins_m1p1, const float *dirs,
                                       const void *grid, int32_t *counts, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_raymarch_count");
  if (int rc = check_level(level, "occ_raymarch_count")) return rc;
  if (n_rays == 0) return GSDF_OK;
  GSDF_REQUIRE(n_rays > 0 && origins_m1p1 && dirs && grid && counts, "occ_raymarch_count: bad arguments");
  occ_raymarch_kernel<false><<<(unsigned)((n_rays + 3) / 4), 256, 0, stream>>>(
      level, make_levels(level), n_rays, origins_m1p1, dirs, (const uint32_t *)grid, counts, nullptr, 0, nullptr, nullptr,
      nullptr);
  GSDF_CHECK_LAUNCH("occ_raymarch_kernel<count>");
  return GSDF_OK;
}

extern "C" int gsdf_occ_raymarch_fill(int level, int64_t n_rays, const float *origins_m1p1, const float *dirs,
                                      const void *grid, const int64_t *voxel_offsets, int num_samples, int32_t *ridx,
                                      float *samples, float *depth_samples, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_occ_raymarch_fill");
  if (int rc = check_level(level, "occ_raymarch_fill")) return rc;
  if (n_rays == 0) return GSDF_OK;
  GSDF_REQUIRE(n_rays > 0 && num_samples >= 1 && origins_m1p1 && dirs && grid && voxel_offsets && ridx && samples &&
                   depth_samples,
               "occ_raymarch_fill: bad arguments");
  occ_raymarch_kernel<true><<<(unsigned)((n_rays + 3) / 4), 256, 0, stream>>>(
      level, make_levels(level), n_rays, origins_m1p1, dirs, (const uint32_t *)grid, nullptr, voxel_offsets, num_samples,
      ridx, samples, depth_samples);
  GSDF_CHECK_LAUNCH("occ_raymarch_kernel<fill>");
  return GSDF_OK;
}

static int sampler_args(const gsdf_ray_sampler_args *g, SamplerArgs *a, const char *who) {
  GSDF_REQUIRE(g != nullptr, "%s: null arguments", who);
  if (int rc = check_level(g->level, who)) return rc;
  GSDF_REQUIRE(g->n_rays >= 0 && g->n_rays < ((int64_t)1 << 29), "%s: ray count %lld out of range", who, (long long)g->n_rays);
  GSDF_REQUIRE(g->free_sample_num >= 0 && g->free_sample_num <= 64 && g->surface_sample_num >= 0 && g->surface_sample_num <= 64,
               "%s: free / surface sample numbers must be in [0, 64] (one lane of the ray's wave each)", who);
  GSDF_REQUIRE(g->n_rays == 0 || (g->origin && g->direction && g->depth && g->end_xyz && g->grid && (g->rand_free || g->free_sample_num == 0) &&
                                  (g->randn_surf || g->surface_sample_num == 0)), "%s: null buffer", who);
  a->L = g->level; a->lv = make_levels(g->level); a->n = g->n_rays;
  a->origin = g->origin; a->direction = g->direction; a->depth = g->depth; a->end_xyz = g->end_xyz;
  a->grid = (const uint32_t *)g->grid; a->rand_free = g->rand_free; a->randn_surf = g->randn_surf;
  a->F = g->free_sample_num; a->S = g->surface_sample_num;
  for (int k = 0; k < 3; ++k) { a->pos[k] = g->map_origin[k]; a->lo[k] = g->range_lo[k]; a->hi[k] = g->range_hi[k]; }
  a->map_size_inv = g->map_size_inv; a->map_half = g->map_half;
  a->inv_F = g->free_sample_num > 0 ? 1.0f / (float)g->free_sample_num : 0.f;   // the reference multiplies by 1.0f / sample_num (utils.cpp:374-376)
  a->sample_std = g->sample_std; a->trunc = g->truncated_dis;
  return GSDF_OK;
}

extern "C" int gsdf_ray_sampler_count(const gsdf_ray_sampler_args *g, int32_t *counts, int64_t *offsets_incl, int64_t *total,
                                      gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_ray_sampler_count");
  SamplerArgs a;
  if (int rc = sampler_args(g, &a, "ray_sampler_count")) return rc;
  GSDF_REQUIRE(total != nullptr, "ray_sampler_count: null total");
  if (a.n == 0) { GSDF_HIP(hipMemsetAsync(total, 0, sizeof(int64_t), stream), "ray_sampler_count memset"); return GSDF_OK; }
  GSDF_REQUIRE(counts && offsets_incl, "ray_sampler_count: null buffer");
  ray_sampler_kernel<false><<<(unsigned)((a.n + 3) / 4), 256, 0, stream>>>(a, counts, nullptr, SamplerOut{});
  GSDF_CHECK_LAUNCH("ray_sampler_kernel<count>");
  sampler_scan_kernel<<<1, 1024, 0, stream>>>(4 * a.n, counts, offsets_incl, total);
  GSDF_CHECK_LAUNCH("sampler_scan_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_ray_sampler_fill(const gsdf_ray_sampler_args *g, const int32_t *counts, const int64_t *offsets_incl, float *xyz,
                                     float *ray_sdf, int64_t *ridx, float *origin, float *direction, float *depth, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_ray_sampler_fill");
  SamplerArgs a;
  if (int rc = sampler_args(g, &a, "ray_sampler_fill")) return rc;
  if (a.n == 0) return GSDF_OK;
  GSDF_REQUIRE(counts && offsets_incl && xyz && ray_sdf && ridx && origin && direction && depth, "ray_sampler_fill: null buffer");
  SamplerOut o{xyz, ray_sdf, origin, direction, depth, ridx};
  ray_sampler_kernel<true><<<(unsigned)((a.n + 3) / 4), 256, 0, stream>>>(a, const_cast<int32_t *>(counts), offsets_incl, o);
  GSDF_CHECK_LAUNCH("ray_sampler_kernel<fill>");
  return GSDF_OK;
}
