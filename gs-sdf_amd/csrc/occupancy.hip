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
  if (threadIdx.x == 0) { *total = s_carry; __threadfence_system(); }   // (the count may be a host-visible word: gsdf_host_words_alloc)
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

#define OCC_EPS 1e-3f

// One thread per ray.  FILL = false: counts[r] = occupied voxels crossed.  FILL = true: writes num_samples stratified
// midpoint samples per crossed voxel at offsets[r] (exclusive scan of the counts, in voxels).
template <bool FILL>
__global__ void __launch_bounds__(256)
    occ_raymarch_kernel(int L, OccLevels lv, int64_t n_rays, const float *__restrict__ origins,
                        const float *__restrict__ dirs, const uint32_t *__restrict__ grid, int32_t *__restrict__ counts,
                        const int64_t *__restrict__ offsets, int num_samples, int32_t *__restrict__ ridx,
                        float *__restrict__ samples, float *__restrict__ depth) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  const int res = 1 << L;
  const float half = 0.5f * (float)res;
  const float o[3] = {origins[3 * r], origins[3 * r + 1], origins[3 * r + 2]};
  const float d[3] = {dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]};
  float go[3], gd[3], u[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { go[a] = (o[a] + 1.0f) * half; gd[a] = d[a] * half; }
  const float len = sqrtf(gd[0] * gd[0] + gd[1] * gd[1] + gd[2] * gd[2]);
  int n = 0;
  bool live = len > 0.f;
  float s0 = 0.f, s1 = INFINITY;
  if (live) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u[a] = gd[a] / len;
      if (u[a] != 0.f) {
        float a0 = (0.f - go[a]) / u[a], a1 = ((float)res - go[a]) / u[a];
        if (a0 > a1) { const float t = a0; a0 = a1; a1 = t; }
        s0 = a0 > s0 ? a0 : s0;
        s1 = a1 < s1 ? a1 : s1;
      } else if (go[a] < 0.f || go[a] >= (float)res) {
        live = false;
      }
    }
    live = live && s0 < s1;
  }
  if (live) {
    const int64_t base = FILL ? offsets[r] : 0;
    const int lvl[3] = {L - 6, L - 3, L};
    float s = s0;
    for (int it = 0; it < 8 * res + 64 && s < s1; ++it) {
      const float sp = s + OCC_EPS;
      int c[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int v = (int)floorf(go[a] + u[a] * sp);
        c[a] = v < 0 ? 0 : (v > res - 1 ? res - 1 : v);
      }
      float s_out = s1;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int l = lvl[k];
        if (l < 1) continue;
        const int sh = L - l;
        const unsigned occ = occ_bit(grid, lv, l, c[0] >> sh, c[1] >> sh, c[2] >> sh);
        if (occ && l < L) continue;  // descend
        float e = INFINITY, in = -INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (u[a] == 0.f) continue;
          const float lo = (float)((c[a] >> sh) << sh), hi = lo + (float)(1 << sh);
          const float ex = ((u[a] > 0.f ? hi : lo) - go[a]) / u[a], en = ((u[a] > 0.f ? lo : hi) - go[a]) / u[a];
          e = ex < e ? ex : e;
          in = en > in ? en : in;
        }
        s_out = e;
        if (occ) {
          in = in < s0 ? s0 : in;
          e = e > s1 ? s1 : e;
          if (e > in) {
            if (FILL) {
              const float t_in = in / len, t_out = e / len;
              for (int q = 0; q < num_samples; ++q) {
                const int64_t j = (base + n) * num_samples + q;
                const float t = t_in + (t_out - t_in) * (((float)q + 0.5f) / (float)num_samples);
                ridx[j] = (int32_t)r;
                depth[j] = t;
                samples[3 * j] = o[0] + d[0] * t;
                samples[3 * j + 1] = o[1] + d[1] * t;
                samples[3 * j + 2] = o[2] + d[2] * t;
              }
            }
            ++n;
          }
        }
        break;
      }
      s = s_out > sp ? s_out : sp;  // always progress
    }
  }
  if (!FILL) counts[r] = n;
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
  occ_raymarch_kernel<false><<<(unsigned)((n_rays + 255) / 256), 256, 0, stream>>>(
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
  occ_raymarch_kernel<true><<<(unsigned)((n_rays + 255) / 256), 256, 0, stream>>>(
      level, make_levels(level), n_rays, origins_m1p1, dirs, (const uint32_t *)grid, nullptr, voxel_offsets, num_samples,
      ridx, samples, depth_samples);
  GSDF_CHECK_LAUNCH("occ_raymarch_kernel<fill>");
  return GSDF_OK;
}
