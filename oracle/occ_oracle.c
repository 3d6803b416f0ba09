/* occ_oracle.c — CPU restatement of the occupancy acceleration structure.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED: the reference keeps this structure in the absent
 * submodule jianhengLiu/kaolin_wisp_cpp (-> NVIDIAGameWorks/kaolin SPC octree, pinned version unknown); what is restated
 * here is the behaviour its call sites rely on:
 *   spc_ops::quantize_points / points_to_neighbors (27) / from_quantized_points   /root/reference/include/neural_net/sub_map.cpp:22-35
 *   OctreeAS::query(xyz, level).pidx > -1                                         sub_map.cpp:76-80, local_map.cpp:511-516
 *   OctreeAS::raymarch(origin, dir, "voxel", n) -> {ridx, samples, depth_samples} local_map.cpp:467-476
 *   OctreeAS::get_quantized_points()                                              neural_mapping.cpp:755-758
 * with kaolin's published conventions: coordinates in [-1,1]^3, q = clamp(floor(2^L (x+1)/2), 0, 2^L-1), rays visit the
 * occupied level-L voxels front to back, depth = ray parameter t of origin + t*dir.  Decisions that cannot be recovered
 * from the call sites (DESIGN.md SPEC A.9): the n samples of a voxel sit at the stratified midpoints
 * t_in + (t_out - t_in)(k + 1/2)/n; voxels a ray touches for less than 1e-3 of a cell may be skipped; a query outside
 * [-1,1]^3 is "not occupied".
 *
 * Representation (the MI355X design, mirrored here bit for bit): a bit pyramid, level l = 2^(3l) bits in x-fastest
 * order packed into uint32 words, levels 0..L back to back; a parent bit is the OR of its 8 children.
 * All arithmetic is fp32 with no contraction, in the same order as gs-sdf_amd/csrc/occupancy.hip: integer outputs
 * (counts, ray ids, voxel lists, masks) are bit-exact parity targets and the sample floats follow from them.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static int64_t level_words(int l) {
  const int64_t bits = (int64_t)1 << (3 * l);
  return bits < 32 ? 1 : bits / 32;
}
static int64_t level_offset(int l) {
  int64_t o = 0;
  for (int k = 0; k < l; ++k) o += level_words(k);
  return o;
}
int64_t orc_occ_words(int L) { return level_offset(L + 1); }

static int get_bit(const uint32_t *grid, int l, int x, int y, int z) {
  const int64_t idx = ((((int64_t)z << l) + y) << l) + x;
  return (grid[level_offset(l) + (idx >> 5)] >> (idx & 31)) & 1u;
}
static void set_bit(uint32_t *grid, int l, int x, int y, int z) {
  const int64_t idx = ((((int64_t)z << l) + y) << l) + x;
  grid[level_offset(l) + (idx >> 5)] |= 1u << (idx & 31);
}

static int quantize(float x, int res) {
  float q = floorf((float)res * (x + 1.0f) / 2.0f);
  if (q < 0.f) q = 0.f;
  if (q > (float)(res - 1)) q = (float)(res - 1);
  return (int)q;
}

/* from_quantized_points(points_to_neighbors?(unique(quantize_points(xyz)))) */
void orc_occ_build(int L, int64_t n, const float *xyz, int dilate27, uint32_t *grid) {
  const int res = 1 << L;
  memset(grid, 0, (size_t)orc_occ_words(L) * 4);
  for (int64_t i = 0; i < n; ++i) {
    const int q[3] = {quantize(xyz[3 * i], res), quantize(xyz[3 * i + 1], res), quantize(xyz[3 * i + 2], res)};
    const int r = dilate27 ? 1 : 0;
    for (int dz = -r; dz <= r; ++dz)
      for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
          int x = q[0] + dx, y = q[1] + dy, z = q[2] + dz; /* .clamp(0, res - 1) */
          x = x < 0 ? 0 : (x > res - 1 ? res - 1 : x);
          y = y < 0 ? 0 : (y > res - 1 ? res - 1 : y);
          z = z < 0 ? 0 : (z > res - 1 ? res - 1 : z);
          set_bit(grid, L, x, y, z);
        }
  }
  for (int l = L - 1; l >= 0; --l) {
    const int r = 1 << l;
    for (int z = 0; z < r; ++z)
      for (int y = 0; y < r; ++y)
        for (int x = 0; x < r; ++x) {
          int any = 0;
          for (int c = 0; c < 8; ++c) any |= get_bit(grid, l + 1, 2 * x + (c & 1), 2 * y + ((c >> 1) & 1), 2 * z + (c >> 2));
          if (any) set_bit(grid, l, x, y, z);
        }
  }
}

/* query(xyz, level).pidx > -1 ; query_level < 0 means L */
void orc_occ_query(int L, int query_level, int64_t n, const float *xyz, const uint32_t *grid, uint8_t *mask) {
  const int l = query_level < 0 ? L : query_level;
  const int res = 1 << l;
  for (int64_t i = 0; i < n; ++i) {
    int ok = 1, q[3];
    for (int a = 0; a < 3; ++a) {
      const float x = xyz[3 * i + a];
      if (!(x >= -1.0f && x <= 1.0f)) ok = 0;
      q[a] = quantize(x, res);
    }
    mask[i] = (uint8_t)(ok && get_bit(grid, l, q[0], q[1], q[2]));
  }
}

/* get_quantized_points(): occupied level-L voxels in x-fastest linear order; returns their number */
int64_t orc_occ_list(int L, const uint32_t *grid, int16_t *out) {
  const int res = 1 << L;
  int64_t v = 0;
  for (int z = 0; z < res; ++z)
    for (int y = 0; y < res; ++y)
      for (int x = 0; x < res; ++x)
        if (get_bit(grid, L, x, y, z)) {
          if (out) { out[3 * v] = (int16_t)x; out[3 * v + 1] = (int16_t)y; out[3 * v + 2] = (int16_t)z; }
          ++v;
        }
  return v;
}

/* One ray: the occupied level-L voxels it crosses, front to back, by SLABS of the ray's major axis (DESIGN.md SPEC A.9, round 5).
 * Grid frame: g(s) = go + u s, |u| = 1, s in [s0, s1] = the ray inside the cube [0, res]^3.  m = the axis with the largest |u| (first on
 * ties).  Slab k (cells with c[m] = k) is crossed for s in [sa, sb] = the two planes g_m = k, k + 1, clipped to [s0, s1]; inside a slab
 * the ray advances at most one cell along either minor axis, so the slab holds at most three cells, separated by at most one integer
 * crossing per minor axis (t = (plane - go[a]) / u[a], clamped to the slab).  Each sub-interval of positive length is one cell: its minor
 * coordinates are floor(g(mid)) of the sub-interval's midpoint (robust: never decided at a boundary), neighbouring sub-intervals that land
 * in the same cell are merged.  An occupied cell reports (t_in, t_out) = its sub-interval / len.  Slabs are independent of each other:
 * the HIP kernel (csrc/occupancy.hip: occ_march_wave) gives one slab to each lane of a wave and mirrors this arithmetic operation for
 * operation (fp32, no contraction) -> bit-identical samples.
 * Returns the number of voxels; when t_io != NULL writes their (t_in, t_out). */
static inline int clampi(float v, int res) { const int c = (int)v; return c < 0 ? 0 : (c > res - 1 ? res - 1 : c); }
static int march(int L, const float *o, const float *d, const uint32_t *grid, float *t_io) {
  const int res = 1 << L;
  const float half = 0.5f * (float)res;
  float go[3], gd[3], u[3];
  for (int a = 0; a < 3; ++a) { go[a] = (o[a] + 1.0f) * half; gd[a] = d[a] * half; }
  const float len = sqrtf(gd[0] * gd[0] + gd[1] * gd[1] + gd[2] * gd[2]);
  if (!(len > 0.f)) return 0;
  for (int a = 0; a < 3; ++a) u[a] = gd[a] / len;
  /* clip to the cube [0,res]^3, s = arc length in cells */
  float s0 = 0.f, s1 = INFINITY;
  for (int a = 0; a < 3; ++a) {
    if (u[a] != 0.f) {
      float a0 = (0.f - go[a]) / u[a], a1 = ((float)res - go[a]) / u[a];
      if (a0 > a1) { const float t = a0; a0 = a1; a1 = t; }
      if (a0 > s0) s0 = a0;
      if (a1 < s1) s1 = a1;
    } else if (go[a] < 0.f || go[a] >= (float)res) {
      return 0;
    }
  }
  if (!(s0 < s1)) return 0;
  int m = 0;
  if (fabsf(u[1]) > fabsf(u[m])) m = 1;
  if (fabsf(u[2]) > fabsf(u[m])) m = 2;
  const int ax[2] = {(m + 1) % 3, (m + 2) % 3};
  const float um = u[m];
  const int k0 = clampi(floorf(go[m] + um * s0), res), k1 = clampi(floorf(go[m] + um * s1), res);
  const int dir = um > 0.f ? 1 : -1;
  const int nsl = (k1 - k0) * dir + 1;
  int n = 0;
  for (int i = 0; i < nsl; ++i) {
    const int k = k0 + dir * i;
    const float lo = (float)k, hi = lo + 1.0f;
    float sa = ((um > 0.f ? lo : hi) - go[m]) / um, sb = ((um > 0.f ? hi : lo) - go[m]) / um;
    if (sa < s0) sa = s0;
    if (sb > s1) sb = s1;
    if (!(sb > sa)) continue;
    float tb[2];
    for (int q = 0; q < 2; ++q) {
      const int a = ax[q];
      tb[q] = sb;
      if (u[a] != 0.f) {
        const float ia = floorf(go[a] + u[a] * sa), ib = floorf(go[a] + u[a] * sb);
        if (ia != ib) {
          float tt = ((u[a] > 0.f ? ia + 1.0f : ia) - go[a]) / u[a];
          if (tt < sa) tt = sa;
          if (tt > sb) tt = sb;
          tb[q] = tt;
        }
      }
    }
    const float b[4] = {sa, tb[0] < tb[1] ? tb[0] : tb[1], tb[0] < tb[1] ? tb[1] : tb[0], sb};
    float in[3], e[3];
    int c1[3], c2[3], valid[3];
    for (int j = 0; j < 3; ++j) {
      in[j] = b[j]; e[j] = b[j + 1];
      valid[j] = e[j] > in[j];
      const float mid = (in[j] + e[j]) * 0.5f;
      c1[j] = clampi(floorf(go[ax[0]] + u[ax[0]] * mid), res);
      c2[j] = clampi(floorf(go[ax[1]] + u[ax[1]] * mid), res);
    }
    /* merge neighbours that landed in the same cell (a breakpoint that did not separate two cells) */
    if (valid[0] && valid[1] && c1[0] == c1[1] && c2[0] == c2[1]) { in[1] = in[0]; valid[0] = 0; }
    {
      const int pj = valid[1] ? 1 : 0;
      if (valid[2] && valid[pj] && c1[pj] == c1[2] && c2[pj] == c2[2]) { in[2] = in[pj]; valid[pj] = 0; }
    }
    for (int j = 0; j < 3; ++j) {
      if (!valid[j]) continue;
      int c[3];
      c[m] = k; c[ax[0]] = c1[j]; c[ax[1]] = c2[j];
      if (!get_bit(grid, L, c[0], c[1], c[2])) continue;
      if (t_io) { t_io[2 * n] = in[j] / len; t_io[2 * n + 1] = e[j] / len; }
      ++n;
    }
  }
  return n;
}

/* phase 1: counts[r] = number of voxels ray r crosses (samples = counts * num_samples) */
void orc_occ_raymarch_count(int L, int64_t n_rays, const float *origins, const float *dirs, const uint32_t *grid,
                            int32_t *counts) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < n_rays; ++r) counts[r] = march(L, origins + 3 * r, dirs + 3 * r, grid, NULL);
}

/* phase 2: offsets[r] = exclusive prefix sum of counts (in voxels).  Outputs sized sum(counts)*num_samples. */
void orc_occ_raymarch_fill(int L, int64_t n_rays, const float *origins, const float *dirs, const uint32_t *grid,
                           const int64_t *offsets, const int32_t *counts, int num_samples, int32_t *ridx, float *samples,
                           float *depth) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < n_rays; ++r) {
    if (counts[r] == 0) continue;
    float t_io[2 * 4096];
    float *buf = t_io;
    float *heap = NULL;
    if (counts[r] > 4096) { heap = (float *)__builtin_malloc((size_t)counts[r] * 8); buf = heap; }
    const int n = march(L, origins + 3 * r, dirs + 3 * r, grid, buf);
    for (int v = 0; v < n; ++v)
      for (int k = 0; k < num_samples; ++k) {
        const int64_t j = (offsets[r] + v) * num_samples + k;
        const float t = buf[2 * v] + (buf[2 * v + 1] - buf[2 * v]) * (((float)k + 0.5f) / (float)num_samples);
        ridx[j] = (int32_t)r;
        depth[j] = t;
        for (int a = 0; a < 3; ++a) samples[3 * j + a] = origins[3 * r + a] + dirs[3 * r + a] * t;
      }
    if (heap) __builtin_free(heap);
  }
}
