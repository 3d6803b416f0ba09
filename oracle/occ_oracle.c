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

#define OCC_EPS 1e-3f

/* One ray.  Returns the number of occupied level-L voxels it crosses; when t_io != NULL writes their (t_in, t_out). */
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
  const int lv[3] = {L - 6, L - 3, L};
  float s = s0;
  int n = 0;
  for (int it = 0; it < 8 * res + 64 && s < s1; ++it) {
    const float sp = s + OCC_EPS;
    int c[3];
    for (int a = 0; a < 3; ++a) {
      int v = (int)floorf(go[a] + u[a] * sp);
      c[a] = v < 0 ? 0 : (v > res - 1 ? res - 1 : v);
    }
    int hit = 0;
    float s_out = s1;
    for (int k = 0; k < 3; ++k) {
      const int l = lv[k];
      if (l < 1) continue;
      const int sh = L - l;
      const int occ = get_bit(grid, l, c[0] >> sh, c[1] >> sh, c[2] >> sh);
      if (occ && l < L) continue; /* descend */
      /* empty cell at level l (skip it), or occupied voxel at level L (record it): exit distance of that cell */
      float e = INFINITY, in = -INFINITY;
      for (int a = 0; a < 3; ++a) {
        if (u[a] == 0.f) continue;
        const float lo = (float)((c[a] >> sh) << sh), hi = lo + (float)(1 << sh);
        const float ex = ((u[a] > 0.f ? hi : lo) - go[a]) / u[a], en = ((u[a] > 0.f ? lo : hi) - go[a]) / u[a];
        if (ex < e) e = ex;
        if (en > in) in = en;
      }
      s_out = e;
      if (occ) {
        hit = 1;
        if (in < s0) in = s0;
        if (e > s1) e = s1;
        if (e > in) {
          if (t_io) { t_io[2 * n] = in / len; t_io[2 * n + 1] = e / len; }
          ++n;
        }
      }
      break;
    }
    (void)hit;
    s = s_out > sp ? s_out : sp; /* always progress */
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
