// hashgrid_common.h — level geometry and cell addressing shared by the hash-grid kernels (hashgrid.hip, hashgrid_binned.hip).
// Semantics: DESIGN.md SPEC A.7 (tiny-cuda-nn GridEncoding: Hash, Linear, 3-D, F = 2).
#pragma once
#include <math.h>

#include "common.h"

namespace gsdf {

static constexpr int HG_MAX_LEVELS = 16;
static constexpr int HG_THREADS = 256;  // 16 points x 16 levels

struct HgLevels {
  float scale[HG_MAX_LEVELS];
  uint32_t res[HG_MAX_LEVELS];
  uint32_t hsize[HG_MAX_LEVELS];
  uint32_t offset[HG_MAX_LEVELS];  // in entries
  int n_levels;
};

static float level_scale_host(int l, float per_level_scale, int base_res) {
  return exp2f((float)l * log2f(per_level_scale)) * (float)base_res - 1.0f;
}

static int64_t build_levels(int n_levels, int log2_hashmap, int base_res, float per_level_scale, HgLevels *lv,
                            int64_t *offsets_out) {
  int64_t off = 0;
  for (int l = 0; l < n_levels; ++l) {
    const float scale = level_scale_host(l, per_level_scale, base_res);
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    const double dense = pow((double)res, 3.0);
    const uint64_t max_params = 0xFFFFFFFFu / 2;
    uint64_t p = dense > (double)max_params ? max_params : (uint64_t)res * res * res;
    p = (p + 7) / 8 * 8;
    const uint64_t cap = 1ull << log2_hashmap;
    if (p > cap) p = cap;
    if (lv) { lv->scale[l] = scale; lv->res[l] = res; lv->hsize[l] = (uint32_t)p; lv->offset[l] = (uint32_t)off; }
    if (offsets_out) offsets_out[l] = off;
    off += (int64_t)p;
  }
  if (lv) lv->n_levels = n_levels;
  if (offsets_out) offsets_out[n_levels] = off;
  return off;
}

__device__ __forceinline__ uint32_t grid_index(uint32_t hsize, uint32_t res, uint32_t gx, uint32_t gy, uint32_t gz) {
  // dense index while the stride still fits the table (tiny-cuda-nn grid_index), else coherent prime hash
  uint32_t stride = 1, index = 0;
  if (stride <= hsize) { index += gx * stride; stride *= res; }
  if (stride <= hsize) { index += gy * stride; stride *= res; }
  if (stride <= hsize) { index += gz * stride; stride *= res; }
  if (hsize < stride) index = gx ^ (gy * 2654435761u) ^ (gz * 805459861u);
  // table sizes are powers of two for every level of the reference configuration (32^3, 64^3, 2^19)
  return (hsize & (hsize - 1u)) == 0u ? (index & (hsize - 1u)) : (index % hsize);
}

struct Cell {
  uint32_t g0[3];
  float fr[3];
  float scale;
  uint32_t res, hsize;
  const float2 *base;
};

__device__ __forceinline__ bool load_cell(const HgLevels &lv, int level, const float *__restrict__ x, int64_t b,
                                          const float *__restrict__ table, Cell &c) {
  c.scale = lv.scale[level];
  c.res = lv.res[level];
  c.hsize = lv.hsize[level];
  c.base = reinterpret_cast<const float2 *>(table) + lv.offset[level];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float pos = fmaf(c.scale, x[3 * b + d], 0.5f);
    const float fl = floorf(pos);
    c.g0[d] = (uint32_t)(int32_t)fl;
    c.fr[d] = pos - fl;
  }
  return true;
}

}  // namespace gsdf
