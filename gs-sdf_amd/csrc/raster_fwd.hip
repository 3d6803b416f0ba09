// raster_fwd.hip — P4: 2DGS alpha compositing, forward.
// Replaces rasterize_to_pixels_2dgs of the reference's absent gsplat_cpp submodule (call site
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223); semantics: SPEC A.4.
#include "raster_quad.h"

namespace gsdf {

struct FwdLds {
  SplatBatch s;
  unsigned short dbg4x4[RT], dbg8x2[RT];   // COUNT build only: per staged splat, reach masks at 4x4 / 8x2 granularity
  unsigned vis[4][RT];  // per wave, per staged splat: max blending weight over the wave's pixels (fp32 bits);
                        // every (wave, splat) pair is visited once per batch -> plain stores, no LDS atomics
};

// COUNT: diagnostic instantiation (gsdf_rasterize_2dgs_fwd_instr with counters): per launch, counters[0] += (wave, splat) visits after the
// quadrant mask, [1] += lanes of those visits whose pixel is still live, [2] += lanes that pass the alpha test, [3] += lanes that blend.
// TRACE: the decision record of the parity gate (gsdf_raster_instr.trace_*): a pixel with trace_rows[pid] >= 0 writes one byte per list
// position it takes a decision at — bit0 blended, bit1 3-D footprint branch, bit2 alpha clamped, bit3 the pixel terminates at this pair
// (not blended), bit4 the median is updated here; positions it skips (alpha test failed, unreachable quadrant, pixel finished) stay 0.
template <bool COUNT, bool TRACE>
__global__ void __launch_bounds__(RT)
    raster_fwd_kernel(int n_xcd, int64_t total_tiles, int64_t n_tiles, int64_t I, int W, int H, int tw,
                      const float *__restrict__ means2d, const float *__restrict__ ray_transforms,
                      const float *__restrict__ colors, const float *__restrict__ opacities,
                      const float *__restrict__ normals, const float *__restrict__ backgrounds,
                      const uint8_t *__restrict__ masks, const int32_t *__restrict__ isect_offsets,
                      const int32_t *__restrict__ flatten_ids, float *__restrict__ render_colors,
                      float *__restrict__ render_depths, float *__restrict__ render_alphas,
                      float *__restrict__ render_normals, float *__restrict__ render_median,
                      int32_t *__restrict__ last_ids, int32_t *__restrict__ median_ids,
                      unsigned *__restrict__ visibilities, float *__restrict__ final_T, unsigned long long *__restrict__ counters,
                      const int32_t *__restrict__ trace_rows, int trace_stride, uint8_t *__restrict__ trace_bits) {
  __shared__ FwdLds lds;
  unsigned long long c_visit = 0, c_live = 0, c_ok = 0, c_blend = 0, c_empty = 0, c_r4 = 0, c_r8 = 0, c_union = 0;
  const int64_t tile = xcd_tile_index(total_tiles, n_xcd);
  if (tile >= total_tiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t cam = tile / n_tiles;
  const int tl = (int)(tile - cam * n_tiles);
  const int ty = tl / tw, tx = tl - ty * tw;
  const int x = tx * TILE + (wave & 1) * 8 + (lane & 7);
  const int y = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = x < W && y < H;
  const int64_t pid = (cam * H + y) * (int64_t)W + x;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float lx = (float)((wave & 1) * 8 + (lane & 7)), ly = (float)((wave >> 1) * 8 + (lane >> 3));
  const int sub_bit = 4 * wave + 2 * ((lane >> 5) & 1) + ((lane >> 2) & 1);   // this pixel's 4x4 sub-block in SplatBatchT::m16 (8x8 row-major lanes)

  int32_t start = isect_offsets[tile];
  int32_t end = (tile == total_tiles - 1) ? (int32_t)I : isect_offsets[tile + 1];
  if (masks != nullptr && !masks[tile]) end = start;  // masked tile: background only

  float T = 1.0f;
  float cr = 0.f, cg = 0.f, cb = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, dsum = 0.f, med = 0.f;
  int32_t cur = 0, med_idx = 0;
  bool done = !inside;
  int g_mine = -1;
  uint8_t *trow = nullptr;
  if (TRACE && inside && trace_rows[pid] >= 0) trow = trace_bits + (int64_t)trace_rows[pid] * trace_stride;

  const int nb = (end - start + RT - 1) / RT;
  for (int b = 0; b < nb; ++b) {
    // barrier A: every wave has finished reading the previous batch
    const int all_done = __syncthreads_and(done ? 1 : 0);
    if (g_mine >= 0) {
      const unsigned v = max(max(lds.vis[0][tid], lds.vis[1][tid]), max(lds.vis[2][tid], lds.vis[3][tid]));
      if (v) atomicMax(visibilities + g_mine, v);
      g_mine = -1;
    }
    if (all_done) break;
    const int32_t bstart = start + b * RT;
    const int32_t idx = bstart + tid;
    if (idx < end) {
      g_mine = flatten_ids[idx];
      stage_splat(lds.s, tid, g_mine, means2d, ray_transforms, colors, opacities, normals, (float)(tx * TILE),
                  (float)(ty * TILE));
      lds.vis[0][tid] = 0u; lds.vis[1][tid] = 0u; lds.vis[2][tid] = 0u; lds.vis[3][tid] = 0u;
      if (COUNT) {
        unsigned a4, a8;
        const float2 xy = *reinterpret_cast<const float2 *>(means2d + 2 * (int64_t)g_mine);
        subblock_masks(ray_transforms + 9 * (int64_t)g_mine, xy.x, xy.y, opacities[g_mine], (float)(tx * TILE), (float)(ty * TILE), a4, a8);
        lds.dbg4x4[tid] = (unsigned short)a4; lds.dbg8x2[tid] = (unsigned short)a8;
      }
    }
    __syncthreads();  // barrier B
    const int count = min(RT, end - bstart);
    if (__ballot(!done) == 0ull) continue;  // this wave's 64 pixels are all finished
    // per-wave compaction: only the splats whose conservative box reaches this wave's quadrant are evaluated
    for (int c0 = 0; c0 < count; c0 += 64) {
      const int ti = c0 + lane;
      unsigned long long todo = __ballot(ti < count && ((lds.s.m16[ti < RT ? ti : 0] >> (4 * wave)) & 0xFu));
      if (COUNT) {   // iterations this chunk would take if each 16-lane row of the wave followed its own list
        int r4 = 0, r8 = 0;
        for (int sb = 0; sb < 4; ++sb) {
          r4 = max(r4, (int)__popcll(__ballot(ti < count && ((lds.dbg4x4[ti < RT ? ti : 0] >> (4 * wave + sb)) & 1u))));
          r8 = max(r8, (int)__popcll(__ballot(ti < count && ((lds.dbg8x2[ti < RT ? ti : 0] >> (4 * wave + sb)) & 1u))));
        }
        c_r4 += r4; c_r8 += r8; c_union += __popcll(todo);
      }
      while (todo) {
        const int t = c0 + __builtin_ctzll(todo);  // front-to-back
        todo &= todo - 1ull;
        const float4 a0 = lds.s.q0[t], a1 = lds.s.q1[t], a2 = lds.s.q2[t], a3 = lds.s.q3[t];
        PairEval e;
        eval_pair(lx, ly, px, py, a0, a1, a2, a3.x, a3.y, e);
        bool valid = !done && e.ok && ((lds.s.m16[t] >> sub_bit) & 1u);   // the pixel's own 4x4 sub-block is reached (the mask the backward takes its lists from)
        if (COUNT) { c_visit += 1; c_live += __popcll(__ballot(!done)); c_ok += __popcll(__ballot(valid)); c_empty += __ballot(valid) == 0ull; }
        if (__ballot(valid) == 0ull) continue;
        const float nT = T * (1.0f - e.alpha);
        if (TRACE && trow != nullptr && valid) {
          const int k = bstart + t - start;
          if (k < trace_stride)
            trow[k] = (uint8_t)((e.b3 ? 2 : 0) | (e.clamped ? 4 : 0) | (nT <= T_EPS ? 8 : (1 | (T > 0.5f ? 16 : 0))));
        }
        if (valid && nT <= T_EPS) {  // this pixel is finished: exclusive (the splat is not blended)
          done = true;
          valid = false;
        }
        if (COUNT) c_blend += __popcll(__ballot(valid));
        const float w = valid ? e.alpha * T : 0.0f;
        const float4 a4 = lds.s.q4[t];
        cr += a3.z * w; cg += a3.w * w; cb += a4.x * w;
        nx += a4.y * w; ny += a4.z * w; nz += a4.w * w;
        dsum += e.dep * w;
        if (valid) {
          if (T > 0.5f) { med = e.dep; med_idx = bstart + t; }
          cur = bstart + t;
          T = nT;
        }
        const unsigned wmax = wave_umax_to_lane63(__float_as_uint(w));  // w >= 0: uint order == float order
        if (lane == 63) lds.vis[wave][t] = wmax;
      }
      if (__ballot(!done) == 0ull) break;
    }
  }
  __syncthreads();
  if (COUNT && lane == 0) {
    atomicAdd(counters + 0, c_visit); atomicAdd(counters + 1, c_live); atomicAdd(counters + 2, c_ok); atomicAdd(counters + 3, c_blend);
    atomicAdd(counters + 7, c_empty); atomicAdd(counters + 8, c_union); atomicAdd(counters + 9, c_r4); atomicAdd(counters + 10, c_r8);
  }
  if (g_mine >= 0) {
    const unsigned v = max(max(lds.vis[0][tid], lds.vis[1][tid]), max(lds.vis[2][tid], lds.vis[3][tid]));
    if (v) atomicMax(visibilities + g_mine, v);
  }
  if (inside) {
    float br = 0.f, bg = 0.f, bb = 0.f;
    if (backgrounds != nullptr) { br = backgrounds[3 * cam]; bg = backgrounds[3 * cam + 1]; bb = backgrounds[3 * cam + 2]; }
    render_colors[3 * pid] = cr + T * br;
    render_colors[3 * pid + 1] = cg + T * bg;
    render_colors[3 * pid + 2] = cb + T * bb;
    render_normals[3 * pid] = nx; render_normals[3 * pid + 1] = ny; render_normals[3 * pid + 2] = nz;
    render_depths[pid] = dsum;
    render_alphas[pid] = 1.0f - T;
    if (final_T != nullptr) final_T[pid] = T;
    render_median[pid] = med;
    last_ids[pid] = cur;
    median_ids[pid] = med_idx;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 4: ROW LISTS.  The kernel above evaluates a staged splat for all 64 pixels of a wave's 8x8 quadrant although its footprint covers
// ~14 of them (profiles/r03_raster_pair_counters_cfg3.json: 16.6 % of the evaluated lanes blend).  Here every 16-lane DPP row of a wave owns
// one 4x4-pixel sub-block and follows ITS OWN list of the staged splats whose conservative box reaches those 16 pixels (subblock_mask4x4,
// evaluated once per (tile, splat) by the staging lane): per batch the wave compacts four lists into LDS with ballot + mbcnt (24 VALU per 64
// staged splats), then iterates k = 0 .. max list length; lane l reads slot list[row(l)][k], so one iteration blends up to FOUR different
// splats.  Each pixel still sees its tile's splats in list order, so every output is bit-identical to the quadrant kernel's.  The maximum
// blending weight per splat goes to LDS with one ds_max_u32 per row and iteration (integer LDS atomics are cheap).
// ---------------------------------------------------------------------------------------------------------------------------------------
struct FwdRowsLds {
  SplatBatch s;
  unsigned vis[RT];              // max blending weight over the tile's pixels (fp32 bits)
  unsigned char list[16][RT];    // per (wave, row): slots of the staged splats that reach the row, in list order
};

// workgroups per CU the register allocation aims for (26 KB of LDS each: six fit).  Round 5: the kernel happened to compile to 96 or 100
// registers (5 or 4 waves per SIMD) depending on unrelated edits of the staging code, 0.385 against 0.426 ms at cfg3: the occupancy is now stated
#ifndef RASTER_FWD_ROWS_WGS
#define RASTER_FWD_ROWS_WGS 5
#endif
template <bool COUNT, bool TRACE>
__global__ void __launch_bounds__(RT, RASTER_FWD_ROWS_WGS)
    raster_fwd_rows_kernel(int n_xcd, int64_t total_tiles, int64_t n_tiles, int64_t I, int W, int H, int tw,
                           const float *__restrict__ means2d, const float *__restrict__ ray_transforms,
                           const float *__restrict__ colors, const float *__restrict__ opacities,
                           const float *__restrict__ normals, const float *__restrict__ backgrounds,
                           const uint8_t *__restrict__ masks, const int32_t *__restrict__ isect_offsets,
                           const int32_t *__restrict__ flatten_ids, float *__restrict__ render_colors,
                           float *__restrict__ render_depths, float *__restrict__ render_alphas,
                           float *__restrict__ render_normals, float *__restrict__ render_median,
                           int32_t *__restrict__ last_ids, int32_t *__restrict__ median_ids,
                           unsigned *__restrict__ visibilities, float *__restrict__ final_T, unsigned long long *__restrict__ counters,
                           const int32_t *__restrict__ trace_rows, int trace_stride, uint8_t *__restrict__ trace_bits) {
  __shared__ FwdRowsLds lds;
  unsigned long long c_visit = 0, c_live = 0, c_ok = 0, c_blend = 0, c_empty = 0;
  const int64_t tile = xcd_tile_index(total_tiles, n_xcd);
  if (tile >= total_tiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4;
  const int64_t cam = tile / n_tiles;
  const int tl = (int)(tile - cam * n_tiles);
  const int ty = tl / tw, tx = tl - ty * tw;
  int plx, ply;
  row_pixel(wave, lane, plx, ply);
  const int x = tx * TILE + plx, y = ty * TILE + ply;
  const bool inside = x < W && y < H;
  const int64_t pid = (cam * H + y) * (int64_t)W + x;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;

  int32_t start = isect_offsets[tile];
  int32_t end = (tile == total_tiles - 1) ? (int32_t)I : isect_offsets[tile + 1];
  if (masks != nullptr && !masks[tile]) end = start;

  float T = 1.0f;
  float cr = 0.f, cg = 0.f, cb = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, dsum = 0.f, med = 0.f;
  int32_t cur = 0, med_idx = 0;
  bool done = !inside;
  int g_mine = -1;
  uint8_t *trow = nullptr;
  if (TRACE && inside && trace_rows[pid] >= 0) trow = trace_bits + (int64_t)trace_rows[pid] * trace_stride;
  unsigned char *my_list = lds.list[wave * 4 + row];

  const int nb = (end - start + RT - 1) / RT;
  for (int b = 0; b < nb; ++b) {
    const int all_done = __syncthreads_and(done ? 1 : 0);   // barrier A: every wave has finished reading the previous batch
    if (g_mine >= 0) {
      const unsigned v = lds.vis[tid];
      if (v) atomicMax(visibilities + g_mine, v);
      g_mine = -1;
    }
    if (all_done) break;
    const int32_t bstart = start + b * RT;
    const int32_t idx = bstart + tid;
    if (idx < end) {
      g_mine = flatten_ids[idx];
      stage_splat(lds.s, tid, g_mine, means2d, ray_transforms, colors, opacities, normals, (float)(tx * TILE), (float)(ty * TILE));
      lds.vis[tid] = 0u;
    }
    __syncthreads();  // barrier B
    const int count = min(RT, end - bstart);
    const unsigned long long live = __ballot(!done);
    if (live == 0ull) continue;  // this wave's 64 pixels are all finished
    // ---- the four row lists of this wave (rows whose 16 pixels are all finished get an empty list)
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    for (int c0 = 0; c0 < count; c0 += 64) {
      const int ti = c0 + lane;
      const unsigned m = ti < count ? (unsigned)lds.s.m16[ti] >> (4 * wave) : 0u;
#define ROW_LIST(r, n)                                                                                         \
  {                                                                                                            \
    const bool bit = (m >> r) & 1u;                                                                            \
    const unsigned long long mk = __ballot(bit);                                                               \
    if (bit) lds.list[wave * 4 + r][n + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] = (unsigned char)ti; \
    n += (int)__popcll(mk);                                                                                    \
  }
      ROW_LIST(0, n0) ROW_LIST(1, n1) ROW_LIST(2, n2) ROW_LIST(3, n3)
#undef ROW_LIST
    }
    if (((live >> 0) & 0xFFFFull) == 0ull) n0 = 0;
    if (((live >> 16) & 0xFFFFull) == 0ull) n1 = 0;
    if (((live >> 32) & 0xFFFFull) == 0ull) n2 = 0;
    if (((live >> 48) & 0xFFFFull) == 0ull) n3 = 0;
    const int n_mine = row == 0 ? n0 : (row == 1 ? n1 : (row == 2 ? n2 : n3));
    const int kmax = max(max(n0, n1), max(n2, n3));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the lists are wave-private: written and read by this wave only
    for (int k = 0; k < kmax; ++k) {
      const bool active = k < n_mine;
      const int t = active ? (int)my_list[k] : 0;
      const float4 a0 = lds.s.q0[t], a1 = lds.s.q1[t], a2 = lds.s.q2[t], a3 = lds.s.q3[t];
      PairEval e;
      eval_pair(0.f, 0.f, px, py, a0, a1, a2, a3.x, a3.y, e);
      bool valid = active && !done && e.ok;
      if (COUNT) { c_visit += 1; c_live += __popcll(__ballot(active && !done)); c_ok += __popcll(__ballot(valid)); c_empty += __ballot(valid) == 0ull; }
      if (__ballot(valid) == 0ull) continue;
      const float nT = T * (1.0f - e.alpha);
      if (TRACE && trow != nullptr && valid) {
        const int kk = bstart + t - start;
        if (kk < trace_stride)
          trow[kk] = (uint8_t)((e.b3 ? 2 : 0) | (e.clamped ? 4 : 0) | (nT <= T_EPS ? 8 : (1 | (T > 0.5f ? 16 : 0))));
      }
      if (valid && nT <= T_EPS) {  // this pixel is finished: exclusive (the splat is not blended)
        done = true;
        valid = false;
      }
      if (COUNT) c_blend += __popcll(__ballot(valid));
      const float w = valid ? e.alpha * T : 0.0f;
      const float4 a4 = lds.s.q4[t];
      cr += a3.z * w; cg += a3.w * w; cb += a4.x * w;
      nx += a4.y * w; ny += a4.z * w; nz += a4.w * w;
      dsum += e.dep * w;
      if (valid) {
        if (T > 0.5f) { med = e.dep; med_idx = bstart + t; }
        cur = bstart + t;
        T = nT;
      }
      const unsigned wmax = row_umax_to_lane15(__float_as_uint(w));  // w >= 0: uint order == float order
      if ((lane & 15) == 15 && wmax) atomicMax(&lds.vis[t], wmax);
      if ((k & 15) == 15 && __ballot(!done) == 0ull) break;
    }
  }
  __syncthreads();
  if (COUNT && lane == 0) {
    atomicAdd(counters + 0, c_visit); atomicAdd(counters + 1, c_live); atomicAdd(counters + 2, c_ok); atomicAdd(counters + 3, c_blend);
    atomicAdd(counters + 7, c_empty);
  }
  if (g_mine >= 0) {
    const unsigned v = lds.vis[tid];
    if (v) atomicMax(visibilities + g_mine, v);
  }
  if (inside) {
    float br = 0.f, bg = 0.f, bb = 0.f;
    if (backgrounds != nullptr) { br = backgrounds[3 * cam]; bg = backgrounds[3 * cam + 1]; bb = backgrounds[3 * cam + 2]; }
    render_colors[3 * pid] = cr + T * br;
    render_colors[3 * pid + 1] = cg + T * bg;
    render_colors[3 * pid + 2] = cb + T * bb;
    render_normals[3 * pid] = nx; render_normals[3 * pid + 1] = ny; render_normals[3 * pid + 2] = nz;
    render_depths[pid] = dsum;
    render_alphas[pid] = 1.0f - T;
    if (final_T != nullptr) final_T[pid] = T;
    render_median[pid] = med;
    last_ids[pid] = cur;
    median_ids[pid] = med_idx;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: QUAD LISTS (raster_quad.h).  Every lane quad (2x2 pixels) follows its own list of the staged splats whose 64-bit reach mask has the
// quad's bit; the staging copies 80 bytes of the packed record + the mask.  Each pixel still sees its tile's splats in list order and evaluates
// them with the same operands, so every output is bit-identical to the row-list and quadrant kernels above.  Quads whose four pixels are
// finished get no list (termination mask).
// ---------------------------------------------------------------------------------------------------------------------------------------
// Measured at cfg3 (tools/exp_raster_quads.py, library variants of tools/build_variants.sh; pack + mask + forward, ms): batch 192 at 5 workgroups
// per CU 0.343, 160 at 6 0.313, 144 at 6 0.302, 128 at 7 0.299, 112 at 8 0.291, 240 at 4 0.330: the kernel is latency-bound (VALU issue 49 %, LDS
// 26 % busy at 5 per CU), so the batch is what eight workgroups' LDS allows (62 registers: eight waves per SIMD fit).
#ifndef RASTER_FWD_QUADS_BATCH
#define RASTER_FWD_QUADS_BATCH 120
#endif
#ifndef RASTER_FWD_QUADS_WGS
#define RASTER_FWD_QUADS_WGS 8
#endif
static constexpr int FQB = RASTER_FWD_QUADS_BATCH;
static constexpr int FQ_CHUNKS = (FQB + 63) / 64;
struct FwdQuadsLds {
  float4 q0[FQB], q1[FQB], q2[FQB], q3[FQB], q4[FQB];
  unsigned long long m64[FQB];
  unsigned vis[FQB];               // max blending weight over the tile's pixels (fp32 bits)
  unsigned char list[64][FQB];     // per (wave, quad): slots of the staged splats that reach the quad, in list order
  unsigned short cmask[4][FQB];    // per wave: quad bits of the staged splats that reach the wave's quadrant (compacted, list order)
  unsigned char cslot[4][FQB];     //           and their slots
};
static_assert(sizeof(FwdQuadsLds) * RASTER_FWD_QUADS_WGS <= 160 * 1024, "raster_fwd_quads: LDS per workgroup against the stated workgroups per CU");
static_assert(FQB % 4 == 0 && FQB <= 256, "raster_fwd_quads: list words, byte slots");

template <bool COUNT, bool TRACE>
__global__ void __launch_bounds__(RT, RASTER_FWD_QUADS_WGS)
    raster_fwd_quads_kernel(int n_xcd, int64_t total_tiles, int64_t n_tiles, int64_t I, int W, int H, int tw,
                            const float4 *__restrict__ rec, const unsigned long long *__restrict__ pair_masks,
                            const float *__restrict__ backgrounds, const uint8_t *__restrict__ masks,
                            const int32_t *__restrict__ isect_offsets, const int32_t *__restrict__ flatten_ids,
                            float *__restrict__ render_colors, float *__restrict__ render_depths, float *__restrict__ render_alphas,
                            float *__restrict__ render_normals, float *__restrict__ render_median, int32_t *__restrict__ last_ids,
                            int32_t *__restrict__ median_ids, unsigned *__restrict__ visibilities, float *__restrict__ final_T,
                            unsigned long long *__restrict__ counters, const int32_t *__restrict__ trace_rows, int trace_stride,
                            uint8_t *__restrict__ trace_bits) {
  __shared__ FwdQuadsLds lds;
  unsigned long long c_visit = 0, c_live = 0, c_ok = 0, c_blend = 0, c_empty = 0;
  const int64_t tile = xcd_tile_index(total_tiles, n_xcd);
  if (tile >= total_tiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t cam = tile / n_tiles;
  const int tl = (int)(tile - cam * n_tiles);
  const int ty = tl / tw, tx = tl - ty * tw;
  int plx, ply;
  quad_pixel(wave, lane, plx, ply);
  const int x = tx * TILE + plx, y = ty * TILE + ply;
  const bool inside = x < W && y < H;
  const int64_t pid = (cam * H + y) * (int64_t)W + x;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;

  int32_t start = isect_offsets[tile];
  int32_t end = (tile == total_tiles - 1) ? (int32_t)I : isect_offsets[tile + 1];
  if (masks != nullptr && !masks[tile]) end = start;

  float T = 1.0f;
  float cr = 0.f, cg = 0.f, cb = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, dsum = 0.f, med = 0.f;
  int32_t cur = 0, med_idx = 0;
  bool done = !inside;
  int g_mine = -1;
  uint8_t *trow = nullptr;
  if (TRACE && inside && trace_rows[pid] >= 0) trow = trace_bits + (int64_t)trace_rows[pid] * trace_stride;
  const unsigned char *my_list = lds.list[wave * 16 + (lane >> 2)];
  const int slot = thread_slot(tid);
  const int mbase = wave_mask_base(wave);

  const int nb = (end - start + FQB - 1) / FQB;
  for (int b = 0; b < nb; ++b) {
    const int all_done = __syncthreads_and(done ? 1 : 0);   // barrier A: every wave has finished reading the previous batch
    if (g_mine >= 0) {
      const unsigned v = lds.vis[slot];
      if (v) atomicMax(visibilities + g_mine, v);
      g_mine = -1;
    }
    if (all_done) break;
    const int32_t bstart = start + b * FQB;
    const int32_t idx = bstart + slot;
    if (slot < FQB && idx < end) {
      g_mine = flatten_ids[idx];
      const float4 *r = rec + 8 * (int64_t)g_mine;
      lds.q0[slot] = r[0]; lds.q1[slot] = r[1]; lds.q2[slot] = r[2]; lds.q3[slot] = r[3]; lds.q4[slot] = r[4];
      lds.m64[slot] = pair_masks[idx];
      lds.vis[slot] = 0u;
    }
    __syncthreads();  // barrier B
    const int count = min(FQB, end - bstart);
    const unsigned long long live = __ballot(!done);
    if (live == 0ull) continue;  // this wave's 64 pixels are all finished
    const unsigned live_q = quads_any(live);
    // ---- step 1: the staged splats that reach a LIVE quad of this wave, compacted in list order
    int ncomp = 0;
#pragma unroll
    for (int c = 0; c < FQ_CHUNKS; ++c) {
      if (64 * c < count) {
        const int ti = 64 * c + lane;
        const unsigned qb = ti < count ? (wave_quad_bits(lds.m64[ti], mbase) & live_q) : 0u;
        const bool any = qb != 0u;
        const unsigned long long mk = __ballot(any);
        if (any) {
          const int pos = ncomp + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
          lds.cslot[wave][pos] = (unsigned char)ti;
          lds.cmask[wave][pos] = (unsigned short)qb;
        }
        ncomp += (int)__popcll(mk);
      }
    }
    if (ncomp == 0) continue;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // wave-private: written and read by this wave only
    // ---- step 2: one ballot per quad and compacted chunk
    unsigned cm[FQ_CHUNKS];
    int cti[FQ_CHUNKS];
#pragma unroll
    for (int c = 0; c < FQ_CHUNKS; ++c) {
      const int i = 64 * c + lane;
      const bool in = 64 * c < ncomp && i < ncomp;
      cm[c] = in ? (unsigned)lds.cmask[wave][i] : 0u;
      cti[c] = in ? (int)lds.cslot[wave][i] : 0;
    }
    int nvec = 0, kmax = 0;
    for_quads([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      int n = 0;
#pragma unroll
      for (int c = 0; c < FQ_CHUNKS; ++c) {
        if (64 * c < ncomp) {
          const bool bit = (cm[c] >> Q) & 1u;
          const unsigned long long mk = __ballot(bit);
          if (bit) lds.list[wave * 16 + Q][n + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] = (unsigned char)cti[c];
          n += (int)__popcll(mk);
        }
      }
      writelane<4 * Q>(nvec, n);
      kmax = max(kmax, n);
    });
    const int n_mine = dpp_quad_i<0x00>(nvec);   // quad_perm [0,0,0,0]: the count written to the quad's first lane
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- the compositing loop: every quad walks its own list; the list bytes come four at a time
    unsigned word = 0u;
    for (int k = 0; k < kmax; ++k) {
      if ((k & 3) == 0) word = *reinterpret_cast<const unsigned *>(my_list + k);   // lists start 4-byte aligned (FQB % 4 == 0); reads past n_mine are unused
      const bool active = k < n_mine;
      const int t = active ? (int)((word >> (8 * (k & 3))) & 0xFFu) : 0;
      const float4 a0 = lds.q0[t], a1 = lds.q1[t], a2 = lds.q2[t], a3 = lds.q3[t], a4 = lds.q4[t];
      PairEval e;
      eval_pair(0.f, 0.f, px, py, a0, a1, a2, a3.x, a3.y, e);
      bool valid = active && !done && e.ok;
      if (COUNT) { c_visit += 1; c_live += __popcll(__ballot(active && !done)); c_ok += __popcll(__ballot(valid)); c_empty += __ballot(valid) == 0ull; }
      const float nT = T * (1.0f - e.alpha);
      if (TRACE && trow != nullptr && valid) {
        const int kk = bstart + t - start;
        if (kk < trace_stride)
          trow[kk] = (uint8_t)((e.b3 ? 2 : 0) | (e.clamped ? 4 : 0) | (nT <= T_EPS ? 8 : (1 | (T > 0.5f ? 16 : 0))));
      }
      if (valid && nT <= T_EPS) {  // this pixel is finished: exclusive (the splat is not blended)
        done = true;
        valid = false;
      }
      if (COUNT) c_blend += __popcll(__ballot(valid));
      const float w = valid ? e.alpha * T : 0.0f;
      cr += a3.z * w; cg += a3.w * w; cb += a4.x * w;
      nx += a4.y * w; ny += a4.z * w; nz += a4.w * w;
      dsum += e.dep * w;
      if (valid) {
        if (T > 0.5f) { med = e.dep; med_idx = bstart + t; }
        cur = bstart + t;
        T = nT;
      }
      const unsigned wmax = quad_umax(__float_as_uint(w));  // w >= 0: uint order == float order
      if ((lane & 3) == 0 && wmax) atomicMax(&lds.vis[t], wmax);
      if ((k & 7) == 7 && __ballot(!done) == 0ull) break;
    }
  }
  __syncthreads();
  if (COUNT && lane == 0) {
    atomicAdd(counters + 0, c_visit); atomicAdd(counters + 1, c_live); atomicAdd(counters + 2, c_ok); atomicAdd(counters + 3, c_blend);
    atomicAdd(counters + 7, c_empty);
  }
  if (g_mine >= 0) {
    const unsigned v = lds.vis[slot];
    if (v) atomicMax(visibilities + g_mine, v);
  }
  if (inside) {
    float br = 0.f, bg = 0.f, bb = 0.f;
    if (backgrounds != nullptr) { br = backgrounds[3 * cam]; bg = backgrounds[3 * cam + 1]; bb = backgrounds[3 * cam + 2]; }
    render_colors[3 * pid] = cr + T * br;
    render_colors[3 * pid + 1] = cg + T * bg;
    render_colors[3 * pid + 2] = cb + T * bb;
    render_normals[3 * pid] = nx; render_normals[3 * pid + 1] = ny; render_normals[3 * pid + 2] = nz;
    render_depths[pid] = dsum;
    render_alphas[pid] = 1.0f - T;
    if (final_T != nullptr) final_T[pid] = T;
    render_median[pid] = med;
    last_ids[pid] = cur;
    median_ids[pid] = med_idx;
  }
}

}  // namespace gsdf

using namespace gsdf;

static int rasterize_fwd_launch(int64_t C, int64_t M, int64_t I, int width, int height, int tile_size,
                                const float *means2d, const float *ray_transforms, const float *colors,
                                const float *opacities, const float *normals, const float *backgrounds,
                                const uint8_t *masks, const int32_t *isect_offsets,
                                const int32_t *flatten_ids, float *render_colors, float *render_depths,
                                float *render_alphas, float *render_normals, float *render_median,
                                int32_t *last_ids, int32_t *median_ids, float *visibilities, float *final_T, void *ws,
                                const gsdf_raster_instr *instr, hipStream_t stream) {
  GSDF_REQUIRE(tile_size == TILE, "rasterize_fwd: tile_size %d unsupported (16 only)", tile_size);
  GSDF_REQUIRE(width > 0 && height > 0 && C >= 1, "rasterize_fwd: bad geometry");
  GSDF_REQUIRE(render_colors && render_depths && render_alphas && render_normals && render_median && last_ids &&
                   median_ids && isect_offsets,
               "rasterize_fwd: null output/offsets");
  GSDF_REQUIRE(I == 0 || (flatten_ids && means2d && ray_transforms && colors && opacities && normals),
               "rasterize_fwd: null input");
  GSDF_REQUIRE(M == 0 || visibilities, "rasterize_fwd: null visibilities");
  unsigned long long *counters = instr ? instr->counters : nullptr;
  const int32_t *trace_rows = instr ? instr->trace_rows : nullptr;
  const int trace_stride = instr ? instr->trace_stride : 0;
  uint8_t *trace_bits = instr ? instr->trace_bits : nullptr;
  GSDF_REQUIRE(!(counters && trace_rows), "rasterize_fwd: counters and the decision trace are separate instrumented launches");
  GSDF_REQUIRE(!trace_rows || (trace_bits && trace_stride > 0), "rasterize_fwd: trace_rows without trace_bits / stride");
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE;
  const int64_t n_tiles = (int64_t)tw * th, total = n_tiles * C;
  if (M > 0) GSDF_HIP(hipMemsetAsync(visibilities, 0, (size_t)M * 4, stream), "rasterize_fwd memset");
  const int n_xcd = xcd_count(stream);
#define FWD_ARGS n_xcd, total, n_tiles, I, width, height, tw, means2d, ray_transforms, colors, opacities, normals, backgrounds, masks, isect_offsets, \
                 flatten_ids, render_colors, render_depths, render_alphas, render_normals, render_median, last_ids, median_ids,          \
                 (unsigned *)visibilities, final_T, counters, trace_rows, trace_stride, trace_bits
  static const int lists_mode = raster_lists_mode();
  if (lists_mode == 0) {   // quad lists (round 6): pack + mask passes, then the compositing kernel
    GSDF_REQUIRE(ws != nullptr || I == 0, "rasterize_fwd: null workspace (gsdf_rasterize_2dgs_fwd_ws_bytes)");
    if (I > 0) {
      const int rc = raster_pack_launch(M, I, total, n_tiles, tw, means2d, ray_transforms, colors, opacities, normals, isect_offsets, flatten_ids, ws, stream);
      if (rc != GSDF_OK) return rc;
    }
#define FWDQ_ARGS n_xcd, total, n_tiles, I, width, height, tw, (const float4 *)ws_records(ws), ws_masks(ws, M), backgrounds, masks, isect_offsets, \
                  flatten_ids, render_colors, render_depths, render_alphas, render_normals, render_median, last_ids, median_ids,          \
                  (unsigned *)visibilities, final_T, counters, trace_rows, trace_stride, trace_bits
    if (counters != nullptr) raster_fwd_quads_kernel<true, false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWDQ_ARGS);
    else if (trace_rows != nullptr) raster_fwd_quads_kernel<false, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWDQ_ARGS);
    else raster_fwd_quads_kernel<false, false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWDQ_ARGS);
#undef FWDQ_ARGS
  } else if (lists_mode == 2) {
    if (counters != nullptr) raster_fwd_kernel<true, false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWD_ARGS);
    else if (trace_rows != nullptr) raster_fwd_kernel<false, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWD_ARGS);
    else raster_fwd_kernel<false, false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWD_ARGS);
  } else {
    if (counters != nullptr) raster_fwd_rows_kernel<true, false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWD_ARGS);
    else if (trace_rows != nullptr) raster_fwd_rows_kernel<false, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWD_ARGS);
    else raster_fwd_rows_kernel<false, false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(FWD_ARGS);
  }
#undef FWD_ARGS
  GSDF_CHECK_LAUNCH("raster_fwd_kernel");
  return GSDF_OK;
}

extern "C" size_t gsdf_rasterize_2dgs_fwd_ws_bytes(int64_t M, int64_t I) { return raster_pack_bytes(M, I); }

extern "C" int gsdf_rasterize_2dgs_fwd(int64_t C, int64_t M, int64_t I, int width, int height, int tile_size,
                                       const float *means2d, const float *ray_transforms, const float *colors,
                                       const float *opacities, const float *normals, const float *backgrounds,
                                       const uint8_t *masks, const int32_t *isect_offsets,
                                       const int32_t *flatten_ids, float *render_colors, float *render_depths,
                                       float *render_alphas, float *render_normals, float *render_median,
                                       int32_t *last_ids, int32_t *median_ids, float *visibilities, float *final_T, void *ws,
                                       gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_rasterize_2dgs_fwd");
  return rasterize_fwd_launch(C, M, I, width, height, tile_size, means2d, ray_transforms, colors, opacities, normals, backgrounds, masks,
                              isect_offsets, flatten_ids, render_colors, render_depths, render_alphas, render_normals, render_median,
                              last_ids, median_ids, visibilities, final_T, ws, nullptr, (hipStream_t)stream_);
}

extern "C" int gsdf_rasterize_2dgs_fwd_instr(int64_t C, int64_t M, int64_t I, int width, int height, int tile_size,
                                             const float *means2d, const float *ray_transforms, const float *colors,
                                             const float *opacities, const float *normals, const float *backgrounds,
                                             const uint8_t *masks, const int32_t *isect_offsets,
                                             const int32_t *flatten_ids, float *render_colors, float *render_depths,
                                             float *render_alphas, float *render_normals, float *render_median,
                                             int32_t *last_ids, int32_t *median_ids, float *visibilities, float *final_T, void *ws,
                                             const gsdf_raster_instr *instr, gsdf_stream_t stream_) {
  return rasterize_fwd_launch(C, M, I, width, height, tile_size, means2d, ray_transforms, colors, opacities, normals, backgrounds, masks,
                              isect_offsets, flatten_ids, render_colors, render_depths, render_alphas, render_normals, render_median,
                              last_ids, median_ids, visibilities, final_T, ws, instr, (hipStream_t)stream_);
}
