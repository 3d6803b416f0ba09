// raster_bwd.hip — P4': VJP of the 2DGS compositing (SPEC A.5).
// Reference: implicit autograd of rasterize_to_pixels_2dgs
// (/root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223); the gradients of the leaf
// tensors `densify` / `means2d_absgrad` are consumed at neural_gaussian.cpp:626-633.
//
// Per tile: replay the depth-sorted list back-to-front from each pixel's last contributor; the per-(pixel, splat) gradient
// terms are reduced over the lanes of a quad with DPP (no LDS traffic) and added to a per-tile LDS record of the splat.
//
// Flush to HBM — measured on MI355X (tools/ubench/atomic_*.hip): the fp32 atomic path retires
// ~21 G *64-byte-line requests*/s chip-wide, independent of footprint and scope, and lanes of one
// instruction that hit the same line are merged (16 consecutive floats -> 307 G atomics/s).  One
// atomic per (tile, splat, field) into six separate [M,k] arrays is 17 line requests per
// intersection (4e7 per view = 1.9 ms of the first kernel's 2.7 ms).  So the accumulator of one splat is a
// single 84-byte RECORD ([M][21] floats), three records are flushed per wave instruction with the
// 21 fields in consecutive lanes (<= 2 lines per splat), and a streaming epilogue unpacks the
// records into the operator's six gradient tensors (+ the densification signal).
#include <type_traits>

#include "raster_quad.h"

namespace gsdf {

static constexpr int NACC = 21;
// Per-splat gradient record.  The cross-product chain of z = h_u x h_v is NOT differentiated per pixel: the record
// accumulates the moments of v_z about the splat's own centre,
//     V0 = sum v_z,   Vx = sum (p_x - mean2d.x) v_z,   Vy = sum (p_y - mean2d.y) v_z,
// (z is affine in the pixel, so these nine numbers carry everything); the depth dep = s . M_w.xy + M_w.z reaches M_w
// directly, (sum v_dep s.x, sum v_dep s.y, sum v_dep), and s through v_z.  The streaming epilogue turns the record into
// dL/dM_u, dL/dM_v, dL/dM_w once per splat.
// slots: 0-2 v_rgb, 3-5 v_normal, 6 v_opacity, 7-9 V0, 10-12 Vx, 13-15 Vy, 16-18 direct dL/dM_w of the depth (both branches in
//        18), 19-20 v_means2d ; v_means2d_abs lives in a second record array (only with absgrad)

__device__ __forceinline__ void lds_add(float *p, float v) { atomicAdd(p, v); }

// ---------------------------------------------------------------------------------------------------------------------------------------
// QUAD LISTS (round 6, raster_quad.h).  Every lane quad (2x2 pixels) replays its own list of the staged splats (those whose 64-bit reach mask
// has the quad's bit and that lie at or before the quad's last contributor), back to front.  The per-pixel gradient terms are reduced over the
// FOUR lanes of the quad with a transposing butterfly (16 values -> 4 per lane: 36 VALU) and added to the splat's record in LDS.
// The record is accumulated in DOUBLE: ds_add_f32 retires 0.33 lanes per clock per CU on gfx950 whatever the address pattern
// (tools/ubench/lds_fadd_patterns.hip: linear, the row kernel's 4 records x 16 fields, 16 records x 4 fields, random — all 0.20 T lane-ops/s
// chip-wide), ds_add_f64 7.7 (4.7 T/s): the row kernel's 20 float adds per (row, splat) were ~0.4 ms of LDS-atomic time per launch, the quad
// kernel's 20 per (quad, splat) would be ~1.2 ms; as double adds they are ~0.05 ms, and the tile-level sum is exact to fp32.
// ---------------------------------------------------------------------------------------------------------------------------------------
// Measured at cfg3 (tools/exp_raster_quads.py, library variants of tools/build_variants.sh; backward + unpack, ms): batch 116 at 4 workgroups per CU
// 0.750; without the SLP vectoriser (its packed-fp32 forms cost more moves than they save: 106 -> 96 registers) 0.731; v_rcp_f32 for 1 / (1 - alpha)
// instead of the IEEE division sequence 0.719; batch 92 at 5 workgroups per CU 0.692 (84: 0.764 — a smaller batch at the same occupancy costs more
// than it gives); batch 76 at 6 per CU (80 registers, 27-32 spilled) 0.911; software-pipelined operand reads 0.713 (10 more spills) / 0.721 at 4 per CU.
#ifndef RASTER_BWD_RCP
#define RASTER_BWD_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
#ifndef RASTER_BWD_QUADS_BATCH
#define RASTER_BWD_QUADS_BATCH 92
#endif
#ifndef RASTER_BWD_QUADS_WGS
#define RASTER_BWD_QUADS_WGS 5
#endif
static constexpr int BQB = RASTER_BWD_QUADS_BATCH;
static constexpr int BQ_CHUNKS = (BQB + 63) / 64;
static constexpr int NACC_D = 20;   // double fields of a record: 0-18 as NACC's 0-18, 19 padding (the 4-value butterfly's fourth output)
// DETERMINISTIC (gsdf_deterministic, round 6): every accumulator is a 64-bit FIXED-POINT integer — the LDS records (ds_add_u64 instead of
// ds_add_f64), the global records (global_atomic_add_x2 instead of the float atomics) — so that neither the order in which the quads of a tile
// reach a record nor the order in which the tiles reach a splat changes a bit of the result.  The unit is 2^(e - 34) with 2^e the first power of
// two above the largest upstream gradient of the launch (a max-reduction pass: the contributions are linear in the upstream gradients): 5.8e-11
// of that maximum.  A tile sum of 2^13 times the maximum or more cannot be told from an overflow of the 8160-tile total: it poisons the
// launch's outputs with NaN instead (a degenerate splat; the float path would return its huge gradient).
static constexpr int DET_UNIT_SHIFT = 34;
static constexpr long long DET_ADDEND_CAP = 1LL << 54, DET_TILE_CAP = 1LL << 47;
struct DetHeader {
  double to_fix;        // 2^(34 - e)
  double from_fix;      // 2^(e - 34)
  unsigned max_bits;    // bit pattern of the largest |upstream gradient| (atomicMax)
  unsigned poisoned;    // a tile sum out of range, or a non-finite upstream gradient
};
__device__ __forceinline__ long long det_fix(float r, double to_fix) {
  double x = (double)r * to_fix;
  x = fmin(fmax(x, -(double)DET_ADDEND_CAP), (double)DET_ADDEND_CAP);   // (a NaN becomes the cap: poisoned by the tile check)
  return __double2ll_rn(x);
}

template <bool ABSGRAD, bool DET = false>
struct BwdQuadsLds {
  float4 q0[BQB], q1[BQB], q2[BQB], q3[BQB], q4[BQB];   // q3.x = M_w.x (the forward's record has D there)
  float mwy[BQB];
  unsigned long long m64[BQB];
  double acc[BQB][NACC_D];                  // (DET: the same 8 bytes as long long)
  typename std::conditional<DET, long long, float>::type acc2[BQB][2];                       // v_means2d of the low-pass branch (rare): float adds
  typename std::conditional<DET, long long, float>::type acc_abs[ABSGRAD ? BQB : 1][2];
  unsigned char list[64][BQB];
  unsigned short cmask[4][BQB];    // per wave: quad bits of the staged splats that reach the wave's quadrant (compacted, list order)
  unsigned char cslot[4][BQB];     //           and their slots
  int bin_final_max;
};
static_assert(BQB % 4 == 0 && BQB <= 256, "raster_bwd_quads: list words, byte slots");
static_assert(sizeof(BwdQuadsLds<false>) * RASTER_BWD_QUADS_WGS <= 160 * 1024, "raster_bwd_quads: LDS per workgroup against the stated workgroups per CU");
static_assert(sizeof(BwdQuadsLds<true>) * RASTER_BWD_QUADS_WGS <= 160 * 1024, "raster_bwd_quads (absgrad): LDS per workgroup against the stated workgroups per CU");
static constexpr int RASTER_BWD_DET_WGS = 4;
static_assert(sizeof(BwdQuadsLds<true, true>) * RASTER_BWD_DET_WGS <= 160 * 1024, "raster_bwd_quads (deterministic): LDS per workgroup");

// Adds the LDS records of this wave's slots (slot = 4 lane + wave, thread_slot) to the global record array and clears them.
// Three splats per instruction: lane = 21 j + k -> field k of the wave's j-th slot of this round.
template <bool ABSGRAD, bool DET>
__device__ __forceinline__ void flush_records_quads(BwdQuadsLds<ABSGRAD, DET> &lds, int wave, int lane, int g_mine, float *__restrict__ grec,
                                                    float *__restrict__ grec_abs, DetHeader *__restrict__ det) {
  if (ABSGRAD) {
    const int slot = 4 * lane + wave;
    if (g_mine >= 0 && slot < BQB) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (DET) {
          const long long v = (long long)lds.acc_abs[slot][k];
          if (v != 0) {
            lds.acc_abs[slot][k] = 0;
            atomicAdd(reinterpret_cast<unsigned long long *>(grec_abs) + 2 * (int64_t)g_mine + k, (unsigned long long)v);
            if (v >= DET_TILE_CAP || v <= -DET_TILE_CAP) det->poisoned = 1u;
          }
        } else {
          const float v = (float)lds.acc_abs[slot][k];
          if (v != 0.f) { lds.acc_abs[slot][k] = 0; atomicAdd(grec_abs + 2 * (int64_t)g_mine + k, v); }
        }
      }
    }
  }
  const int j = lane / NACC, k = lane - j * NACC;
  constexpr int PER_WAVE = (BQB + 3) / 4;          // staging lanes per wave
#pragma unroll 2
  for (int it = 0; it < (PER_WAVE + 2) / 3; ++it) {
    const int sl = 3 * it + j;                     // staging lane of this wave
    const int g = __shfl(g_mine, sl & 63, 64);
    const int slot = 4 * sl + wave;
    if (j < 3 && sl < PER_WAVE && slot < BQB && g >= 0) {
      if (DET) {
        long long *a = k < 19 ? reinterpret_cast<long long *>(&lds.acc[slot][k]) : reinterpret_cast<long long *>(&lds.acc2[slot][k - 19]);
        const long long v = *a;
        if (v != 0) {
          *a = 0;
          atomicAdd(reinterpret_cast<unsigned long long *>(grec) + (int64_t)g * NACC + k, (unsigned long long)v);
          if (v >= DET_TILE_CAP || v <= -DET_TILE_CAP) det->poisoned = 1u;
        }
      } else {
        float v;
        if (k < 19) {
          double *a = &lds.acc[slot][k];
          const double d = *a;
          v = (float)d;
          if (d != 0.0) *a = 0.0;
        } else {
          float *a = reinterpret_cast<float *>(&lds.acc2[slot][k - 19]);
          v = *a;
          if (v != 0.f) *a = 0.f;
        }
        if (v != 0.f) atomicAdd(grec + (int64_t)g * NACC + k, v);
      }
    }
  }
}

template <bool ABSGRAD, bool COUNT = false, bool DET = false>
__global__ void __launch_bounds__(RT, DET ? RASTER_BWD_DET_WGS : RASTER_BWD_QUADS_WGS)
    raster_bwd_quads_kernel(int n_xcd, int64_t total_tiles, int64_t n_tiles, int64_t I, int W, int H, int tw,
                            const float4 *__restrict__ rec, const unsigned long long *__restrict__ pair_masks,
                            const float *__restrict__ backgrounds, const uint8_t *__restrict__ masks,
                            const int32_t *__restrict__ isect_offsets, const int32_t *__restrict__ flatten_ids,
                            const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                            const int32_t *__restrict__ median_ids, const float *__restrict__ v_render_colors,
                            const float *__restrict__ v_render_depths, const float *__restrict__ v_render_alphas,
                            const float *__restrict__ v_render_normals, const float *__restrict__ v_render_median,
                            float *__restrict__ grec, float *__restrict__ grec_abs, const float *__restrict__ final_T,
                            unsigned long long *__restrict__ counters = nullptr, DetHeader *__restrict__ det = nullptr) {
  __shared__ BwdQuadsLds<ABSGRAD, DET> lds;
  const double to_fix = DET ? det->to_fix : 0.0;
  unsigned long long c_visit = 0, c_live = 0, c_valid = 0;
  const int64_t tile = xcd_tile_index(total_tiles, n_xcd);
  if (tile >= total_tiles) return;
  if (masks != nullptr && !masks[tile]) return;
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

  const int32_t start = isect_offsets[tile];
  const int32_t end = (tile == total_tiles - 1) ? (int32_t)I : isect_offsets[tile + 1];
  if (end <= start) return;

  float T_final = 1.0f, vCr = 0.f, vCg = 0.f, vCb = 0.f, vNx = 0.f, vNy = 0.f, vNz = 0.f, vD = 0.f, vA = 0.f, vMed = 0.f;
  int32_t bin_final = -1, med_idx = -1;
  if (inside) {
    T_final = final_T != nullptr ? final_T[pid] : 1.0f - render_alphas[pid];
    bin_final = last_ids[pid];
    med_idx = median_ids[pid];
    vCr = v_render_colors[3 * pid]; vCg = v_render_colors[3 * pid + 1]; vCb = v_render_colors[3 * pid + 2];
    vNx = v_render_normals[3 * pid]; vNy = v_render_normals[3 * pid + 1]; vNz = v_render_normals[3 * pid + 2];
    vD = v_render_depths[pid]; vA = v_render_alphas[pid]; vMed = v_render_median[pid];
  }
  float bgdot = 0.f;
  if (backgrounds != nullptr)
    bgdot = backgrounds[3 * cam] * vCr + backgrounds[3 * cam + 1] * vCg + backgrounds[3 * cam + 2] * vCb;
  const float tfa = T_final * (vA - bgdot);   // the alpha / background term of every pair of this pixel: T_final (vA - bg . vC) / (1 - alpha)
  float T = T_final;
  float bCr = 0.f, bCg = 0.f, bCb = 0.f, bNx = 0.f, bNy = 0.f, bNz = 0.f, bD = 0.f;

  if (tid == 0) lds.bin_final_max = -1;
  for (int i = tid; i < BQB * NACC_D; i += RT) (&lds.acc[0][0])[i] = 0.0;
  for (int i = tid; i < BQB * 2; i += RT) { (&lds.acc2[0][0])[i] = 0; if (ABSGRAD) (&lds.acc_abs[0][0])[i] = 0; }
  __syncthreads();
  // last contributor of the quad (4 lanes), of the wave, of the tile
  const int quad_bin_final = quad_imax(bin_final);
  int wave_bin_final = quad_bin_final;
#pragma unroll
  for (int d = 4; d <= 32; d <<= 1) wave_bin_final = max(wave_bin_final, __shfl_xor(wave_bin_final, d, 64));
  wave_bin_final = __builtin_amdgcn_readfirstlane(wave_bin_final);   // uniform: say so (scalar list counters below)
  if (lane == 0) atomicMax(&lds.bin_final_max, wave_bin_final);
  __syncthreads();
  const int tile_bin_final = __builtin_amdgcn_readfirstlane(lds.bin_final_max);
  if (tile_bin_final < start) return;
  const unsigned char *my_list = lds.list[wave * 16 + (lane >> 2)];
  const int slot = thread_slot(tid);
  const int mbase = wave_mask_base(wave);
  const int f0 = quad_reduce16_first(lane), f4 = 16 + quad_reduce4_index(lane);

  int g_mine = -1;
  const int last = min(end, tile_bin_final + 1);
  const int nb = (last - start + BQB - 1) / BQB;
  for (int b = nb - 1; b >= 0; --b) {
    __syncthreads();  // barrier A: previous batch fully consumed, its accumulators complete
    flush_records_quads<ABSGRAD, DET>(lds, wave, lane, g_mine, grec, grec_abs, det);
    g_mine = -1;
    const int32_t bstart = start + b * BQB;
    const int32_t idx = bstart + slot;
    if (slot < BQB && idx < last) {
      g_mine = flatten_ids[idx];
      const float4 *r = rec + 8 * (int64_t)g_mine;
      const float4 r3 = r[3], r5 = r[5];
      lds.q0[slot] = r[0]; lds.q1[slot] = r[1]; lds.q2[slot] = r[2];
      lds.q3[slot] = make_float4(r5.x, r3.y, r3.z, r3.w);
      lds.q4[slot] = r[4];
      lds.mwy[slot] = r5.y;
      lds.m64[slot] = pair_masks[idx];
    }
    __syncthreads();  // barrier B
    const int count = min(BQB, last - bstart);
    const int wcount = min(count, wave_bin_final - bstart + 1);
    if (wcount <= 0) continue;
    // ---- step 1: the staged splats that reach a quad of this wave which still replays at this depth, compacted in list order
    const unsigned long long replays = __ballot(quad_bin_final >= bstart);
    const unsigned live_q = quads_any(replays);
    int ncomp = 0;
#pragma unroll
    for (int c = 0; c < BQ_CHUNKS; ++c) {
      if (64 * c < wcount) {
        const int ti = 64 * c + lane;
        const unsigned qb = ti < wcount ? (wave_quad_bits(lds.m64[ti], mbase) & live_q) : 0u;
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
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- step 2: one ballot per quad and compacted chunk; a quad drops what lies behind its last contributor
    unsigned cm[BQ_CHUNKS];
    int cti[BQ_CHUNKS];
#pragma unroll
    for (int c = 0; c < BQ_CHUNKS; ++c) {
      const int i = 64 * c + lane;
      const bool in = 64 * c < ncomp && i < ncomp;
      cm[c] = in ? (unsigned)lds.cmask[wave][i] : 0u;
      cti[c] = in ? (int)lds.cslot[wave][i] : 0;
    }
    int nvec = 0, kmax = 0;
    for_quads([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      int n = 0;
      const int qbf = __builtin_amdgcn_readlane(quad_bin_final, 4 * Q) - bstart;   // last batch slot the quad replays
#pragma unroll
      for (int c = 0; c < BQ_CHUNKS; ++c) {
        if (64 * c < ncomp) {
          const bool bit = ((cm[c] >> Q) & 1u) && (cti[c] <= qbf);
          const unsigned long long mk = __ballot(bit);
          if (bit) lds.list[wave * 16 + Q][n + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] = (unsigned char)cti[c];
          n += (int)__popcll(mk);
        }
      }
      writelane<4 * Q>(nvec, n);
      kmax = max(kmax, n);
    });
    const int n_mine = dpp_quad_i<0x00>(nvec);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int k = 0; k < kmax; ++k) {      // every quad walks its own list from the back
      const bool active = k < n_mine;
      const int t = active ? (int)my_list[n_mine - 1 - k] : 0;
      const float4 a0 = lds.q0[t], a1 = lds.q1[t], a2 = lds.q2[t], a3 = lds.q3[t];
      PairEval e;
      const float mwx = a3.x, mwy = lds.mwy[t];
      eval_pair<true>(0.f, 0.f, px, py, a0, a1, a2, mwx, a3.y, e, mwy);
      const bool valid = active && inside && (bstart + t <= bin_final) && e.ok;
      if (COUNT) { c_visit += 1; c_live += __popcll(__ballot(active && inside && (bstart + t <= bin_final))); c_valid += __popcll(__ballot(valid)); }
      const float4 a4 = lds.q4[t];
      const float cR = a3.z, cG = a3.w, cB = a4.x, nX = a4.y, nY = a4.z, nZ = a4.w;
      float g_rgb0 = 0.f, g_rgb1 = 0.f, g_rgb2 = 0.f, g_n0 = 0.f, g_n1 = 0.f, g_n2 = 0.f, g_op = 0.f;
      float vzx = 0.f, vzy = 0.f, vzz = 0.f, g_dx = 0.f, g_dy = 0.f, g_mwz = 0.f, g_x = 0.f, g_y = 0.f;
      bool v2 = false;
      if (valid) {
        const float ra = RASTER_BWD_RCP(1.0f - e.alpha);
        T *= ra;
        const float fac = e.alpha * T;
        g_rgb0 = fac * vCr; g_rgb1 = fac * vCg; g_rgb2 = fac * vCb;
        g_n0 = fac * vNx; g_n1 = fac * vNy; g_n2 = fac * vNz;
        float v_alpha = (cR * T - bCr * ra) * vCr + (cG * T - bCg * ra) * vCg + (cB * T - bCb * ra) * vCb;
        v_alpha += (nX * T - bNx * ra) * vNx + (nY * T - bNy * ra) * vNy + (nZ * T - bNz * ra) * vNz;
        v_alpha += (e.dep * T - bD * ra) * vD;
        v_alpha += tfa * ra;
        const float v_dep = fac * vD + ((bstart + t) == med_idx ? vMed : 0.f);
        bCr += cR * fac; bCg += cG * fac; bCb += cB * fac;
        bNx += nX * fac; bNy += nY * fac; bNz += nZ * fac;
        bD += e.dep * fac;
        float v_sigma = 0.f;
        if (!e.clamped) {
          g_op = e.vis * v_alpha;
          v_sigma = -a2.w * e.vis * v_alpha;
        }
        g_mwz = v_dep;
        if (e.b3) {
          vzx = fmaf(v_sigma, e.sx, v_dep * mwx) * e.inv;
          vzy = fmaf(v_sigma, e.sy, v_dep * mwy) * e.inv;
          vzz = -(vzx * e.sx + vzy * e.sy);
          g_dx = v_dep * e.sx; g_dy = v_dep * e.sy;
        } else {
          v2 = true;
          g_x = v_sigma * FILTER_INV_SQUARE * e.dx;
          g_y = v_sigma * FILTER_INV_SQUARE * e.dy;
        }
      }
      const bool any2 = __ballot(v2) != 0ull;
      int qv = (int)valid;
      qv |= dpp_quad_i<0xB1>(qv);
      qv |= dpp_quad_i<0x4E>(qv);
      const bool quad_valid = qv != 0;
      {  // slots 0..15: one transposing butterfly over the quad, four double adds per lane into the quad's own splat record
        const float mxp = -e.dx, myp = -e.dy;
        const float v16[16] = {g_rgb0, g_rgb1, g_rgb2, g_n0, g_n1, g_n2, g_op, vzx, vzy, vzz,
                               mxp * vzx, mxp * vzy, mxp * vzz, myp * vzx, myp * vzy, myp * vzz};
        float r4[4];
        quad_transpose_reduce16(v16, lane, r4);
        const float v4[4] = {g_dx, g_dy, g_mwz, 0.f};   // slots 16..18 (+ padding)
        const float r1 = quad_transpose_reduce4(v4, lane);
        if (quad_valid) {
          if (DET) {
            unsigned long long *a = reinterpret_cast<unsigned long long *>(&lds.acc[t][0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(a + f0 + i, (unsigned long long)det_fix(r4[i], to_fix));
            atomicAdd(a + f4, (unsigned long long)det_fix(r1, to_fix));
          } else {
            double *a = &lds.acc[t][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(a + f0 + i, (double)r4[i]);
            atomicAdd(a + f4, (double)r1);
          }
        }
      }
      if (any2) {  // screen-space low-pass branch (rare)
        auto add2 = [&](auto *p, float r) {
          if ((lane & 3) == 0 && r != 0.f) {
            if constexpr (DET) atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)det_fix(r, to_fix));
            else lds_add(reinterpret_cast<float *>(p), r);
          }
        };
        add2(&lds.acc2[t][0], quad_sum(g_x));
        add2(&lds.acc2[t][1], quad_sum(g_y));
        if (ABSGRAD) {
          add2(&lds.acc_abs[t][0], quad_sum(fabsf(g_x)));
          add2(&lds.acc_abs[t][1], quad_sum(fabsf(g_y)));
        }
      }
    }
  }
  __syncthreads();
  flush_records_quads<ABSGRAD, DET>(lds, wave, lane, g_mine, grec, grec_abs, det);
  if (COUNT && lane == 0) { atomicAdd(counters + 4, c_visit); atomicAdd(counters + 5, c_live); atomicAdd(counters + 6, c_valid); }
}

// Streaming epilogue: unpack the 80-byte records into the operator's gradient tensors and derive the
// densification signal (SPEC S-4, 2DGS convention consumed at neural_gaussian.cpp:660-665):
// v_densify = (dL/dM_u.z, dL/dM_v.z) * M_w.z.
// deterministic mode: the largest |upstream gradient| of the launch (unsigned max of bit patterns: order-independent) and the header made of it
__global__ void __launch_bounds__(256)
    det_upstream_max_kernel(int64_t P, const float *__restrict__ v_rc, const float *__restrict__ v_rd, const float *__restrict__ v_ra,
                            const float *__restrict__ v_rn, const float *__restrict__ v_rm, DetHeader *__restrict__ det) {
  __shared__ unsigned s_max;
  if (threadIdx.x == 0) s_max = 0u;
  __syncthreads();
  unsigned m = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
    const float v[9] = {v_rc[3 * i], v_rc[3 * i + 1], v_rc[3 * i + 2], v_rd[i], v_ra[i], v_rn[3 * i], v_rn[3 * i + 1], v_rn[3 * i + 2], v_rm[i]};
#pragma unroll
    for (int k = 0; k < 9; ++k) m = max(m, __float_as_uint(v[k]) & 0x7FFFFFFFu);   // (a NaN / Inf pattern is above every finite one)
  }
  atomicMax(&s_max, m);
  __syncthreads();
  if (threadIdx.x == 0 && s_max) atomicMax(&det->max_bits, s_max);
}
__global__ void det_header_kernel(DetHeader *__restrict__ det) {
  const unsigned b = det->max_bits;
  int e = (int)(b >> 23) - 126;                    // max < 2^e
  if (b >= 0x7F800000u) { det->poisoned = 1u; e = 0; }
  if (b == 0u) e = -100;
  det->to_fix = ldexp(1.0, DET_UNIT_SHIFT - e);
  det->from_fix = ldexp(1.0, e - DET_UNIT_SHIFT);
}

template <bool DET>
__global__ void __launch_bounds__(256)
    unpack_records_kernel(int64_t M, const float *__restrict__ grec, const float *__restrict__ grec_abs,
                          const float *__restrict__ means2d, const float *__restrict__ ray_transforms,
                          float *__restrict__ v_means2d, float *__restrict__ v_ray_transforms,
                          float *__restrict__ v_colors, float *__restrict__ v_opacities, float *__restrict__ v_normals,
                          float *__restrict__ v_densify, float *__restrict__ v_means2d_abs, const DetHeader *__restrict__ det = nullptr) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float r[NACC], rabs[2] = {0.f, 0.f};
  if (DET) {
    const double from_fix = det->poisoned ? (double)__builtin_nanf("") : det->from_fix;
    const long long *g64 = reinterpret_cast<const long long *>(grec);
#pragma unroll
    for (int k = 0; k < NACC; ++k) r[k] = (float)((double)g64[m * NACC + k] * from_fix);
    if (v_means2d_abs != nullptr) {
      const long long *a64 = reinterpret_cast<const long long *>(grec_abs);
      rabs[0] = (float)((double)a64[2 * m] * from_fix); rabs[1] = (float)((double)a64[2 * m + 1] * from_fix);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NACC; ++k) r[k] = grec[m * NACC + k];
    if (v_means2d_abs != nullptr) { rabs[0] = grec_abs[2 * m]; rabs[1] = grec_abs[2 * m + 1]; }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { v_colors[3 * m + k] = r[k]; v_normals[3 * m + k] = r[3 + k]; }
  v_opacities[m] = r[6];
  v_means2d[2 * m] = r[19]; v_means2d[2 * m + 1] = r[20];
  if (v_means2d_abs != nullptr) { v_means2d_abs[2 * m] = rabs[0]; v_means2d_abs[2 * m + 1] = rabs[1]; }
  // moments -> dL/dM.  With h_u = m_x M_w - M_u, h_v = m_y M_w - M_v evaluated at the splat centre (m_x, m_y):
  //   v_hu = h_v x V0 + M_w x Vy,   v_hv = V0 x h_u + Vx x M_w,
  //   dL/dM_u = -v_hu,   dL/dM_v = -v_hv,
  //   dL/dM_w = m_x v_hu + m_y v_hv + h_v x Vx + Vy x h_u + (sum v_dep s.x, sum v_dep s.y, sum v_dep)
  const float *Mr = ray_transforms + 9 * m;
  const float mu[3] = {Mr[0], Mr[1], Mr[2]}, mv[3] = {Mr[3], Mr[4], Mr[5]}, mw[3] = {Mr[6], Mr[7], Mr[8]};
  const float mx = means2d[2 * m], my = means2d[2 * m + 1];
  const float hu[3] = {fmaf(mx, mw[0], -mu[0]), fmaf(mx, mw[1], -mu[1]), fmaf(mx, mw[2], -mu[2])};   // one rounding each (see stage_splat)
  const float hv[3] = {fmaf(my, mw[0], -mv[0]), fmaf(my, mw[1], -mv[1]), fmaf(my, mw[2], -mv[2])};
  const float *V0 = r + 7, *Vx = r + 10, *Vy = r + 13;
#define CROSS(o, a, b) o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]
  float t1[3], t2[3], vhu[3], vhv[3], t3[3], t4[3];
  CROSS(t1, hv, V0); CROSS(t2, mw, Vy);
  CROSS(t3, V0, hu); CROSS(t4, Vx, mw);
#pragma unroll
  for (int k = 0; k < 3; ++k) { vhu[k] = t1[k] + t2[k]; vhv[k] = t3[k] + t4[k]; }
  CROSS(t1, hv, Vx); CROSS(t2, Vy, hu);
#undef CROSS
  float gmu[3], gmv[3], gmw[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gmu[k] = -vhu[k];
    gmv[k] = -vhv[k];
    gmw[k] = mx * vhu[k] + my * vhv[k] + t1[k] + t2[k] + r[16 + k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v_ray_transforms[9 * m + k] = gmu[k]; v_ray_transforms[9 * m + 3 + k] = gmv[k]; v_ray_transforms[9 * m + 6 + k] = gmw[k];
  }
  v_densify[2 * m] = gmu[2] * mw[2];
  v_densify[2 * m + 1] = gmv[2] * mw[2];
}

}  // namespace gsdf

using namespace gsdf;

// (sized for the deterministic mode's 64-bit records whether or not it is on: the size query and the launch cannot disagree)
static size_t bwd_records_bytes(int64_t M) {
  return align_up((size_t)(M > 0 ? M : 1) * NACC * sizeof(long long), 256) + align_up((size_t)(M > 0 ? M : 1) * 2 * sizeof(long long), 256) + 256;
}
// gradient records + room for the pack / mask passes (used when the caller does not hand over the forward's workspace)
extern "C" size_t gsdf_rasterize_2dgs_bwd_ws_bytes(int64_t M, int64_t I) { return bwd_records_bytes(M) + raster_pack_bytes(M, I); }

static int rasterize_bwd_launch(int64_t C, int64_t M, int64_t I, int width, int height, int tile_size,
                                       const float *means2d, const float *ray_transforms, const float *colors,
                                       const float *opacities, const float *normals, const float *backgrounds,
                                       const uint8_t *masks, const int32_t *isect_offsets,
                                       const int32_t *flatten_ids, const float *render_alphas,
                                       const int32_t *last_ids, const int32_t *median_ids,
                                       const float *v_render_colors, const float *v_render_depths,
                                       const float *v_render_alphas, const float *v_render_normals,
                                       const float *v_render_median, float *v_means2d, float *v_ray_transforms,
                                       float *v_colors, float *v_opacities, float *v_normals, float *v_densify,
                                       float *v_means2d_abs, void *ws, const float *final_T, const void *fwd_ws, unsigned long long *counters,
                                       hipStream_t stream) {
  GSDF_REQUIRE(tile_size == TILE, "rasterize_bwd: tile_size %d unsupported (16 only)", tile_size);
  GSDF_REQUIRE(width > 0 && height > 0 && C >= 1, "rasterize_bwd: bad geometry");
  if (M == 0) return GSDF_OK;
  GSDF_REQUIRE(v_means2d && v_ray_transforms && v_colors && v_opacities && v_normals && v_densify && ws,
               "rasterize_bwd: null gradient output / workspace");
  GSDF_REQUIRE(render_alphas && last_ids && median_ids && v_render_colors && v_render_depths && v_render_alphas &&
                   v_render_normals && v_render_median && isect_offsets,
               "rasterize_bwd: null input");
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE;
  const int64_t n_tiles = (int64_t)tw * th, total = n_tiles * C;
  const bool det = deterministic() && counters == nullptr;
  const size_t rec_size = det ? sizeof(long long) : sizeof(float);
  float *grec = (float *)ws;
  float *grec_abs = (float *)((char *)ws + align_up((size_t)M * NACC * rec_size, 256));
  DetHeader *det_hdr = (DetHeader *)((char *)ws + bwd_records_bytes(M) - 256);
  GSDF_HIP(hipMemsetAsync(grec, 0, (size_t)M * NACC * rec_size, stream), "rasterize_bwd memset");
  if (v_means2d_abs) GSDF_HIP(hipMemsetAsync(grec_abs, 0, (size_t)M * 2 * rec_size, stream), "rasterize_bwd memset");
  if (det) {
    GSDF_HIP(hipMemsetAsync(det_hdr, 0, sizeof(DetHeader), stream), "rasterize_bwd memset");
    det_upstream_max_kernel<<<1024, 256, 0, stream>>>(C * (int64_t)height * width, v_render_colors, v_render_depths, v_render_alphas, v_render_normals,
                                                      v_render_median, det_hdr);
    det_header_kernel<<<1, 1, 0, stream>>>(det_hdr);
    GSDF_CHECK_LAUNCH("det_header_kernel");
  }
  if (I > 0) {
    const int n_xcd = xcd_count(stream);
    if (fwd_ws == nullptr) {   // no forward workspace handed over: run the pack + mask passes into this call's own
      void *own = (char *)ws + bwd_records_bytes(M);
      const int rc = raster_pack_launch(M, I, total, n_tiles, tw, means2d, ray_transforms, colors, opacities, normals, isect_offsets, flatten_ids, own, stream);
      if (rc != GSDF_OK) return rc;
      fwd_ws = own;
    }
#define QARGS n_xcd, total, n_tiles, I, width, height, tw, (const float4 *)ws_records(fwd_ws), ws_masks(fwd_ws, M), backgrounds, masks, isect_offsets, \
              flatten_ids, render_alphas, last_ids, median_ids, v_render_colors, v_render_depths, v_render_alphas, v_render_normals,               \
              v_render_median, grec, grec_abs, final_T
    if (det && v_means2d_abs)
      raster_bwd_quads_kernel<true, false, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(QARGS, nullptr, det_hdr);
    else if (det)
      raster_bwd_quads_kernel<false, false, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(QARGS, nullptr, det_hdr);
    else if (counters != nullptr && v_means2d_abs)
      raster_bwd_quads_kernel<true, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(QARGS, counters);
    else if (counters != nullptr)
      raster_bwd_quads_kernel<false, true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(QARGS, counters);
    else if (v_means2d_abs)
      raster_bwd_quads_kernel<true><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(QARGS);
    else
      raster_bwd_quads_kernel<false><<<xcd_grid(total, n_xcd), RT, 0, stream>>>(QARGS);
#undef QARGS
    GSDF_CHECK_LAUNCH("raster_bwd_quads_kernel");
  }
  if (det)
    unpack_records_kernel<true><<<(unsigned)((M + 255) / 256), 256, 0, stream>>>(M, grec, grec_abs, means2d, ray_transforms, v_means2d, v_ray_transforms,
                                                                                 v_colors, v_opacities, v_normals, v_densify, v_means2d_abs, det_hdr);
  else
    unpack_records_kernel<false><<<(unsigned)((M + 255) / 256), 256, 0, stream>>>(M, grec, grec_abs, means2d, ray_transforms, v_means2d,
                                                                                  v_ray_transforms, v_colors, v_opacities,
                                                                                  v_normals, v_densify, v_means2d_abs);
  GSDF_CHECK_LAUNCH("unpack_records_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_rasterize_2dgs_bwd(int64_t C, int64_t M, int64_t I, int width, int height, int tile_size,
                                       const float *means2d, const float *ray_transforms, const float *colors,
                                       const float *opacities, const float *normals, const float *backgrounds,
                                       const uint8_t *masks, const int32_t *isect_offsets,
                                       const int32_t *flatten_ids, const float *render_alphas,
                                       const int32_t *last_ids, const int32_t *median_ids,
                                       const float *v_render_colors, const float *v_render_depths,
                                       const float *v_render_alphas, const float *v_render_normals,
                                       const float *v_render_median, float *v_means2d, float *v_ray_transforms,
                                       float *v_colors, float *v_opacities, float *v_normals, float *v_densify,
                                       float *v_means2d_abs, void *ws, const float *final_T, const void *fwd_ws, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_rasterize_2dgs_bwd");
  return rasterize_bwd_launch(C, M, I, width, height, tile_size, means2d, ray_transforms, colors, opacities, normals, backgrounds, masks,
                              isect_offsets, flatten_ids, render_alphas, last_ids, median_ids, v_render_colors, v_render_depths,
                              v_render_alphas, v_render_normals, v_render_median, v_means2d, v_ray_transforms, v_colors, v_opacities,
                              v_normals, v_densify, v_means2d_abs, ws, final_T, fwd_ws, nullptr, (hipStream_t)stream_);
}

extern "C" int gsdf_rasterize_2dgs_bwd_instr(int64_t C, int64_t M, int64_t I, int width, int height, int tile_size,
                                       const float *means2d, const float *ray_transforms, const float *colors,
                                       const float *opacities, const float *normals, const float *backgrounds,
                                       const uint8_t *masks, const int32_t *isect_offsets,
                                       const int32_t *flatten_ids, const float *render_alphas,
                                       const int32_t *last_ids, const int32_t *median_ids,
                                       const float *v_render_colors, const float *v_render_depths,
                                       const float *v_render_alphas, const float *v_render_normals,
                                       const float *v_render_median, float *v_means2d, float *v_ray_transforms,
                                       float *v_colors, float *v_opacities, float *v_normals, float *v_densify,
                                       float *v_means2d_abs, void *ws, const float *final_T, const void *fwd_ws, const gsdf_raster_instr *instr,
                                       gsdf_stream_t stream_) {
  return rasterize_bwd_launch(C, M, I, width, height, tile_size, means2d, ray_transforms, colors, opacities, normals, backgrounds, masks,
                              isect_offsets, flatten_ids, render_alphas, last_ids, median_ids, v_render_colors, v_render_depths,
                              v_render_alphas, v_render_normals, v_render_median, v_means2d, v_ray_transforms, v_colors, v_opacities,
                              v_normals, v_densify, v_means2d_abs, ws, final_T, fwd_ws, instr ? instr->counters : nullptr, (hipStream_t)stream_);
}
