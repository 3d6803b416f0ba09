// raster_fwd.hip — P4: 2DGS alpha compositing, forward.
// Replaces rasterize_to_pixels_2dgs of the reference's absent gsplat_cpp submodule (call site
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:215-223); semantics: SPEC A.4.
#include "raster_quad.h"

namespace gsdf {

// QUAD LISTS (round 6, raster_quad.h).  Every lane quad (2x2 pixels) follows its own list of the staged splats whose 64-bit reach mask has the
// quad's bit; the staging copies 80 bytes of the packed record + the mask.  Each pixel sees its tile's splats in list order and evaluates
// them with the operands of rounds 3-5, so every output is bit-identical to the row-list kernels of round 5 (measured at cfg3 before they
// were removed: profiles/r06_raster_quad_vs_row_lists_cfg3.json).  Quads whose four pixels are finished get no list (termination mask).
//
// COUNT: diagnostic instantiation (gsdf_rasterize_2dgs_fwd_instr with counters): per launch, counters[0] += wave iterations of the compositing
// loop, [1] += lanes of those iterations that follow a list entry and are still live, [2] += lanes that pass the alpha test, [3] += lanes that
// blend, [7] += iterations without a lane passing the alpha test.
// TRACE: the decision record of the parity gate (gsdf_raster_instr.trace_*): a pixel with trace_rows[pid] >= 0 writes one byte per list
// position it takes a decision at — bit0 blended, bit1 3-D footprint branch, bit2 alpha clamped, bit3 the pixel terminates at this pair
// (not blended), bit4 the median is updated here; positions it skips (alpha test failed, unreachable quad, pixel finished) stay 0.
// Measured at cfg3 (tools/exp_raster_quads.py, library variants of tools/build_variants.sh; pack + mask + forward, ms): batch 192 at 5 workgroups
// per CU 0.343, 160 at 6 0.313, 144 at 6 0.302, 128 at 7 0.299, 112 at 8 0.291, 240 at 4 0.330: the kernel is latency-bound (VALU issue 49 %, LDS
// 26 % busy at 5 per CU), so the batch is what eight workgroups' LDS allows (62 registers: eight waves per SIMD fit).  Requesting the next
// iteration's q0..q2 before the current one computes (software pipeline, 63 registers) changes nothing: 0.302 against 0.296.
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
  {   // pack + mask passes, then the compositing kernel
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
  }
  GSDF_CHECK_LAUNCH("raster_fwd_quads_kernel");
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
