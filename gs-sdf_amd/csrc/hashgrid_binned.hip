// hashgrid_binned.hip — S1': the table gradient of the hash-grid encoding WITHOUT global atomics.
// Same result as gsdf_hashgrid_bwd(..., v_table, v_x = NULL) (replaces tiny-cuda-nn's GridEncoding backward behind
// TCNNEncoding, /root/reference/include/neural_net/encoding_map.cpp:15-26,59), for the large batches of the joint
// iteration (7 x (ray points + visible splat samples) ~ 3 M query points, neural_mapping.cpp:106-136,448-451).
//
// Why: fp32 global atomics retire ~21 G requests/s chip-wide on MI355X whatever the footprint, scope or XCD placement
// (tools/ubench/atomic_rate.hip, atomic_xcd.hip): they are executed memory-side.  A query point makes 16 levels x 8
// corners = 128 contributions to random places of a 61 MB table, i.e. ~72 requests -> 3 M points cost >= 10 ms.
// Plain traffic is an order of magnitude cheaper, so the scatter becomes a one-digit MSD radix sort + LDS accumulation:
//
//   count  : per (level, table tile of 8192 entries = 64 KB) how many contributions arrive          (LDS histograms)
//   plan   : exclusive scan of the bucket sizes; work items of <= bin_item_records(B) records for the last pass (one workgroup)
//   emit   : every contribution becomes an 8-byte record {entry within tile, block-float (g0, g1)} (pack_record); a workgroup (one thread per (point,
//            level), 8 waves) sorts the records of its 256 points x 2 levels by bucket in LDS and appends each run to its bucket with ONE reservation per
//            (workgroup, bucket) and fully coalesced stores
//   apply  : one workgroup per item: the tile lives in LDS (64 KB) as 64-BIT FIXED POINT, records stream in coalesced,
//            ds_add_u64, then the tile is added to the gradient with plain coalesced read-modify-writes (the tile has
//            exactly one owner; an item of a bucket that was split leaves its tile as a slab of 64-bit integers: round 6).  Fixed point because ds_add_f32 retires ~0.3 lanes/clock/CU on gfx950 — 18x slower than the
//            integer LDS atomics (tools/ubench/lds_atomic.hip: 0.2 T/s against 3.6 T/s chip-wide); the scale is a power
//            of two per level, 2^41 / 2^ceil(log2 max|v_feat|), so a tile's sum is EXACT up to 2^-41 of the largest
//            contribution per record and independent of the order of the records (bit-reproducible, unlike atomics).
//   reduce : the slabs of a split bucket are summed (integers) and added to the gradient, 8 workgroups per bucket
//
// Traffic: 8 B written + 8 B read per contribution = 2 KB per query point, all of it coalesced.
// Workspace: sized for the worst case, 8 B x 8 corners x n_levels per point (1 KB per point: 0.5 GB for the step's 0.49 M base rows,
// 3.4 GB for a 3.3 M-row stencil batch before merging) + 32 MB of slabs for the split buckets — gsdf_hashgrid_bwd_binned_ws_bytes; a caller that cannot afford it passes the
// batch to gsdf_hashgrid_bwd (atomics, no workspace), which is what the host layers do when the size query returns 0.
// Non-finite contributions: bin_vmax marks the level, bin_apply writes NaN into the level's touched tiles (nothing is silently dropped).
// The order in which records are summed is not fixed, but every sum is an integer sum: the result is bit-reproducible from call to call
// (the only exception: more than BIN_SLOTS = 256 split buckets or BIN_SLABS = 512 items in them, whose surplus falls back to float atomics).
#include "hashgrid_common.h"

namespace gsdf {

static constexpr int BIN_TILE_LOG2 = 12;               // 4096 entries x 2 x int64 = 64 KB of LDS
static constexpr int BIN_TILE = 1 << BIN_TILE_LOG2;
#ifndef GSDF_BIN_PTS
#define GSDF_BIN_PTS 256
#endif
static constexpr int BIN_PTS = GSDF_BIN_PTS;            // points per emit workgroup
static constexpr int BIN_G = 2;                        // levels per emit workgroup (16 B of v_feat per point)
static constexpr int BIN_MAX_LOCAL = 256;              // buckets one emit workgroup can address
#ifndef GSDF_BIN_GPW
#define GSDF_BIN_GPW 1
#endif
static constexpr int BIN_GPW = GSDF_BIN_GPW;           // level groups one emit workgroup walks (its points' feature rows are read once per GPW groups)
#ifndef GSDF_BIN_ITEM_MIN
#define GSDF_BIN_ITEM_MIN (32 * 1024)
#endif
#ifndef GSDF_BIN_APPLY_INFLIGHT
#define GSDF_BIN_APPLY_INFLIGHT 8
#endif
static constexpr int64_t BIN_ITEM_MAX = 512 * 1024;    // most records an apply work item may hold (fixed-point headroom: 2^41 x 2^19 < 2^62)
static constexpr int64_t BIN_ITEM_MIN = GSDF_BIN_ITEM_MIN;
static constexpr int BIN_SLOTS = 256;                  // split buckets whose items leave their tiles as 64-bit slabs (the order-independent flush) ...
static constexpr int BIN_SLABS = 512;                  // ... and the slabs they share (64 KB each: 32 MB of workspace)
static constexpr int BIN_SEGS = 8;                     // reduce pass: workgroups per split bucket
static constexpr int BIN_SLOT_LOG2_RECORDS = 21;       // records a bucket sums at full resolution (2^41 x 2^21 < 2^63); beyond: pre-shifted sums
// Records per work item.  The apply pass is one workgroup per item and ends with its longest item: round 5's fixed 2^19 left the dense
// coarse levels (level 0: 8 n records in 9 tiles) as a handful of 375 us items under a 345 us kernel.  Now an item holds about 1.6 x the
// records of an average HASHED bucket (8 B / 128 tiles per level): hashed buckets stay whole (one owner, plain read-modify-write flush),
// the dense levels' buckets split into many items, each of which leaves its tile as a 64-bit slab for the reduce pass (apply: 345 -> 140 us
// at 0.44 M points).
static int64_t bin_item_records(int64_t B) {
  int64_t r = BIN_ITEM_MIN;
  while (r < BIN_ITEM_MAX && r * 10 < B) r <<= 1;
  return r;
}
static constexpr int BIN_APPLY_THREADS = 512;
static_assert(BIN_PTS >= BIN_MAX_LOCAL, "the emit kernel scans its local histogram with one thread per bucket");

struct BinPlan {
  int tile_base[HG_MAX_LEVELS + 1];  // first bucket of each level; [n_levels] = number of buckets
  int n_groups;                      // level groups of BIN_G levels
};

// Round 4: 8-byte records (12 before).  A record = the entry within its tile (12 bits) + the two gradient values as a BLOCK FLOAT relative
// to the level's largest contribution 2^e (bin_vmax): shift s = e - e_rec (6 bits; e_rec = exponent of the larger of the two) and two signed
// 23-bit mantissas m = round(g 2^(22 - e_rec)).  The apply pass turns (s, m) into the same 2^(e - 41) fixed point it accumulated before
// (m << (19 - s)), so the sums stay exact integer sums, independent of the order of the records (bit-reproducible); what changed is that a
// contribution is rounded to 22 bits relative to the larger value of ITS record (2.4e-7) instead of entering with its 24: the table gradient
// agrees with the oracle's double-precision accumulation to ~1e-6 as before (tests/test_gpu_sdf_parity.py), and the scatter moves 2/3 of
// the bytes (records are written once and read once: 16 B per contribution instead of 24).
typedef unsigned long long BinRecord;
static constexpr int BIN_EXP_HEADROOM = 3;   // 2^e bounds ONE contribution; a base row that carries its six stencil rows sums up to seven
__device__ __forceinline__ BinRecord pack_record(uint32_t entry, float g0, float g1, int e_level) {
  const float mx = fmaxf(fabsf(g0), fabsf(g1));
  const int e_rec = (int)((__float_as_uint(mx) >> 23) & 255u) - 126;            // mx < 2^e_rec (0 -> -126: everything rounds to 0)
  int s = e_level - e_rec;
  s = s < 0 ? 0 : (s > 63 ? 63 : s);
  const int up = 22 - (e_level - s);                                           // m = round(g 2^(22 - e_rec')), e_rec' = e_level - s (v_ldexp_f32: exact)
  int m0 = (int)rintf(ldexpf(g0, up)), m1 = (int)rintf(ldexpf(g1, up));
  m0 = max(-4194303, min(4194303, m0)); m1 = max(-4194303, min(4194303, m1));
  return (BinRecord)(entry & 0xFFFu) | ((BinRecord)(unsigned)s << 12) | ((BinRecord)((unsigned)m0 & 0x7FFFFFu) << 18) |
         ((BinRecord)((unsigned)m1 & 0x7FFFFFu) << 41);
}
__device__ __forceinline__ void unpack_record(BinRecord r, uint32_t &entry, long long &f0, long long &f1) {
  entry = (uint32_t)r & 0xFFFu;
  const int s = (int)((r >> 12) & 63u);
  const long long m0 = ((long long)(r << 23)) >> 41, m1 = ((long long)r) >> 41;   // sign-extended 23-bit fields
  // fixed point of the level: 2^(e - 41); the record's unit is 2^(e - s - 22)
  // contributions more than 2^19 below the level's maximum lose their low bits here: ROUND to nearest (an arithmetic shift alone floors toward
  // minus infinity, a one-sided bias of up to one fixed-point unit on every small contribution)
  f0 = s <= 19 ? m0 << (19 - s) : (m0 + (1LL << (s - 20))) >> (s - 19);
  f1 = s <= 19 ? m1 << (19 - s) : (m1 + (1LL << (s - 20))) >> (s - 19);
}

struct BinItem {
  int64_t begin, end;  // record range
  int32_t bucket, nit;   // nit > 1: the bucket has several items
  int32_t slab, shift;   // slab >= 0: where this item leaves its tile as 64-bit integers (plain stores; bin_reduce sums a bucket's slabs — integer sums,
                         // order-independent — and adds them to the gradient); shift: the sums are rounded by that many bits first (buckets of more than
                         // 2^21 records); slab < 0 (more than BIN_SLOTS split buckets or BIN_SLABS slabs): float atomics on the gradient itself
};
struct BinSlot {
  int32_t bucket, slab0, nit, shift;   // nit == 0: not in use
};

static bool make_plan(const HgLevels &lv, BinPlan *bp) {
  int t = 0;
  for (int l = 0; l < lv.n_levels; ++l) {
    bp->tile_base[l] = t;
    t += (int)((lv.hsize[l] + BIN_TILE - 1) >> BIN_TILE_LOG2);
  }
  for (int l = lv.n_levels; l <= HG_MAX_LEVELS; ++l) bp->tile_base[l] = t;
  bp->n_groups = (lv.n_levels + BIN_G - 1) / BIN_G;
  for (int g = 0; g < bp->n_groups; ++g) {
    const int l0 = g * BIN_G, l1 = l0 + BIN_G < lv.n_levels ? l0 + BIN_G : lv.n_levels;
    if (bp->tile_base[l1] - bp->tile_base[l0] > BIN_MAX_LOCAL) return false;
  }
  return t <= 4096;
}

// workspace layout (all 256-byte aligned): counts u32[nb] + lmax u32[16] | cursor u32[nb] | start i64[nb+1] | n_items u32, n_slots u32 (+pad) |
// slots BinSlot[BIN_SLOTS] | slabs i64[BIN_SLABS][2 * BIN_TILE] | items BinItem[max_items] | records BinRecord[B * n_levels * 8]
struct BinWs {
  uint32_t *counts, *lmax, *cursor, *n_items;
  BinSlot *slots;
  long long *slabs;
  int64_t item_records;
  int64_t *start;
  BinItem *items;
  BinRecord *records;
  int64_t max_items;
  size_t bytes;
};

static BinWs carve(void *ws, int64_t B, int n_levels, int nb) {
  BinWs w;
  char *p = (char *)ws;
  size_t o = 0;
  auto take = [&](size_t n) { char *q = p ? p + o : nullptr; o = align_up(o + n, 256); return q; };
  w.counts = (uint32_t *)take(sizeof(uint32_t) * (nb + HG_MAX_LEVELS));
  w.lmax = p ? w.counts + nb : nullptr;
  w.cursor = (uint32_t *)take(sizeof(uint32_t) * nb);
  w.start = (int64_t *)take(sizeof(int64_t) * (nb + 1));
  w.n_items = (uint32_t *)take(256);
  w.slots = (BinSlot *)take(sizeof(BinSlot) * BIN_SLOTS);
  w.slabs = (long long *)take(sizeof(long long) * 2 * BIN_TILE * BIN_SLABS);
  const int64_t total = B * n_levels * 8;
  w.item_records = bin_item_records(B);
  w.max_items = total / w.item_records + nb + 1;
  w.items = (BinItem *)take(sizeof(BinItem) * (size_t)w.max_items);
  w.records = (BinRecord *)take(sizeof(BinRecord) * (size_t)total);
  w.bytes = o;
  return w;
}

// The 8 corner contributions of (point b, level): entry indices and weight * v_feat.
struct Corner8 {
  uint32_t idx[8];
  float w[8];
};
__device__ __forceinline__ void corners_of(const HgLevels &lv, int level, float px, float py, float pz, Corner8 &c) {
  const float scale = lv.scale[level];
  const uint32_t res = lv.res[level], hsize = lv.hsize[level];
  uint32_t g0[3];
  float fr[3];
  const float xin[3] = {px, py, pz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float pos = fmaf(scale, xin[d], 0.5f);
    const float fl = floorf(pos);
    g0[d] = (uint32_t)(int32_t)fl;
    fr[d] = pos - fl;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int hx = k & 1, hy = (k >> 1) & 1, hz = k >> 2;
    c.idx[k] = grid_index(hsize, res, g0[0] + hx, g0[1] + hy, g0[2] + hz);
    // same association as hashgrid_bwd_kernel: (wx * wy) * wz
    const float wx = hx ? fr[0] : 1.f - fr[0], wy = hy ? fr[1] : 1.f - fr[1], wz = hz ? fr[2] : 1.f - fr[2];
    c.w[k] = wx * wy * wz;
  }
}

// d/dx of the 8 trilinear weights contracted with vv: dw[k] = scale * sum_d vv[d] * d w_k / d pos_d  (same expression as
// hashgrid_bwd_bwd_kernel's `t`): the weight with which v_feat2 reaches the table in the SECOND-ORDER term
// d/d table [(J(x, table)^T v_feat2) . vv_x] of the analytic eikonal regulariser.
__device__ __forceinline__ void corner_dweights(const HgLevels &lv, int level, float px, float py, float pz, float vx, float vy,
                                                float vz, float (&dw)[8]) {
  const float scale = lv.scale[level];
  float fr[3];
  const float xin[3] = {px, py, pz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float pos = fmaf(scale, xin[d], 0.5f);
    fr[d] = pos - floorf(pos);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int hx = k & 1, hy = (k >> 1) & 1, hz = k >> 2;
    const float wx = hx ? fr[0] : 1.f - fr[0], wy = hy ? fr[1] : 1.f - fr[1], wz = hz ? fr[2] : 1.f - fr[2];
    const float sx = hx ? 1.f : -1.f, sy = hy ? 1.f : -1.f, sz = hz ? 1.f : -1.f;
    dw[k] = scale * (vx * sx * wy * wz + vy * sy * wx * wz + vz * sz * wx * wy);
  }
}

// Stencil structure of a batch (optional): rows [0, n) are base points, rows n + k n + i (k = 0..5) the six central-
// difference points of base i (gsdf_sdf_query_points).  At the coarse levels the seven points of a group usually lie in
// the same grid cell, i.e. touch the same 8 entries: their contributions are then summed in registers and emitted as ONE
// set of 8 records by the base row; a stencil row emits nothing at a level where it shares the cell of its base row.
struct BinStencil {
  int64_t n;           // 0: no structure
  int merge_levels;    // levels [0, merge_levels) are checked for merging
};
__device__ __forceinline__ void cell_coords(const HgLevels &lv, int level, float px, float py, float pz, int32_t g[3]) {
  const float scale = lv.scale[level];
  g[0] = (int32_t)floorf(fmaf(scale, px, 0.5f));
  g[1] = (int32_t)floorf(fmaf(scale, py, 0.5f));
  g[2] = (int32_t)floorf(fmaf(scale, pz, 0.5f));
}
// does stencil row b (>= n) share the level's cell with its base row?
__device__ __forceinline__ bool merged_into_base(const HgLevels &lv, int level, const BinStencil &stn, const float *__restrict__ x,
                                                 int64_t b, float px, float py, float pz) {
  const int64_t i = (b - stn.n) % stn.n;
  int32_t gb[3], gs[3];
  cell_coords(lv, level, x[3 * i], x[3 * i + 1], x[3 * i + 2], gb);
  cell_coords(lv, level, px, py, pz, gs);
  return gb[0] == gs[0] && gb[1] == gs[1] && gb[2] == gs[2];
}

// ---- level maxima of |v_feat| (bit patterns of non-negative floats: unsigned max) -------------------------------------
// corner weights are <= 1, so this bounds every contribution of the level; it fixes the apply pass's fixed point.
// A reduction kernel of its own (0.1 ms at 3 M points): folding it into the emit kernel as one atomicMax per wave puts
// ~1e6 atomics on the ONE 64-byte line that holds the 16 maxima and costs 3-12 ms of serialisation, filtered or not.
// With a second-order term the bound of (point, level) is |v_feat| + scale_l * |vv_x|_1 * |v_feat2| (|d w / d pos| <= 1).
__global__ void __launch_bounds__(256)
    bin_vmax_kernel(int64_t n2, int n_levels, const float2 *__restrict__ v_feat, uint32_t *__restrict__ lmax,
                    const float2 *__restrict__ v_feat2 = nullptr, const float *__restrict__ vv_x = nullptr, HgLevels lv = HgLevels()) {
  __shared__ uint32_t s_max[HG_MAX_LEVELS];
  if (threadIdx.x < HG_MAX_LEVELS) s_max[threadIdx.x] = 0u;
  __syncthreads();
  // element i of the [B, n_levels] array of feature pairs belongs to level i % n_levels; the stride is a multiple of
  // n_levels, so a lane keeps its level and needs ONE LDS atomic at the end
  const int64_t stride = (int64_t)gridDim.x * 256 / n_levels * n_levels;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float m = 0.f;
  bool bad = false;   // a NaN / Inf contribution: fmaxf would drop the NaN and the fixed-point conversion would turn it into finite garbage
  if (i0 < stride)
    for (int64_t i = i0; i < n2; i += stride) {
      float b = 0.f;
      if (v_feat != nullptr) {
        const float2 v = v_feat[i];
        b = fmaxf(fabsf(v.x), fabsf(v.y));
        bad |= !(fabsf(v.x) < __builtin_inff()) || !(fabsf(v.y) < __builtin_inff());   // (fmaxf returns the other operand for a NaN)
      }
      if (v_feat2 != nullptr) {
        const float2 v = v_feat2[i];
        const int64_t pt = i / n_levels;
        const float l1 = fabsf(vv_x[3 * pt]) + fabsf(vv_x[3 * pt + 1]) + fabsf(vv_x[3 * pt + 2]);
        b += lv.scale[(int)(i0 % n_levels)] * l1 * fmaxf(fabsf(v.x), fabsf(v.y));
        bad |= !(fabsf(v.x) < __builtin_inff()) || !(fabsf(v.y) < __builtin_inff()) || !(l1 < __builtin_inff());
      }
      bad |= !(b < __builtin_inff());
      m = fmaxf(m, b);
    }
  if (bad) atomicMax(&s_max[(int)(i0 % n_levels)], 0x7FC00000u);   // above every finite pattern: the level is marked non-finite
  else if (m > 0.f) atomicMax(&s_max[(int)(i0 % n_levels)], __float_as_uint(m));
  __syncthreads();
  if (threadIdx.x < HG_MAX_LEVELS && s_max[threadIdx.x]) atomicMax(&lmax[threadIdx.x], s_max[threadIdx.x]);
}

// ---- count ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BIN_PTS)
    bin_count_kernel(int64_t B, HgLevels lv, BinPlan bp, BinStencil stn, const float *__restrict__ x, uint32_t *__restrict__ counts) {
  __shared__ uint32_t s_hist[BIN_MAX_LOCAL];
  const int grp = blockIdx.x % bp.n_groups;
  const int64_t chunk = blockIdx.x / bp.n_groups;
  const int l0 = grp * BIN_G, l1 = min(l0 + BIN_G, lv.n_levels);
  const int b0 = bp.tile_base[l0], nloc = bp.tile_base[l1] - b0;
  for (int i = threadIdx.x; i < nloc; i += BIN_PTS) s_hist[i] = 0;
  __syncthreads();
  // 4 x 256 points per workgroup: fewer global atomics per bucket
  for (int rep = 0; rep < 4; ++rep) {
    const int64_t b = (chunk * 4 + rep) * BIN_PTS + threadIdx.x;
    if (b < B) {
      const float px = x[3 * b], py = x[3 * b + 1], pz = x[3 * b + 2];
      for (int level = l0; level < l1; ++level) {
        if (stn.n > 0 && level < stn.merge_levels && b >= stn.n && merged_into_base(lv, level, stn, x, b, px, py, pz)) continue;
        Corner8 c;
        corners_of(lv, level, px, py, pz, c);
        const int lb = bp.tile_base[level] - b0;
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&s_hist[lb + (int)(c.idx[k] >> BIN_TILE_LOG2)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nloc; i += BIN_PTS)
    if (s_hist[i]) atomicAdd(&counts[b0 + i], s_hist[i]);
}

// ---- plan: one workgroup ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
    bin_plan_kernel(int nb, int64_t item_records, const uint32_t *__restrict__ counts, uint32_t *__restrict__ cursor,
                    int64_t *__restrict__ start, BinItem *__restrict__ items, uint32_t *__restrict__ n_items, BinSlot *__restrict__ slots) {
  __shared__ int64_t s_rec[1024];
  __shared__ int32_t s_it[1024], s_sh[1024], s_sl[1024];
  __shared__ int64_t s_rec_base;
  __shared__ int32_t s_it_base, s_sh_base, s_sl_base;
  if (threadIdx.x == 0) { s_rec_base = 0; s_it_base = 0; s_sh_base = 0; s_sl_base = 0; }
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int b = base + threadIdx.x;
    const int64_t cnt = b < nb ? (int64_t)counts[b] : 0;
    const int32_t nit = cnt > 0 ? (int32_t)((cnt + item_records - 1) / item_records) : 0;
    const int32_t sh = nit > 1 ? 1 : 0;
    s_rec[threadIdx.x] = cnt;
    s_it[threadIdx.x] = nit;
    s_sh[threadIdx.x] = sh;
    s_sl[threadIdx.x] = sh ? nit : 0;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {  // inclusive Hillis-Steele scan
      const int64_t a = threadIdx.x >= s ? s_rec[threadIdx.x - s] : 0;
      const int32_t c = threadIdx.x >= s ? s_it[threadIdx.x - s] : 0;
      const int32_t d = threadIdx.x >= s ? s_sh[threadIdx.x - s] : 0;
      const int32_t f = threadIdx.x >= s ? s_sl[threadIdx.x - s] : 0;
      __syncthreads();
      s_rec[threadIdx.x] += a;
      s_it[threadIdx.x] += c;
      s_sh[threadIdx.x] += d;
      s_sl[threadIdx.x] += f;
      __syncthreads();
    }
    const int64_t rec0 = s_rec_base + s_rec[threadIdx.x] - cnt;
    const int32_t it0 = s_it_base + s_it[threadIdx.x] - nit;
    const int32_t slot0 = s_sh_base + s_sh[threadIdx.x] - sh;
    const int32_t slab0 = s_sl_base + s_sl[threadIdx.x] - (sh ? nit : 0);
    if (b < nb) {
      start[b] = rec0;
      cursor[b] = 0;
      int32_t shift = 0;
      while (shift < 40 && (cnt >> (BIN_SLOT_LOG2_RECORDS + shift)) > 0) ++shift;
#ifdef GSDF_BIN_NO_SLOTS
      const bool slabbed = false;
#else
      const bool slabbed = sh && slot0 < BIN_SLOTS && slab0 + nit <= BIN_SLABS;
#endif
      for (int32_t i = 0; i < nit; ++i) {
        BinItem it;
        it.begin = rec0 + (int64_t)i * item_records;
        it.end = min(rec0 + cnt, it.begin + item_records);
        it.bucket = b;
        it.nit = nit;
        it.slab = slabbed ? slab0 + i : -1;
        it.shift = shift;
        items[it0 + i] = it;
      }
      if (sh && slot0 < BIN_SLOTS) {
        BinSlot sl;
        sl.bucket = b; sl.slab0 = slab0; sl.nit = slabbed ? nit : 0; sl.shift = shift;
        slots[slot0] = sl;
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) { s_rec_base += s_rec[1023]; s_it_base += s_it[1023]; s_sh_base += s_sh[1023]; s_sl_base += s_sl[1023]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { start[nb] = s_rec_base; n_items[0] = (uint32_t)s_it_base; n_items[1] = (uint32_t)min(s_sh_base, BIN_SLOTS); }
}

// ---- emit -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const uint32_t o = __shfl_up(v, s, 64);
    if (lane >= s) v += o;
  }
  return v;
}

// one thread per (point, level of the group); PTS points x G levels per workgroup = PTS * G * 8 records staged in LDS.
// The records of a workgroup leave as one run per bucket: the fewer buckets a workgroup addresses and the more points it
// holds, the longer the runs (PTS * G * 8 / (G * 128 tiles) records of 8 bytes)
// Round 6, measured and NOT adopted (GPW stays 1; the template parameter is kept for the experiment): a workgroup that walks GPW level groups of ITS
// points one after the other.  A point's feature gradients are a 128-byte row of which one level group uses 16 bytes: with one group per workgroup
// every row is fetched by eight workgroups (FETCH_SIZE 0.48 GB for 0.12 GB of inputs, as much as the records the pass writes).  GPW = 4 (a 64-byte
// half row per point, prefetched into registers) removes those fetches and changes nothing: 230 us either way at 0.44 M points (GPW 1 / 2 / 4 / 8:
// 231 / 241 / 229 / 229 us), 4 % slower at 3 M (97 registers instead of 68) — the re-fetches are L2 hits, the pass is bound by its 8-byte-record
// runs of 16 (128 B) landing in 1900 buckets.
template <int PTS, int G, int GPW>
__global__ void __launch_bounds__(PTS * G)
    bin_emit_kernel(int64_t B, HgLevels lv, BinPlan bp, BinStencil stn, const float *__restrict__ x,
                    const float *__restrict__ v_feat, const int64_t *__restrict__ start, uint32_t *__restrict__ cursor,
                    BinRecord *__restrict__ records, const float *__restrict__ v_feat2, const float *__restrict__ vv_x,
                    const uint32_t *__restrict__ lmax) {
  constexpr int BIN_EMIT_THREADS = PTS * G, BIN_REC = PTS * G * 8;
  __shared__ BinRecord s_rec[BIN_REC];      // packed records at their sorted position
  __shared__ unsigned char s_bkt[BIN_REC];  // their local bucket
  __shared__ uint32_t s_hist[BIN_MAX_LOCAL], s_off[BIN_MAX_LOCAL], s_wtot[BIN_MAX_LOCAL / 64];
  __shared__ int64_t s_dst[BIN_MAX_LOCAL];
  const int n_groups = (lv.n_levels + G - 1) / G;
  const int n_wg_groups = (n_groups + GPW - 1) / GPW;
  const int grp0 = (int)(blockIdx.x % n_wg_groups) * GPW;
  const int64_t chunk = blockIdx.x / n_wg_groups;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // thread -> (point, level): waves 0..3 take the group's first level, waves 4..7 the second (the level is wave-uniform)
  const int64_t b = chunk * PTS + (t % PTS);
  const bool row = b < B;
  float px = 0.f, py = 0.f, pz = 0.f, vvx = 0.f, vvy = 0.f, vvz = 0.f;
  if (row) {
    px = x[3 * b]; py = x[3 * b + 1]; pz = x[3 * b + 2];
    if (v_feat2 != nullptr) { vvx = vv_x[3 * b]; vvy = vv_x[3 * b + 1]; vvz = vv_x[3 * b + 2]; }
  }
  float2 vf_all[GPW], v2_all[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int lvl = (grp0 + gi) * G + t / PTS;
    const bool ok = row && lvl < lv.n_levels && lvl < (grp0 + gi + 1) * G;
    vf_all[gi] = ok && v_feat != nullptr ? *reinterpret_cast<const float2 *>(v_feat + (b * lv.n_levels + lvl) * 2) : make_float2(0.f, 0.f);
    v2_all[gi] = ok && v_feat2 != nullptr ? *reinterpret_cast<const float2 *>(v_feat2 + (b * lv.n_levels + lvl) * 2) : make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int grp = grp0 + gi;
    if (grp >= n_groups) break;   // (workgroup-uniform)
    const int l0 = grp * G, l1 = min(l0 + G, lv.n_levels);
    const int b0 = bp.tile_base[l0], nloc = bp.tile_base[l1] - b0;
    if (t < BIN_MAX_LOCAL) s_hist[t] = 0;
    __syncthreads();
    const int level = l0 + t / PTS;
    // pass 1: this (point, level)'s 8 contributions in registers, slot within the workgroup's bucket run from an LDS counter
    uint32_t key[8], slot[8];
    float g0[8], g1[8];
    bool emits = false;
    if (row && level < l1) {
      const bool try_merge = stn.n > 0 && level < stn.merge_levels;
      if (!(try_merge && b >= stn.n && merged_into_base(lv, level, stn, x, b, px, py, pz))) {   // else: the base row carries it
        emits = true;
        Corner8 c;
        corners_of(lv, level, px, py, pz, c);
        const float2 vf = vf_all[gi];
#pragma unroll
        for (int k = 0; k < 8; ++k) { g0[k] = c.w[k] * vf.x; g1[k] = c.w[k] * vf.y; }
        if (v_feat2 != nullptr) {   // second-order term of the analytic eikonal regulariser (no stencil structure with it)
          const float2 v2 = v2_all[gi];
          float dw[8];
          corner_dweights(lv, level, px, py, pz, vvx, vvy, vvz, dw);
#pragma unroll
          for (int k = 0; k < 8; ++k) { g0[k] = fmaf(dw[k], v2.x, g0[k]); g1[k] = fmaf(dw[k], v2.y, g1[k]); }
        }
        if (try_merge && b < stn.n) {
          // base row: add the stencil rows that share this cell (same 8 entries, their own trilinear weights), in row order
          int32_t gb[3];
          cell_coords(lv, level, px, py, pz, gb);
          for (int k6 = 0; k6 < 6; ++k6) {
            const int64_t r = stn.n + (int64_t)k6 * stn.n + b;
            const float qx = x[3 * r], qy = x[3 * r + 1], qz = x[3 * r + 2];
            int32_t gs[3];
            cell_coords(lv, level, qx, qy, qz, gs);
            if (gs[0] != gb[0] || gs[1] != gb[1] || gs[2] != gb[2]) continue;
            Corner8 cs;
            corners_of(lv, level, qx, qy, qz, cs);
            const float2 vs = *reinterpret_cast<const float2 *>(v_feat + (r * lv.n_levels + level) * 2);
#pragma unroll
            for (int k = 0; k < 8; ++k) { g0[k] += cs.w[k] * vs.x; g1[k] += cs.w[k] * vs.y; }
          }
        }
        const int lb = bp.tile_base[level] - b0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t bucket = (uint32_t)lb + (c.idx[k] >> BIN_TILE_LOG2);
          key[k] = (c.idx[k] & (BIN_TILE - 1)) | (bucket << 16);
          slot[k] = atomicAdd(&s_hist[bucket], 1u);
        }
      }
    }
    __syncthreads();
    // reserve the runs in the global buckets (the reply is only needed by pass 3: its latency hides behind the scan and
    // pass 2); exclusive scan of the local histogram by the first 256 threads: wave scans + one combine
    uint32_t mine = 0u, reserved = 0u, incl = 0u;
    if (t < BIN_MAX_LOCAL) {
      mine = t < nloc ? s_hist[t] : 0u;
      if (mine) reserved = atomicAdd(&cursor[b0 + t], mine);
      incl = wave_inclusive_scan(mine, lane);
      if (lane == 63) s_wtot[wave] = incl;
    }
    __syncthreads();
    uint32_t n_rec = 0u;
#pragma unroll
    for (int w = 0; w < BIN_MAX_LOCAL / 64; ++w) n_rec += s_wtot[w];
    if (t < BIN_MAX_LOCAL) {
      uint32_t before = 0u;
#pragma unroll
      for (int w = 0; w < BIN_MAX_LOCAL / 64; ++w) before += w < wave ? s_wtot[w] : 0u;
      s_off[t] = before + incl - mine;
    }
    __syncthreads();
    // pass 2: packed records to their sorted position in LDS
    if (emits) {
      const uint32_t mxl = lmax[level];
      const int e_level = (int)(mxl >> 23) - 126 + BIN_EXP_HEADROOM;     // |g| < 2^e (a non-finite level is poisoned by the apply pass whatever is written here)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t bucket = key[k] >> 16;
        const uint32_t p = s_off[bucket] + slot[k];
        s_rec[p] = pack_record(key[k] & 0xFFFFu, g0[k], g1[k], e_level);
        s_bkt[p] = (unsigned char)bucket;
      }
    }
    // destination of sorted position p in bucket b: s_dst[b] + p  (s_dst already net of the bucket's first position in LDS)
    if (mine) s_dst[t] = start[b0 + t] + (int64_t)reserved - (int64_t)s_off[t];
    __syncthreads();
    // pass 3: runs to the global buckets; consecutive lanes write consecutive 8-byte records
    for (uint32_t p = t; p < n_rec; p += BIN_EMIT_THREADS) records[s_dst[s_bkt[p]] + (int64_t)p] = s_rec[p];
    __syncthreads();   // the next group reuses the staging arrays
  }
}

// ---- apply ------------------------------------------------------------------------------------------------------
static constexpr int BIN_FIX_BITS = 41;   // 2^41 * 2^ceil(log2 max) * 2^19 records per item < 2^62: no overflow
__global__ void __launch_bounds__(BIN_APPLY_THREADS)
    bin_apply_kernel(HgLevels lv, BinPlan bp, const BinItem *__restrict__ items, const uint32_t *__restrict__ n_items,
                     const uint32_t *__restrict__ lmax, const BinRecord *__restrict__ records, float *__restrict__ v_table,
                     long long *__restrict__ slabs) {
  __shared__ unsigned long long s_tile[2 * BIN_TILE];
  if (blockIdx.x >= *n_items) return;
  const BinItem it = items[blockIdx.x];
  // bucket -> (level, tile)
  int level = 0;
  while (level + 1 < lv.n_levels && bp.tile_base[level + 1] <= it.bucket) ++level;
  const uint32_t mx = lmax[level];
  if (mx == 0u) return;                            // every contribution of this level is zero
  if (mx >= 0x7F800000u) {
    // a non-finite contribution somewhere in this level (bin_vmax): no fixed-point scale exists.  The level's touched tiles are
    // poisoned with NaN so that the caller's NaN checks trip, as they would after the atomic kernel (which poisons only the entries the
    // offending point reaches)
    const int tile_p = it.bucket - bp.tile_base[level];
    const int64_t e0p = (int64_t)tile_p << BIN_TILE_LOG2;
    const int n_entp = (int)min((int64_t)BIN_TILE, (int64_t)lv.hsize[level] - e0p);
    float *dstp = v_table + ((int64_t)lv.offset[level] + e0p) * 2;
    for (int j = threadIdx.x; j < 2 * n_entp; j += BIN_APPLY_THREADS) dstp[j] = __builtin_nanf("");
    return;
  }
  const int e = (int)(mx >> 23) - 126 + BIN_EXP_HEADROOM;   // |g| < 2^e, the exponent pack_record used
  const double inv = ldexp(1.0, e - BIN_FIX_BITS);
  for (int i = threadIdx.x; i < 2 * BIN_TILE; i += BIN_APPLY_THREADS) s_tile[i] = 0ull;
  __syncthreads();
  const int64_t n = it.end - it.begin;
  const BinRecord *rec = records + it.begin;
  int64_t i = threadIdx.x;
  constexpr int U = GSDF_BIN_APPLY_INFLIGHT;
  for (; i + (U - 1) * BIN_APPLY_THREADS < n; i += U * BIN_APPLY_THREADS) {  // U loads in flight per lane
    BinRecord r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = rec[i + u * BIN_APPLY_THREADS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t key;
      long long f0, f1;
      unpack_record(r[u], key, f0, f1);
      atomicAdd(&s_tile[2 * key], (unsigned long long)f0);
      atomicAdd(&s_tile[2 * key + 1], (unsigned long long)f1);
    }
  }
  for (; i < n; i += BIN_APPLY_THREADS) {
    uint32_t key;
    long long f0, f1;
    unpack_record(rec[i], key, f0, f1);
    atomicAdd(&s_tile[2 * key], (unsigned long long)f0);
    atomicAdd(&s_tile[2 * key + 1], (unsigned long long)f1);
  }
  __syncthreads();
  const int tile = it.bucket - bp.tile_base[level];
  const int64_t e0 = (int64_t)tile << BIN_TILE_LOG2;
  const int n_ent = (int)min((int64_t)BIN_TILE, (int64_t)lv.hsize[level] - e0);
  float *dst = v_table + ((int64_t)lv.offset[level] + e0) * 2;
  auto val = [&](int j) { return (float)((double)(long long)s_tile[j] * inv); };
  if (it.nit <= 1) {
    // hsize is a multiple of 8 entries and level offsets are too: 16-byte aligned float4 read-modify-writes
    for (int j = threadIdx.x; j < n_ent / 2; j += BIN_APPLY_THREADS) {
      float4 d = reinterpret_cast<float4 *>(dst)[j];
      d.x += val(4 * j); d.y += val(4 * j + 1); d.z += val(4 * j + 2); d.w += val(4 * j + 3);
      reinterpret_cast<float4 *>(dst)[j] = d;
    }
  } else if (it.slab >= 0) {
    // a split bucket: this item's tile leaves as 64-bit integers; bin_reduce sums the bucket's slabs (integer sums: whatever the order, the same
    // bits) and adds them to the gradient — the split buckets are as bit-reproducible as the whole ones
    longlong2 *g = reinterpret_cast<longlong2 *>(slabs + (size_t)it.slab * (2 * BIN_TILE));
    const long long half = it.shift > 0 ? 1LL << (it.shift - 1) : 0;
    for (int j = threadIdx.x; j < n_ent; j += BIN_APPLY_THREADS) {
      longlong2 v;
      v.x = ((long long)s_tile[2 * j] + half) >> it.shift;
      v.y = ((long long)s_tile[2 * j + 1] + half) >> it.shift;
      g[j] = v;
    }
  } else {
    for (int j = threadIdx.x; j < 2 * n_ent; j += BIN_APPLY_THREADS) {
      const float s = val(j);
      if (s != 0.f) atomicAdd(dst + j, s);
    }
  }
}

// ---- reduce: the slabs of the split buckets ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    bin_reduce_kernel(HgLevels lv, BinPlan bp, const BinSlot *__restrict__ slots, const uint32_t *__restrict__ n_items,
                      const uint32_t *__restrict__ lmax, const long long *__restrict__ slabs, float *__restrict__ v_table) {
  const int slot = blockIdx.x / BIN_SEGS, seg = blockIdx.x % BIN_SEGS;
  if (slot >= (int)n_items[1]) return;
  const BinSlot sl = slots[slot];
  if (sl.nit == 0) return;
  int level = 0;
  while (level + 1 < lv.n_levels && bp.tile_base[level + 1] <= sl.bucket) ++level;
  const uint32_t mx = lmax[level];
  if (mx == 0u || mx >= 0x7F800000u) return;       // nothing arrived / the apply pass poisoned the level's tiles
  const int e = (int)(mx >> 23) - 126 + BIN_EXP_HEADROOM;
  const double inv_s = ldexp(1.0, e - BIN_FIX_BITS + sl.shift);
  const int tile = sl.bucket - bp.tile_base[level];
  const int64_t e0 = (int64_t)tile << BIN_TILE_LOG2;
  const int n_ent = (int)min((int64_t)BIN_TILE, (int64_t)lv.hsize[level] - e0);
  float *dst = v_table + ((int64_t)lv.offset[level] + e0) * 2;
  // float4 j = entries 2 j, 2 j + 1; a segment = n_ent / 2 / BIN_SEGS of them (n_ent is a multiple of 8)
  constexpr int PER = BIN_TILE / 2 / BIN_SEGS;
  for (int j = seg * PER + threadIdx.x; j < min((seg + 1) * PER, n_ent / 2); j += 256) {
    long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < sl.nit; ++i) {
      const longlong2 *g = reinterpret_cast<const longlong2 *>(slabs + (size_t)(sl.slab0 + i) * (2 * BIN_TILE));
      const longlong2 a = g[2 * j], b = g[2 * j + 1];
      a0 += a.x; a1 += a.y; a2 += b.x; a3 += b.y;
    }
    float4 d = reinterpret_cast<float4 *>(dst)[j];
    d.x += (float)((double)a0 * inv_s); d.y += (float)((double)a1 * inv_s);
    d.z += (float)((double)a2 * inv_s); d.w += (float)((double)a3 * inv_s);
    reinterpret_cast<float4 *>(dst)[j] = d;
  }
}

}  // namespace gsdf

using namespace gsdf;

static int binned_setup(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                        HgLevels *lv, BinPlan *bp) {
  if (n_levels < 1 || n_levels > HG_MAX_LEVELS || n_feat != 2 || log2_hashmap < 3 || log2_hashmap > 30 || base_res < 1 ||
      per_level_scale < 1.0f || B < 0)
    return -1;
  build_levels(n_levels, log2_hashmap, base_res, per_level_scale, lv, nullptr);
  return make_plan(*lv, bp) ? 0 : -2;
}

extern "C" size_t gsdf_hashgrid_bwd_binned_ws_bytes(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                                    float per_level_scale) {
  HgLevels lv;
  BinPlan bp;
  if (binned_setup(B, n_levels, n_feat, log2_hashmap, base_res, per_level_scale, &lv, &bp) != 0) return 0;
  return carve(nullptr, B, n_levels, bp.tile_base[n_levels]).bytes;
}

extern "C" int gsdf_hashgrid_bwd_binned(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res,
                                        float per_level_scale, const float *x, const float *v_feat, float *v_table,
                                        void *ws, size_t ws_bytes, gsdf_stream_t stream) {
  return gsdf_hashgrid_bwd_binned_stencil(B, 0, 0, n_levels, n_feat, log2_hashmap, base_res, per_level_scale, x, v_feat, v_table, ws,
                                          ws_bytes, stream);
}

static int binned_scatter(int64_t B, int64_t stencil_n, int merge_levels, int n_levels, int n_feat, int log2_hashmap, int base_res,
                          float per_level_scale, const float *x, const float *v_feat, const float *v_feat2, const float *vv_x,
                          float *v_table, void *ws, size_t ws_bytes, hipStream_t stream);

extern "C" int gsdf_hashgrid_bwd_binned_stencil(int64_t B, int64_t stencil_n, int merge_levels, int n_levels, int n_feat,
                                                int log2_hashmap, int base_res, float per_level_scale, const float *x,
                                                const float *v_feat, float *v_table, void *ws, size_t ws_bytes,
                                                gsdf_stream_t stream_) {
  GSDF_REQUIRE(B == 0 || v_feat, "hashgrid_bwd_binned: null v_feat");
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd_binned_stencil");
  return binned_scatter(B, stencil_n, merge_levels, n_levels, n_feat, log2_hashmap, base_res, per_level_scale, x, v_feat, nullptr, nullptr,
                        v_table, ws, ws_bytes, (hipStream_t)stream_);
}

extern "C" int gsdf_hashgrid_bwd_binned2(int64_t B, int n_levels, int n_feat, int log2_hashmap, int base_res, float per_level_scale,
                                         const float *x, const float *v_feat, const float *v_feat2, const float *vv_x, float *v_table,
                                         void *ws, size_t ws_bytes, gsdf_stream_t stream_) {
  GSDF_REQUIRE(B == 0 || (v_feat2 && vv_x), "hashgrid_bwd_binned2: null second-order inputs");
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_hashgrid_bwd_binned2");
  return binned_scatter(B, 0, 0, n_levels, n_feat, log2_hashmap, base_res, per_level_scale, x, v_feat, v_feat2, vv_x, v_table, ws, ws_bytes,
                        (hipStream_t)stream_);
}

static int binned_scatter(int64_t B, int64_t stencil_n, int merge_levels, int n_levels, int n_feat, int log2_hashmap, int base_res,
                          float per_level_scale, const float *x, const float *v_feat, const float *v_feat2, const float *vv_x,
                          float *v_table, void *ws, size_t ws_bytes, hipStream_t stream) {
  GSDF_REQUIRE(stencil_n == 0 || (stencil_n > 0 && B == 7 * stencil_n), "hashgrid_bwd_binned: a stencil batch has 7 * stencil_n rows");
  const BinStencil stn{stencil_n, stencil_n > 0 ? (merge_levels < 0 ? 0 : merge_levels) : 0};
  HgLevels lv;
  BinPlan bp;
  const int rc = binned_setup(B, n_levels, n_feat, log2_hashmap, base_res, per_level_scale, &lv, &bp);
  GSDF_REQUIRE(rc != -1, "hashgrid_bwd_binned: bad grid configuration");
  GSDF_REQUIRE(rc == 0, "hashgrid_bwd_binned: table too large for the binned scatter (use gsdf_hashgrid_bwd)");
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(x && (v_feat || v_feat2) && v_table && ws, "hashgrid_bwd_binned: null buffer");
  GSDF_REQUIRE(((uintptr_t)v_table & 15) == 0 && ((uintptr_t)ws & 255) == 0, "hashgrid_bwd_binned: v_table must be 16-byte and ws 256-byte aligned");
  const int nb = bp.tile_base[n_levels];
  const BinWs w = carve(ws, B, n_levels, nb);
  GSDF_REQUIRE(ws_bytes >= w.bytes, "hashgrid_bwd_binned: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
  GSDF_HIP(hipMemsetAsync(w.counts, 0, sizeof(uint32_t) * (nb + HG_MAX_LEVELS), stream), "hashgrid_bwd_binned memset");
  const int64_t chunks4 = (B + 4 * BIN_PTS - 1) / (4 * BIN_PTS), chunks = (B + BIN_PTS - 1) / BIN_PTS;
  GSDF_REQUIRE(chunks * bp.n_groups < (int64_t)1 << 31, "hashgrid_bwd_binned: batch too large");
  bin_vmax_kernel<<<1024, 256, 0, stream>>>(B * n_levels, n_levels, reinterpret_cast<const float2 *>(v_feat), w.lmax,
                                            reinterpret_cast<const float2 *>(v_feat2), vv_x, lv);
  GSDF_CHECK_LAUNCH("bin_vmax_kernel");
  bin_count_kernel<<<(unsigned)(chunks4 * bp.n_groups), BIN_PTS, 0, stream>>>(B, lv, bp, stn, x, w.counts);
  GSDF_CHECK_LAUNCH("bin_count_kernel");
  bin_plan_kernel<<<1, 1024, 0, stream>>>(nb, w.item_records, w.counts, w.cursor, w.start, w.items, w.n_items, w.slots);
  GSDF_CHECK_LAUNCH("bin_plan_kernel");
  // (measured and rejected: records straight from registers to their slots without the LDS sort — 2.0 ms against 1.74 ms)
  // (measured and rejected, round 2: 8192-entry tiles — 2x longer runs, 128 KB apply tiles — 3.23 ms; records staged as one
  //  16-byte LDS entry instead of three 4-byte arrays — 64 KB per workgroup, 2 per CU — 3.35 ms; tools/ubench/store_runs.hip:
  //  12-byte records in runs of 16 are written at 3.1 TB/s, runs of 32 at 4.6, one stream at 5.3)
  // (measured and rejected: 512 points x 1 level and 1024 x 1 per workgroup, i.e. 2x / 4x longer runs per bucket: 3.15 and
  //  3.48 ms against 2.94 ms for the whole scatter at 3.3 M points — the run length is not what bounds the emit pass)
  bin_emit_kernel<BIN_PTS, BIN_G, BIN_GPW><<<(unsigned)(chunks * ((bp.n_groups + BIN_GPW - 1) / BIN_GPW)), BIN_PTS * BIN_G, 0, stream>>>(B, lv, bp, stn, x, v_feat, w.start, w.cursor, w.records, v_feat2, vv_x, w.lmax);
  GSDF_CHECK_LAUNCH("bin_emit_kernel");
  bin_apply_kernel<<<(unsigned)w.max_items, BIN_APPLY_THREADS, 0, stream>>>(lv, bp, w.items, w.n_items, w.lmax, w.records, v_table, w.slabs);
  GSDF_CHECK_LAUNCH("bin_apply_kernel");
  bin_reduce_kernel<<<BIN_SLOTS * BIN_SEGS, 256, 0, stream>>>(lv, bp, w.slots, w.n_items, w.lmax, w.slabs, v_table);
  GSDF_CHECK_LAUNCH("bin_reduce_kernel");
  return GSDF_OK;
}
