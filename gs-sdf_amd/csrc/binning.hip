// binning.hip — P3: tile binning (count -> scan -> emit -> sort -> offsets).
// Replaces gsplat_cpp::tile_encode (call site
// /root/reference/include/neural_gaussian/neural_gaussian.cpp:207-209).  SPEC A.3:
//   key = cam << (32+tile_bits) | tile_id << 32 | fp32 depth bits, value = packed splat index,
//   stable sort on the used key bits, isect_offsets[c,ty,tx] = first sorted slot of that tile.
// All outputs are integers and are the bit-exact parity target; the fp32 tile-rectangle math is
// compiled with -ffp-contract=off (only exact operations: /16, floor, ceil, clamp).
//
// The sort is hand-written (radix.hip) and exploits the key's structure instead of sorting 46-bit keys of all I
// intersections: (1) the M visible splats are sorted by their 32 depth bits ONCE (3 stable 11-bit passes over M elements,
// M ~ I/3), (2) the intersections are emitted in that order, (3) two stable 7-bit passes over the I (tile, splat) pairs by
// (camera, tile) index finish the job: within a tile the entries keep the depth order, ties in emission order (= increasing
// packed row), exactly the stable sort of SPEC A.3.  ~2.7x fewer bytes than six passes over 12-byte (key, value) pairs.
// (Round 1 sorted the 64-bit keys with rocPRIM; no library sort is left on any path.)
#include <cstring>

#include "radix.h"
#include "scan.h"

namespace gsdf {

static constexpr int BT = 256;

__device__ __forceinline__ void tile_rect(float mx, float my, int32_t radius, int tile_size, int tw, int th, int &x0,
                                          int &y0, int &x1, int &y1) {
  const float r = (float)radius / (float)tile_size;
  const float tx = mx / (float)tile_size, ty = my / (float)tile_size;
  x0 = (int)fminf(fmaxf(floorf(tx - r), 0.f), (float)tw);
  x1 = (int)fminf(fmaxf(ceilf(tx + r), 0.f), (float)tw);
  y0 = (int)fminf(fmaxf(floorf(ty - r), 0.f), (float)th);
  y1 = (int)fminf(fmaxf(ceilf(ty + r), 0.f), (float)th);
}

__global__ void __launch_bounds__(BT) tile_count_kernel(int64_t M, int tile_size, int tw, int th,
                                                        const float *__restrict__ means2d,
                                                        const int32_t *__restrict__ radii,
                                                        int32_t *__restrict__ tiles_per_gauss) {
  const int64_t m = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (m >= M) return;
  int32_t cnt = 0;
  const int32_t r = radii[m];
  if (r > 0) {
    const float2 xy = *reinterpret_cast<const float2 *>(means2d + 2 * m);
    int x0, y0, x1, y1;
    tile_rect(xy.x, xy.y, r, tile_size, tw, th, x0, y0, x1, y1);
    cnt = (y1 - y0) * (x1 - x0);
  }
  tiles_per_gauss[m] = cnt;
}

// ---- depth-first pipeline ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BT)
    tile_emit_sorted_kernel(int64_t M, int tile_size, int tw, int th, int64_t n_tiles, const uint32_t *__restrict__ order,
                            const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                            const int64_t *__restrict__ camera_ids, const int64_t *__restrict__ cum_sorted,
                            uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const int64_t j = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (j >= M) return;
  const int64_t m = order[j];
  const int32_t r = radii[m];
  if (r <= 0) return;
  const float2 xy = *reinterpret_cast<const float2 *>(means2d + 2 * m);
  int x0, y0, x1, y1;
  tile_rect(xy.x, xy.y, r, tile_size, tw, th, x0, y0, x1, y1);
  int64_t pos = (j == 0) ? 0 : cum_sorted[j - 1];
  const uint32_t cbase = (uint32_t)(camera_ids[m] * n_tiles);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      keys[pos] = cbase + (uint32_t)(y * tw + x);
      vals[pos] = (uint32_t)m;
      ++pos;
    }
}

__global__ void __launch_bounds__(BT) tile_offsets_kernel(int64_t I, int64_t n_tiles, int64_t total_tiles, int tile_bits,
                                                          const uint64_t *__restrict__ keys,
                                                          int32_t *__restrict__ offsets) {
  const int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x;
  if (i >= I) return;
  const uint64_t tmask = (1ull << tile_bits) - 1ull;
  const uint64_t k = keys[i] >> 32;
  const int64_t cur = (int64_t)(k >> tile_bits) * n_tiles + (int64_t)(k & tmask);
  if (i == 0) {
    for (int64_t t = 0; t <= cur; ++t) offsets[t] = 0;
  } else {
    const uint64_t kp = keys[i - 1] >> 32;
    const int64_t prev = (int64_t)(kp >> tile_bits) * n_tiles + (int64_t)(kp & tmask);
    for (int64_t t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;
  }
  if (i == I - 1)
    for (int64_t t = cur + 1; t < total_tiles; ++t) offsets[t] = (int32_t)I;
}

static inline int bits_for(int64_t n) {  // floor(log2(n)) + 1
  int b = 0;
  while ((1LL << (b + 1)) <= n) ++b;
  return b + 1;
}

}  // namespace gsdf

using namespace gsdf;

extern "C" size_t gsdf_tile_count_ws_bytes(int64_t M) { return scan_ws_bytes(M) + 256; }

extern "C" int gsdf_tile_count(int64_t M, int width, int height, int tile_size, const float *means2d,
                               const int32_t *radii, int32_t *tiles_per_gauss, int64_t *cum_tiles, void *ws,
                               int64_t *n_isects, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_tile_count");
  GSDF_REQUIRE(tile_size > 0 && width > 0 && height > 0, "tile_count: bad geometry");
  GSDF_REQUIRE(n_isects && ws, "tile_count: null workspace / n_isects");
  const int tw = (width + tile_size - 1) / tile_size, th = (height + tile_size - 1) / tile_size;
  if (M > 0) {
    GSDF_REQUIRE(means2d && radii && tiles_per_gauss && cum_tiles, "tile_count: null buffer");
    tile_count_kernel<<<(unsigned)((M + BT - 1) / BT), BT, 0, stream>>>(M, tile_size, tw, th, means2d, radii,
                                                                        tiles_per_gauss);
    GSDF_CHECK_LAUNCH("tile_count_kernel");
  }
  return scan_inclusive_i32_i64(tiles_per_gauss, cum_tiles, M, ws, n_isects, stream);
}

// workspace of the depth-first pipeline
struct BinWs2 {
  uint32_t *kA, *vA, *kB, *vB, *kC, *vC, *kD, *vD, *hist;
  int32_t *cnt;
  int64_t *cum, *total;
  void *scan_ws;
  size_t bytes;
};
static BinWs2 carve2(void *ws, int64_t M, int64_t I) {
  BinWs2 w;
  char *p = (char *)ws;
  size_t o = 0;
  auto take = [&](size_t n) { char *q = p ? p + o : nullptr; o = align_up(o + (n ? n : 4), 256); return q; };
  const size_t m4 = (size_t)(M > 0 ? M : 1) * 4, i4 = (size_t)(I > 0 ? I : 1) * 4;
  w.kA = (uint32_t *)take(m4); w.vA = (uint32_t *)take(m4); w.kB = (uint32_t *)take(m4); w.vB = (uint32_t *)take(m4);
  w.cnt = (int32_t *)take(m4); w.cum = (int64_t *)take(2 * m4); w.total = (int64_t *)take(256);
  w.scan_ws = take(scan_ws_bytes(M));
  w.kC = (uint32_t *)take(i4); w.vC = (uint32_t *)take(i4); w.kD = (uint32_t *)take(i4); w.vD = (uint32_t *)take(i4);
  w.hist = (uint32_t *)take(radix_ws_bytes(M > I ? M : I));
  w.bytes = o;
  return w;
}

extern "C" size_t gsdf_tile_encode_ws_bytes(int64_t M, int64_t I) {
  if (I <= 0) return 256;
  return carve2(nullptr, M, I).bytes;
}

extern "C" int gsdf_tile_encode(int64_t M, int64_t C, int64_t I, int width, int height, int tile_size,
                                const float *means2d, const int32_t *radii, const float *depths,
                                const int64_t *camera_ids, const int64_t *cum_tiles, void *ws, int64_t *isect_ids,
                                int32_t *flatten_ids, int32_t *isect_offsets, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_tile_encode");
  GSDF_REQUIRE(tile_size > 0 && width > 0 && height > 0 && C >= 1, "tile_encode: bad geometry");
  GSDF_REQUIRE(isect_offsets, "tile_encode: null isect_offsets");
  const int tw = (width + tile_size - 1) / tile_size, th = (height + tile_size - 1) / tile_size;
  const int64_t n_tiles = (int64_t)tw * th, total_tiles = n_tiles * C;
  GSDF_REQUIRE(I < (1LL << 31), "tile_encode: %ld intersections overflow int32 offsets", (long)I);
  if (I <= 0 || M <= 0) {
    GSDF_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)total_tiles * 4, stream), "tile_encode memset");
    return GSDF_OK;
  }
  GSDF_REQUIRE(ws && isect_ids && flatten_ids && means2d && radii && depths && camera_ids && cum_tiles,
               "tile_encode: null buffer");
  const int tile_bits = bits_for(n_tiles), cam_bits = bits_for(C);
  GSDF_REQUIRE(32 + tile_bits + cam_bits <= 64, "tile_encode: key needs %d bits", 32 + tile_bits + cam_bits);
  GSDF_REQUIRE(total_tiles < (1LL << 32) && M < (1LL << 32), "tile_encode: more than 2^32 tiles or visible splats");
  {
    const BinWs2 w = carve2(ws, M, I);
    const unsigned gm = (unsigned)((M + BT - 1) / BT);
    // (1) the M rows by the raw fp32 depth bits: 3 stable 11-bit passes (bits 0-10, 11-21, 22-32): depths -> B -> A -> B.  The
    // first pass reads the depths in place (values = row numbers), the last one also writes the rows' tile counts in sorted order.
    const uint32_t *dkeys = reinterpret_cast<const uint32_t *>(depths);
    RadixHooks h0{true, nullptr, nullptr, nullptr, nullptr, 1, 0}, h2{false, cum_tiles, w.cnt, nullptr, nullptr, 1, 0};
    int rc = radix_pass(M, 0, 11, dkeys, nullptr, w.kB, w.vB, w.hist, &h0, stream);
    if (!rc) rc = radix_pass(M, 11, 11, w.kB, w.vB, w.kA, w.vA, w.hist, nullptr, stream);
    if (!rc) rc = radix_pass(M, 22, 11, w.kA, w.vA, nullptr, w.vB, w.hist, &h2, stream);
    if (rc) return rc;
    const uint32_t *order = w.vB;
    // (2) emit the intersections in depth order
    rc = scan_inclusive_i32_i64(w.cnt, w.cum, M, w.scan_ws, w.total, stream);
    if (rc) return rc;
    tile_emit_sorted_kernel<<<gm, BT, 0, stream>>>(M, tile_size, tw, th, n_tiles, order, means2d, radii, camera_ids, w.cum, w.kC, w.vC);
    GSDF_CHECK_LAUNCH("tile_emit_sorted_kernel");
    // (3) stable passes over the I pairs by (camera, tile) index; the last one writes flatten_ids and the 64-bit keys
    const int ct_bits = bits_for(total_tiles > 1 ? total_tiles - 1 : 1);
    const int n_pass = (ct_bits + 7) / 8;
    const int db = ct_bits <= 6 * n_pass ? 6 : ct_bits <= 7 * n_pass ? 7 : 8;          // e.g. 13 bits -> 2 passes of 7
    RadixHooks hf{false, nullptr, nullptr, (uint64_t *)isect_ids, depths, n_tiles, tile_bits};
    uint32_t *ki = w.kC, *vi = w.vC, *ko = w.kD, *vo = w.vD;
    for (int pass = 0; pass < n_pass; ++pass) {
      const bool last = pass == n_pass - 1;
      rc = radix_pass(I, db * pass, db, ki, vi, last ? nullptr : ko, last ? (uint32_t *)flatten_ids : vo, w.hist, last ? &hf : nullptr, stream);
      if (rc) return rc;
      uint32_t *t = ki; ki = ko; ko = t;
      t = vi; vi = vo; vo = t;
    }
  }
  tile_offsets_kernel<<<(unsigned)((I + BT - 1) / BT), BT, 0, stream>>>(I, n_tiles, total_tiles, tile_bits,
                                                                        (const uint64_t *)isect_ids, isect_offsets);
  GSDF_CHECK_LAUNCH("tile_offsets_kernel");
  return GSDF_OK;
}
