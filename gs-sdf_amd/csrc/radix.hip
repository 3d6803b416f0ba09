// radix.hip — hand-written stable LSD radix sort passes (radix.h) for the depth sort of the tile binning (P3, binning.hip;
// replaces the sort inside gsplat_cpp::tile_encode, call site /root/reference/include/neural_gaussian/neural_gaussian.cpp:207-209).
//
// wave64 design of a pass:
//   hist     a workgroup owns 4096 consecutive elements: digit counts in LDS -> one column of the digit-major matrix;
//   scan     one workgroup per digit: exclusive scan of the digit's row (its count in every block) + the digit's total;
//   scatter  the digits' own offsets (exclusive scan of the totals) are formed by every workgroup in LDS; each of the 4 waves owns
//            a contiguous quarter of the block, walked in 16 rounds of 64 (input order = (wave, round, lane)): phase 1 counts
//            the wave's digits (LDS atomics on its own row of counters), ONE barrier, then a wave ranks the 64 elements of a
//            round by digit with `bits` ballots (multi-split: the lanes holding the same digit form a mask, rank = popcount of
//            the mask below the lane) on top of a running slot per digit in the wave's own LDS row.  LDS operations of one wave
//            execute in order: no barrier in phase 2.
// No atomics on the output side, no sorting network, no spin-waiting between workgroups, bit-reproducible.
// Measured and rejected: single-kernel passes with a decoupled look-back (onesweep style): same time at the BASELINE shapes
// (0.376 against 0.367 ms for the whole binning at cfg3) and a pathological case — 5.8 ms at I = 7 M with 2048-element blocks,
// when more workgroups are in flight than fit the chip and the spinning ones starve their predecessors (rocPRIM's onesweep
// shows the same cliff at that shape: 6.2 ms).
#include "radix.h"

#include <stdlib.h>

namespace gsdf {

template <int BITS, int ITEMS>
__global__ void __launch_bounds__(RADIX_THREADS)
    radix_hist_kernel(int64_t n, int shift, const uint32_t *__restrict__ keys, uint32_t *__restrict__ hist, int64_t nblk) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t s_h[BINS];
  const int t = threadIdx.x;
  for (int d = t; d < BINS; d += RADIX_THREADS) s_h[d] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (RADIX_THREADS * ITEMS);
#pragma unroll 4
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t i = base + r * RADIX_THREADS + t;
    if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & (BINS - 1)], 1u);
  }
  __syncthreads();
  for (int d = t; d < BINS; d += RADIX_THREADS) hist[(int64_t)d * nblk + blockIdx.x] = s_h[d];      // digit-major: the scan order
}

// one workgroup per digit: exclusive scan of the digit's row (its count in every block, in block order) in place, row total
// to totals[digit]
__global__ void __launch_bounds__(RADIX_THREADS) radix_scan_kernel(int64_t nblk, uint32_t *__restrict__ hist, uint32_t *__restrict__ totals) {
  __shared__ uint32_t s_w[RADIX_THREADS / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t *row = hist + (int64_t)blockIdx.x * nblk;
  uint32_t carry = 0u;
  for (int64_t base = 0; base < nblk; base += RADIX_THREADS) {
    const int64_t i = base + t;
    const uint32_t v = i < nblk ? row[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t o = __shfl_up(incl, s, 64);
      if (lane >= s) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0u, tot = 0u;
#pragma unroll
    for (int w = 0; w < RADIX_THREADS / 64; ++w) { before += w < wave ? s_w[w] : 0u; tot += s_w[w]; }
    if (i < nblk) row[i] = carry + before + incl - v;
    carry += tot;
    __syncthreads();
  }
  if (t == 0) totals[blockIdx.x] = carry;
}

template <int BITS, int ITEMS>
__global__ void __launch_bounds__(RADIX_THREADS)
    radix_scatter_kernel(int64_t n, int shift, const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                         uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ hist,
                         const uint32_t *__restrict__ totals, int64_t nblk, RadixHooks hk) {
  constexpr int BINS = 1 << BITS;
  constexpr int RADIX_BLOCK = RADIX_THREADS * ITEMS, NW = RADIX_THREADS / 64, PER_WAVE = RADIX_BLOCK / NW, ROUNDS = PER_WAVE / 64;
  __shared__ uint32_t s_dig[BINS];                       // digit offset + this block's offset within the digit
  __shared__ uint32_t s_cnt[NW][BINS];                   // phase 1: counts; phase 2: running slots of the wave
  __shared__ uint32_t s_w[NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {  // digit offsets: block-wide exclusive scan of the digit totals (BINS / 256 consecutive digits per thread)
    constexpr int PT = BINS >= RADIX_THREADS ? BINS / RADIX_THREADS : 1;
    uint32_t v[PT], sum = 0u;
#pragma unroll
    for (int q = 0; q < PT; ++q) { const int d = t * PT + q; v[q] = d < BINS ? totals[d] : 0u; sum += v[q]; }
    uint32_t incl = sum;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t o = __shfl_up(incl, s, 64);
      if (lane >= s) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    for (int d = t; d < BINS; d += RADIX_THREADS)
#pragma unroll
      for (int w = 0; w < NW; ++w) s_cnt[w][d] = 0u;
    __syncthreads();
    uint32_t run = incl - sum;
#pragma unroll
    for (int w = 0; w < NW; ++w) run += w < wave ? s_w[w] : 0u;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int d = t * PT + q;
      if (d < BINS) s_dig[d] = run + hist[(int64_t)d * nblk + blockIdx.x];
      run += v[q];
    }
  }
  const int64_t wbase = (int64_t)blockIdx.x * RADIX_BLOCK + (int64_t)wave * PER_WAVE;
  uint32_t k[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    k[r] = i < n ? keys_in[i] : 0u;
    if (i < n) atomicAdd(&s_cnt[wave][(k[r] >> shift) & (BINS - 1)], 1u);
  }
  __syncthreads();
  for (int d = t; d < BINS; d += RADIX_THREADS) {  // running slots of digit d for every wave
    uint32_t run = s_dig[d];
#pragma unroll
    for (int w = 0; w < NW; ++w) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = run; run += c; }
  }
  __syncthreads();
  volatile uint32_t *slots = s_cnt[wave];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (k[r] >> shift) & (BINS - 1);
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(valid && bit);
      mask &= bit ? bal : ~bal;
    }
    const uint32_t rank = (uint32_t)__popcll(mask & lt);
    const uint32_t slot0 = slots[d];                                  // every lane of the digit reads the same word
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0u) slots[d] = slot0 + (uint32_t)__popcll(mask);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const uint32_t pos = slot0 + rank;
      const uint32_t v = hk.iota_values ? (uint32_t)i : vals_in[i];
      if (keys_out != nullptr) keys_out[pos] = k[r];
      vals_out[pos] = v;
      if (hk.cnt_out != nullptr) hk.cnt_out[pos] = (int32_t)(hk.cum_tiles[v] - (v == 0u ? 0 : hk.cum_tiles[v - 1]));
      if (hk.keys64 != nullptr) {
        const uint64_t cam = (uint64_t)k[r] / (uint64_t)hk.n_tiles, tile = (uint64_t)k[r] - cam * (uint64_t)hk.n_tiles;
        hk.keys64[pos] = (cam << (32 + hk.tile_bits)) | (tile << 32) | (uint64_t)__float_as_uint(hk.depths[v]);
      }
    }
  }
}

template <int BITS, int ITEMS>
static int radix_pass_t(int64_t n, int shift, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                        uint32_t *hist, const RadixHooks &hk, hipStream_t stream) {
  constexpr int BINS = 1 << BITS;
  const int64_t nblk = radix_blocks(n, ITEMS);
  radix_hist_kernel<BITS, ITEMS><<<(unsigned)nblk, RADIX_THREADS, 0, stream>>>(n, shift, keys_in, hist, nblk);
  GSDF_CHECK_LAUNCH("radix_hist_kernel");
  uint32_t *totals = hist + nblk * BINS;
  radix_scan_kernel<<<BINS, RADIX_THREADS, 0, stream>>>(nblk, hist, totals);
  GSDF_CHECK_LAUNCH("radix_scan_kernel");
  radix_scatter_kernel<BITS, ITEMS><<<(unsigned)nblk, RADIX_THREADS, 0, stream>>>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, totals, nblk, hk);
  GSDF_CHECK_LAUNCH("radix_scatter_kernel");
  return GSDF_OK;
}
template <int BITS>
static int radix_pass_b(int64_t n, int shift, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                        uint32_t *hist, const RadixHooks &hk, hipStream_t stream) {
  switch (radix_items(n)) {
    case 4: return radix_pass_t<BITS, 4>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
    case 8: return radix_pass_t<BITS, 8>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
    default: return radix_pass_t<BITS, 16>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
  }
}

// items per thread of a pass: by the input's length
int radix_items(int64_t n) {
  // measured on the binning (tools/exp_binning.py, whole tile_encode): 300 k splats / 0.64 M intersections 0.247 (16) 0.218 (8) 0.203 ms (4);
  // 1 M / 2.1 M: 0.357, 0.353, 0.366; 3 M / 7.1 M: 0.845, 0.881, 0.973 (and one 6.9 ms outlier at 4)
  return n < (1 << 19) ? 4 : n < 3 * (1 << 19) ? 8 : 16;
}

int radix_pass(int64_t n, int shift, int bits, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
               uint32_t *hist, const RadixHooks *hooks, hipStream_t stream) {
  if (n <= 0) return GSDF_OK;
  GSDF_REQUIRE(n < (1LL << 32), "radix: too many elements");
  const RadixHooks hk = hooks ? *hooks : RadixHooks{false, nullptr, nullptr, nullptr, nullptr, 1, 0};
  switch (bits) {
    case 6: return radix_pass_b<6>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
    case 7: return radix_pass_b<7>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
    case 8: return radix_pass_b<8>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
    case 11: return radix_pass_b<11>(n, shift, keys_in, vals_in, keys_out, vals_out, hist, hk, stream);
    default: set_error("radix_pass: digit width %d not instantiated", bits); return GSDF_ERR_INVALID_ARG;
  }
}

}  // namespace gsdf
