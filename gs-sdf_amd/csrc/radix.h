// radix.h — hand-written stable LSD radix sort passes on (uint32 key, uint32 value) pairs (radix.hip): the building block of
// the depth sort in the tile binning (binning.hip).  One pass = histogram -> scan -> stable scatter (three launches, no
// spin-waiting between workgroups), 6..11-bit digits, wave64 ballot multi-split ranks.
#pragma once
#include "common.h"

namespace gsdf {

static constexpr int RADIX_MAX_BITS = 11;                               // digit widths up to 11 (2048 counters per wave in LDS)
static constexpr int RADIX_THREADS = 256;
// elements per thread: a workgroup owns 256 x items consecutive elements (radix_items(): 16 for long inputs, fewer for the inputs of
// the binning, where a pass is a few hundred workgroups that each walk their block in `items` serial rounds)
static constexpr int RADIX_MIN_ITEMS = 4;
int radix_items(int64_t n);
static inline int64_t radix_blocks(int64_t n, int items) { return (n + (int64_t)RADIX_THREADS * items - 1) / ((int64_t)RADIX_THREADS * items); }
// scratch of one pass: the per-(digit, block) histogram matrix + the digit totals (sized for the smallest block)
static inline size_t radix_ws_bytes(int64_t n) {
  const int64_t nb = radix_blocks(n, RADIX_MIN_ITEMS);
  return align_up(((size_t)(nb > 0 ? nb : 1) + 1) * ((size_t)4 << RADIX_MAX_BITS), 256);
}

// Optional work fused into a pass (all pointers may be null):
struct RadixHooks {
  // FIRST pass of the depth sort: keys are the fp32 bit patterns in keys_in, values are the element indices (vals_in unused)
  bool iota_values;
  // LAST pass of the depth sort: besides (key, value) also cnt_out[pos] = tiles of row `value` (from the row-order inclusive scan)
  const int64_t *cum_tiles;
  int32_t *cnt_out;
  // LAST pass of the tile sort: vals_out = flatten_ids and the 64-bit intersection key of SPEC A.3
  uint64_t *keys64;
  const float *depths;
  int64_t n_tiles;
  int tile_bits;
};

// One stable pass: elements ordered by the `bits`-wide digit (key >> shift) & (2^bits - 1), ties in input order.
// bits in {6, 7, 8, 11}.  keys_out may be null (keys not needed afterwards); outputs must not alias the inputs.
// `hist` = radix_ws_bytes(n) bytes of scratch.
int radix_pass(int64_t n, int shift, int bits, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
               uint32_t *hist, const RadixHooks *hooks, hipStream_t stream);

}  // namespace gsdf
