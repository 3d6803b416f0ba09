// common.h — shared device/host helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gsdf_hip.h"

namespace gsdf {

void set_error(const char *fmt, ...);

#define GSDF_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      gsdf::set_error(__VA_ARGS__);             \
      return GSDF_ERR_INVALID_ARG;              \
    }                                           \
  } while (0)

#define GSDF_CHECK_LAUNCH(name)                                                  \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      gsdf::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
      return GSDF_ERR_LAUNCH;                                                    \
    }                                                                            \
  } while (0)

#define GSDF_HIP(call, name)                                                     \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                      \
      gsdf::set_error("%s: %s", name, hipGetErrorString(e_));                    \
      return GSDF_ERR_LAUNCH;                                                    \
    }                                                                            \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ----------------------------------------------------------------------------------------------
// wave64 cross-lane primitives (DPP: no LDS traffic).  dpp_ctrl encodings: quad_perm 0x00-0xFF,
// row_shr:n 0x110+n, row_mirror 0x140, row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143.
// ----------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_CTRL = true>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL));
}

// Sum over the 64 lanes of the wave; the full sum is valid in lane 63 (and only there).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_mov<0x111>(v);                     // row_shr:1
  v += dpp_mov<0x112>(v);                     // row_shr:2
  v += dpp_mov<0x114>(v);                     // row_shr:4
  v += dpp_mov<0x118>(v);                     // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_mov<0x142, 0xA, 0xF, false>(v);    // row_bcast:15 into rows 1 and 3
  v += dpp_mov<0x143, 0xC, 0xF, false>(v);    // row_bcast:31 into rows 2 and 3
  return v;
}

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_CTRL = true>
__device__ __forceinline__ unsigned dpp_mov_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL);
}
// lane ^ 4 / lane ^ 8 exchange inside a DPP row: two masked row shifts
template <int CTRL_LO, int CTRL_HI, int BANK_LO, int BANK_HI>
__device__ __forceinline__ float dpp_xchg(float v) {
  // lanes of BANK_LO read with CTRL_LO (row_shl), lanes of BANK_HI with CTRL_HI (row_shr)
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL_LO, 0xF, BANK_LO, false);
  t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), CTRL_HI, 0xF, BANK_HI, false);
  return __int_as_float(t);
}

// Optional per-entry-point device timing (gsdf_timing_begin / gsdf_timing_end, include/gsdf_hip.h): a HIP event pair on the
// call's own stream around everything the entry point launches.  Off (one relaxed load) unless a caller switched it on.
bool timing_wants(const char *name);
void timing_push(const char *name, hipEvent_t a, hipEvent_t b);
struct TimedScope {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  const char *n;
  TimedScope(const char *name, hipStream_t st) : s(st), n(name) {
    if (timing_wants(name) && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, st);
    else a = nullptr;
  }
  ~TimedScope() {
    if (a != nullptr) { (void)hipEventRecord(b, s); timing_push(n, a, b); }
  }
};
#define GSDF_TIMED(name) gsdf::TimedScope timed_scope_(name, stream)

// Number of XCDs the queue behind `stream` may use: 8, or what the caller registered with gsdf_stream_set_xcds for a
// CU-masked stream (workgroups are dealt round-robin over the enabled XCDs only).  A locality hint for the XCD-aware
// kernels (tile bands of the compositing kernels, level groups of the hash-grid forward), never a correctness input.
int xcd_count(hipStream_t stream);
// gsdf_deterministic(): the process asked for order-independent accumulation (fixed point / ordered reductions) in the kernels it launches
bool deterministic();

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// A count the HOST may be polling (gsdf_host_words_alloc: pinned, device-mapped, fine-grained host memory): one system-scope store straight to the
// word.  (Round 4 wrote it with a plain store + __threadfence_system(): on gfx950 that fence writes the whole L2's dirty lines back — 8-65 us
// beside a kernel of the other leg — although the host reads nothing but the word.)
__device__ __forceinline__ void store_host_visible(int64_t *p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }


// ---- the last step of a one-value-per-workgroup reduction (loss values) -----------------------------------------------------------------------
// Thread 0 of every workgroup holds its workgroup's value(s) and the outputs ACCUMULATE.  Default: one float atomic per workgroup (their order varies
// from launch to launch: the value wanders in its last bits).  Deterministic mode (gsdf_deterministic; `slot` non-null): the values wait in a
// device-global slot of the kernel, the LAST workgroup to arrive (a ticket) sums them in index order — a fixed tree — and adds the totals with one
// atomic each; the ticket resets itself.  One slot per kernel: a kernel must not run on two streams at once in that mode.  Call with all 256
// threads of the workgroup.
static constexpr int DET_MAX_BLOCKS = 16384;
struct DetScalarSlot {
  float part[2][DET_MAX_BLOCKS];
  unsigned ticket;
};
#ifdef __HIPCC__
// (the ordered path is a call: its registers must not count against the occupancy targets of the kernels that end with it)
static __device__ __noinline__ void finish_scalars_ordered(float t0, float t1, float *o0, float *o1, DetScalarSlot *slot) {
  __shared__ bool s_last;
  __shared__ float s_red[2][4];
  if (threadIdx.x == 0) {
    __hip_atomic_store(&slot->part[0][blockIdx.x], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&slot->part[1][blockIdx.x], t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(&slot->ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float a = 0.f, b = 0.f;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) {
    a += __hip_atomic_load(&slot->part[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b += __hip_atomic_load(&slot->part[1][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  a = wave_sum_to_lane63(a); b = wave_sum_to_lane63(b);      // (a fixed butterfly: the same tree every launch)
  if ((threadIdx.x & 63) == 63) { s_red[0][threadIdx.x >> 6] = a; s_red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    slot->ticket = 0u;
    const float ta = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]), tb = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    if (ta != 0.f) atomicAdd(o0, ta);
    if (o1 != nullptr && tb != 0.f) atomicAdd(o1, tb);
  }
}
__device__ __forceinline__ void finish_scalars(float t0, float t1, float *__restrict__ o0, float *__restrict__ o1, DetScalarSlot *slot) {
  if (slot != nullptr) { finish_scalars_ordered(t0, t1, o0, o1, slot); return; }
  if (threadIdx.x == 0) {
    if (t0 != 0.f) atomicAdd(o0, t0);
    if (o1 != nullptr && t1 != 0.f) atomicAdd(o1, t1);
  }
}
#endif

}  // namespace gsdf
