// micro-benchmark: does giving every XCD a table region of its own raise the fp32 atomic rate?
// (design input for the hash-grid scatter: "XCD k owns levels {2k,2k+1}" vs "every XCD hits every level")
// 4 lanes = 16 consecutive bytes per request, like hashgrid_bwd_kernel's (x-corner, feature) groups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }

// mode 0: all blocks share one region of `region_floats`
// mode 1: block b uses region (b % 8)          (= its XCD, blocks are dealt round-robin over the XCDs)
// mode 2: block b uses region ((b / 8) % 8)    (control: same number of regions, not aligned with the XCDs)
// op 0: atomicAdd, op 1: plain racy RMW (upper bound of the memory path), op 2: plain store
template <int OP>
__global__ void k(float *buf, uint32_t region_floats, int mode, int per_thread) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t grp = t >> 2, sub = t & 3;
  const uint32_t region = mode == 0 ? 0u : mode == 1 ? (blockIdx.x & 7u) : ((blockIdx.x >> 3) & 7u);
  float *base = buf + (size_t)region * region_floats;
  const uint32_t mask = region_floats - 1;
  for (int i = 0; i < per_thread; ++i) {
    const uint32_t idx = ((mix(grp * 977u + i * 131071u) & mask) & ~3u) + sub;
    if (OP == 0) atomicAdd(base + idx, 1.0f);
    else if (OP == 1) base[idx] += 1.0f;
    else base[idx] = 1.0f;
  }
}

int main() {
  const size_t total = size_t(256) << 20;
  float *buf;
  hipMalloc(&buf, total);
  hipMemset(buf, 0, total);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256 * 16, threads = 256, per = 32;
  const char *opn[] = {"atomic", "plainRMW", "store"};
  const char *moden[] = {"shared", "per-XCD", "per-8blk(ctl)"};
  for (int op = 0; op < 3; ++op)
    for (size_t region_bytes : {size_t(256) << 10, size_t(2) << 20, size_t(8) << 20, size_t(32) << 20})
      for (int mode = 0; mode < 3; ++mode) {
        const uint32_t rf = (uint32_t)(region_bytes / 4);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(a);
          if (op == 0) k<0><<<blocks, threads>>>(buf, rf, mode, per);
          else if (op == 1) k<1><<<blocks, threads>>>(buf, rf, mode, per);
          else k<2><<<blocks, threads>>>(buf, rf, mode, per);
          hipEventRecord(b);
          hipEventSynchronize(b);
          hipEventElapsedTime(&ms, a, b);
        }
        const double req = double(blocks) * threads * per / 4;  // 16-byte requests
        printf("%-8s region=%6zu KB x%d  %-14s : %.3f ms  %.1f G 16B-requests/s\n", opn[op], region_bytes >> 10,
               mode == 0 ? 1 : 8, moden[mode], ms, req / ms / 1e6);
      }
  return 0;
}
