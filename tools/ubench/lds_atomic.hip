// micro-benchmark: LDS accumulation rate with random addresses in a 64 KB tile (design input for the binned
// hash-grid scatter's apply pass): ds_add_f32 vs ds_add_u32 vs plain ds_write, and the record-streaming loop
// (12-byte records from HBM) with each of them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
struct __attribute__((packed, aligned(4))) Rec { uint32_t key; float g0, g1; };

// MODE 5: ds_add_u64 x2 on a 4096-entry tile (fixed point via the 2^52 magic-number trick)
// MODE 0: ds_add_f32 x2, 1: ds_add_u32 x2, 2: plain store x2, 3: ds_add_f32 x1 (one float per record), 4: nothing (load only)
template <int MODE, bool STREAM>
__global__ void __launch_bounds__(512) k(const Rec *rec, int64_t per_block, float *out) {
  __shared__ float tile[16384];
  uint32_t *ti = reinterpret_cast<uint32_t *>(tile);
  for (int i = threadIdx.x; i < 16384; i += 512) tile[i] = 0.f;
  __syncthreads();
  const Rec *r = rec + (int64_t)blockIdx.x * per_block;
  float sink = 0.f;
  for (int64_t i = threadIdx.x; i + 3 * 512 < per_block; i += 4 * 512) {
    Rec q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (STREAM) q[u] = r[i + u * 512];
      else { q[u].key = mix((uint32_t)(i + u * 512) * 2654435761u + blockIdx.x) & 8191u; q[u].g0 = 1.f; q[u].g1 = 2.f; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) { atomicAdd(&tile[2 * q[u].key], q[u].g0); atomicAdd(&tile[2 * q[u].key + 1], q[u].g1); }
      else if (MODE == 1) { atomicAdd(&ti[2 * q[u].key], __float_as_uint(q[u].g0)); atomicAdd(&ti[2 * q[u].key + 1], __float_as_uint(q[u].g1)); }
      else if (MODE == 2) { tile[2 * q[u].key] = q[u].g0; tile[2 * q[u].key + 1] = q[u].g1; }
      else if (MODE == 5) {
        unsigned long long *t64 = reinterpret_cast<unsigned long long *>(tile);
        const double sc = 1099511627776.0;
        const long long a0 = __double_as_longlong(fma((double)q[u].g0, sc, 6755399441055744.0)) - 0x4338000000000000LL;
        const long long a1 = __double_as_longlong(fma((double)q[u].g1, sc, 6755399441055744.0)) - 0x4338000000000000LL;
        atomicAdd(&t64[2 * (q[u].key & 4095u)], (unsigned long long)a0); atomicAdd(&t64[2 * (q[u].key & 4095u) + 1], (unsigned long long)a1);
      }
      else if (MODE == 3) { atomicAdd(&tile[2 * q[u].key], q[u].g0 + q[u].g1); }
      else sink += q[u].g0 + q[u].g1 + __uint_as_float(q[u].key);
    }
  }
  __syncthreads();
  float s = sink;
  for (int i = threadIdx.x; i < 16384; i += 512) s += tile[i];
  if (s == 12345.678f) out[0] = s;
}

__global__ void fill(Rec *rec, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { rec[i].key = mix((uint32_t)i) & 8191u; rec[i].g0 = 1.f; rec[i].g1 = 0.5f; }
}

int main() {
  const int64_t per_block = 384 * 1024, blocks = 1024, n = per_block * blocks;   // 4.8 GB of records
  Rec *rec; float *out;
  hipMalloc(&rec, n * sizeof(Rec)); hipMalloc(&out, 4);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(rec, n);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char *names[] = {"ds_add_f32 x2", "ds_add_u32 x2", "ds_write x2", "ds_add_f32 x1", "load only", "ds_add_u64 x2"};
  for (int stream = 0; stream < 2; ++stream)
    for (int mode = 0; mode < 6; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
#define RUN(M) if (stream) k<M, true><<<blocks, 512>>>(rec, per_block, out); else k<M, false><<<blocks, 512>>>(rec, per_block, out);
        switch (mode) { case 0: RUN(0) break; case 1: RUN(1) break; case 2: RUN(2) break; case 3: RUN(3) break; case 5: RUN(5) break; default: RUN(4) }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      }
      printf("%-9s %-14s: %.3f ms  %.1f G records/s  %.2f TB/s of records\n", stream ? "streamed" : "synthetic", names[mode], ms,
             n / ms / 1e6, n * 12.0 / ms / 1e9);
    }
  return 0;
}
