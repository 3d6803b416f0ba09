// append_runs.hip — the emit pass of the binned scatter in isolation: every workgroup appends one run of R 8-byte records to each of the NB buckets it
// addresses (one atomic reservation per (workgroup, bucket) on the bucket's cursor, coalesced stores), runs unaligned as the reservations fall.
// Question (round 6): do the short runs cost what they cost because the two halves of a 128-byte line are written by workgroups on DIFFERENT XCDs
// (each L2 writes its partial line back on its own)?  Variants: one cursor per bucket (what the kernel does); one cursor per (bucket, XCD) with
// XCD = blockIdx % 8 (the halves of a line then meet in one L2); runs padded to whole 128-byte lines.
// Build: hipcc --offload-arch=gfx950 -O3 append_runs.hip -o append_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: shared cursors; 1: per-XCD cursors (bucket region split in 8 sub-regions); 2: shared cursors, runs of exactly 16 records (aligned);
// 3 / 4: random lengths padded with null records to whole 128-byte / 64-byte pieces (the reservations then fall aligned)
template <int MODE>
__global__ void __launch_bounds__(512) append(unsigned long long *rec, unsigned *cursor, long long cap_per_bucket, int nb_local, int nb_total, int R) {
  __shared__ long long s_dst[256];
  const int t = threadIdx.x;
  // a workgroup addresses nb_local consecutive buckets starting at a pseudo-random group (as one level pair of the grid does)
  const int g0 = (int)((blockIdx.x * 2654435761u) % (unsigned)(nb_total / nb_local)) * nb_local;
  const int xcd = blockIdx.x & 7;
  if (t < nb_local) {
    // run lengths R +- R/2 (pseudo-random): the reservations fall unaligned
    int len = MODE == 2 ? R : R / 2 + (int)(((blockIdx.x * 31u + t * 17u) * 2654435761u) >> 20) % (R + 1);
    if (MODE == 3) len = (len + 15) & ~15;
    if (MODE == 4) len = (len + 7) & ~7;
    long long base;
    if (MODE == 1) {
      const unsigned at = atomicAdd(&cursor[(g0 + t) * 8 + xcd], (unsigned)len);
      base = (long long)(g0 + t) * cap_per_bucket + (long long)xcd * (cap_per_bucket / 8) + at;
    } else {
      const unsigned at = atomicAdd(&cursor[(g0 + t) * 8], (unsigned)len);
      base = (long long)(g0 + t) * cap_per_bucket + at;
    }
    s_dst[t] = base | ((long long)len << 48);
  }
  __syncthreads();
  // 16 lanes per run, 32 runs per pass of the workgroup
  for (int r = t >> 4; r < nb_local; r += 32) {
    const long long d = s_dst[r];
    const int len = (int)(d >> 48);
    unsigned long long *p = rec + (d & 0xFFFFFFFFFFFFll);
    for (int i = t & 15; i < len; i += 16) p[i] = 0x123456789abcdefull + i;
  }
}

int main() {
  const int NB = 1920, NBL = 256, R = 16;
  const long long n_wg = 15000;                       // 0.44 M points x 8 level pairs / 256 points
  const long long cap = (n_wg * NBL / NB + 64) * (R * 3 / 2 + 24) * 2 / 8 * 8;
  unsigned long long *rec; unsigned *cur;
  CK(hipMalloc(&rec, (size_t)NB * cap * 8)); CK(hipMalloc(&cur, NB * 8 * 4));
  printf("%d buckets, %lld workgroups x %d runs of ~%d records: ~%.2f GB of records\n", NB, n_wg, NBL, R, n_wg * NBL * R * 8e-9);
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipMemset(cur, 0, NB * 8 * 4));
      CK(hipEventRecord(e0));
      if (mode == 0) append<0><<<(unsigned)n_wg, 512>>>(rec, cur, cap, NBL, NB, R);
      if (mode == 1) append<1><<<(unsigned)n_wg, 512>>>(rec, cur, cap, NBL, NB, R);
      if (mode == 2) append<2><<<(unsigned)n_wg, 512>>>(rec, cur, cap, NBL, NB, R);
      if (mode == 3) append<3><<<(unsigned)n_wg, 512>>>(rec, cur, cap, NBL, NB, R);
      if (mode == 4) append<4><<<(unsigned)n_wg, 512>>>(rec, cur, cap, NBL, NB, R);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const char *names[5] = {"one cursor per bucket", "one cursor per (bucket, XCD)", "one cursor per bucket, runs of exactly 16 (aligned)",
                            "random lengths padded to 16 records (128 B)", "random lengths padded to 8 records (64 B)"};
    printf("  %-52s %.3f ms (%.2f TB/s of payload)\n", names[mode], best, n_wg * NBL * R * 8.0 / (best * 1e-3) * 1e-12);
  }
  return 0;
}
