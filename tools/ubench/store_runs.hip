// store_runs.hip — HBM write rate of 12-byte records as a function of how they are laid down: one long stream, or runs of R
// records at scattered run offsets (what the binned scatter's emit pass produces: ~16 records per (workgroup, bucket) run).
// Also 16-byte records for comparison.  Build: hipcc --offload-arch=gfx950 -O3 store_runs.hip -o store_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct __attribute__((packed)) Rec12 { uint32_t k; float a, b; };

// n records; record i goes to run (i / R) placed at perm(run) * R: consecutive lanes write consecutive records of a run
template <typename T>
__global__ void __launch_bounds__(256) put(T *dst, long long n, int R, long long n_runs, long long mul) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long run = i / R, off = i - run * R;
    const long long where = (run * mul) % n_runs;   // mul odd and coprime with n_runs: a permutation of the runs
    T r;
    r.k = (uint32_t)i; r.a = 1.f; r.b = 2.f;
    dst[where * R + off] = r;
  }
}
struct Rec16 { uint32_t k; float a, b, pad; };

template <typename T>
static void run(void *buf, long long n, int R, const char *name) {
  const long long n_runs = n / R, mul = R >= n ? 1 : 1000003;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  put<T><<<8192, 256>>>((T *)buf, n_runs * R, R, n_runs, mul);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) put<T><<<8192, 256>>>((T *)buf, n_runs * R, R, n_runs, mul);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  %-10s runs of %8d records: %.3f ms per %.2f GB -> %.2f TB/s\n", name, R, ms / 3, n_runs * R * sizeof(T) * 1e-9, n_runs * R * sizeof(T) / (ms / 3 * 1e-3) * 1e-12);
}

int main() {
  const long long n = 400ll << 20;   // 400 Mi records (4.7 GiB of 12-byte records)
  void *buf; CK(hipMalloc(&buf, n * 16));
  for (int R : {1 << 24, 4096, 256, 64, 32, 16, 8}) run<Rec12>(buf, n, R, "12-byte");
  for (int R : {1 << 24, 64, 16}) run<Rec16>(buf, n, R, "16-byte");
  return 0;
}
