// pk_fma.hip — is packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) a lever for the VALU-issue-bound compositing
// kernels?  N independent FMA chains per lane, once as scalar v_fma_f32 and once as v_pk_fma_f32 on float2 pairs, at full
// occupancy (8 waves / SIMD) and at 2 waves / SIMD; prints element-FMAs per second and the ratio.
// Build: hipcc --offload-arch=gfx950 -O3 pk_fma.hip -o pk_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool PACKED>
__global__ void __launch_bounds__(256) fma_loop(float *out, int iters, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
    if (PACKED) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        v2f v = {x[i], x[i + 1]};
        const v2f va = {a, a}, vb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(va), "v"(vb));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(a), "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool PACKED>
static double run(float *out, int blocks) {
  const int iters = 4000;
  fma_loop<PACKED><<<blocks, 256>>>(out, iters, 0.999f, 1e-3f);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  fma_loop<PACKED><<<blocks, 256>>>(out, iters, 0.999f, 1e-3f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)blocks * 256 * 16 * iters / (ms * 1e-3);
}

int main() {
  float *out; CK(hipMalloc(&out, 8192 * 256 * 4));
  for (int blocks : {256 * 2, 256 * 8}) {
    const double s = run<false>(out, blocks), p = run<true>(out, blocks);
    printf("%d workgroups of 256 (%d waves / SIMD): v_fma_f32 %.1f T element-FMA/s (%.1f TFLOP/s), v_pk_fma_f32 %.1f T (%.1f TFLOP/s), ratio %.2f\n",
           blocks, blocks / 256, s * 1e-12, 2 * s * 1e-12, p * 1e-12, 2 * p * 1e-12, p / s);
  }
  return 0;
}
