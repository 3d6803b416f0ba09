// mfma_shadow.hip — how many vector instructions of the SAME wave hide behind a v_mfma_f32_32x32x16_bf16 at one wave per SIMD?
// A wave loops over {1 MFMA, K fillers}; fillers are independent v_and/v_sub pairs (the operand-split chain of mlp_split.hip)
// or ds_write_b64 / ds_read_b64_tr_b16; the MFMAs rotate over R accumulators.  Prints cycles per MFMA (s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shadow.hip -o mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int K, int R, int KIND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) shadow(float *out, long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) uint2 lds[1024];
  v16f acc[R];
  for (int r = 0; r < R; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  const uint4 a4 = make_uint4(0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u);
  const bf16x8 va = __builtin_bit_cast(bf16x8, a4);
  float f[8], idle[8];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) idle[i] = (float)i;
  for (int i = 0; i < 8; ++i) { f[i] = 1.f + threadIdx.x * 1e-3f * (i + 1); u[i] = 0; }
  lds[threadIdx.x] = make_uint2(threadIdx.x, 1);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, va, acc[r], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) {   // alternating and / sub on 8 independent chains
          if (k & 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[k % 8]) : "v"(__uint_as_float(u[k % 8])));
          else asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[k % 8]) : "v"(f[k % 8]));
        } else if (KIND == 1) {
          asm volatile("ds_write_b64 %0, %1" :: "v"((uint32_t)(threadIdx.x * 8)), "v"(make_uint2(u[0], k)) : "memory");
        } else if (KIND == 3) {   // read an accumulator register no MFMA of the loop writes (acc_idle lives in AGPRs)
          float v;
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(idle[k % 8]));
          u[k % 8] ^= __float_as_uint(v);
        } else {
          uint2 v;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(threadIdx.x * 8)) : "memory");
          u[k % 8] ^= v.x;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < R; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
  for (int i = 0; i < 8; ++i) s += f[i] + u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int R, int KIND>
static void run(float *out, long long *cyc) {
  const int iters = 2000;
  shadow<K, R, KIND><<<256, 256>>>(out, cyc, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  shadow<K, R, KIND><<<256, 256>>>(out, cyc, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("  kind %d  R=%d  K=%2d fillers/MFMA: %.3f ms -> %.1f ns per MFMA (%.1f cycles at 2.0 GHz; readcyclecounter ticks %.1f)\n", KIND, R, K, ms,
         ms * 1e6 / (iters * R), ms * 1e6 / (iters * R) * 2.0, (double)c / (iters * R));
}

int main() {
  float *out; long long *cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 8));
  printf("VALU fillers (and/sub), accumulators rotated over R:\n");
  run<0, 1, 0>(out, cyc); run<4, 1, 0>(out, cyc); run<8, 1, 0>(out, cyc);
  run<0, 2, 0>(out, cyc); run<2, 2, 0>(out, cyc); run<4, 2, 0>(out, cyc); run<5, 2, 0>(out, cyc); run<6, 2, 0>(out, cyc); run<8, 2, 0>(out, cyc); run<12, 2, 0>(out, cyc);
  run<0, 4, 0>(out, cyc); run<4, 4, 0>(out, cyc); run<6, 4, 0>(out, cyc); run<8, 4, 0>(out, cyc); run<12, 4, 0>(out, cyc);
  printf("v_accvgpr_read_b32 fillers (of registers no MFMA in flight writes):\n");
  run<1, 4, 3>(out, cyc); run<2, 4, 3>(out, cyc); run<4, 4, 3>(out, cyc); run<8, 4, 3>(out, cyc);
  printf("ds_write_b64 fillers:\n");
  run<1, 2, 1>(out, cyc); run<2, 2, 1>(out, cyc); run<4, 2, 1>(out, cyc);
  printf("ds_read_b64_tr_b16 fillers:\n");
  run<1, 2, 2>(out, cyc); run<2, 2, 2>(out, cyc); run<4, 2, 2>(out, cyc);
  return 0;
}
