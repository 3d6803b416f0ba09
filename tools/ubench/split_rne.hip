// split_rne.hip — the exact 3-term bf16 split of an fp32 value, two ways:
//   trunc : t0 = top half of x, r1 = x - t0, t1 = top half of r1, r2 = r1 - t1, t2 = r2  (v_and / v_sub, + one v_perm per packed pair and term:
//           5.5 VALU instructions per value)
//   rne   : pairs at a time — p0 = v_cvt_pk_bf16_f32(a, b), r = v_dot2c_f32_bf16(p0, {-1,0} / {0,-1}, a / b), ... : the packed operand words
//           come out of the conversion directly, 3.5 instructions per value
// Checks that t0 + t1 + t2 == x EXACTLY for the rne form (the subtraction inside v_dot2c must not lose bits) and measures both rates.
// Build: hipcc --offload-arch=gfx950 -O3 split_rne.hip -o split_rne
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair_rne(float a, float b, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
  const bf16x2 m_lo = {(__bf16)-1.0f, (__bf16)0.0f}, m_hi = {(__bf16)0.0f, (__bf16)-1.0f};
  const bf16x2 q0 = __builtin_convertvector((f32x2){a, b}, bf16x2);
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(q0, m_lo, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(q0, m_hi, b, false);
  const bf16x2 q1 = __builtin_convertvector((f32x2){ra, rb}, bf16x2);
  const float sa = __builtin_amdgcn_fdot2_f32_bf16(q1, m_lo, ra, false), sb = __builtin_amdgcn_fdot2_f32_bf16(q1, m_hi, rb, false);
  const bf16x2 q2 = __builtin_convertvector((f32x2){sa, sb}, bf16x2);
  p0 = __builtin_bit_cast(uint32_t, q0); p1 = __builtin_bit_cast(uint32_t, q1); p2 = __builtin_bit_cast(uint32_t, q2);
}
__device__ __forceinline__ void split_pair_trunc(float a, float b, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
  uint32_t t[2][3];
  const float x[2] = {a, b};
  for (int i = 0; i < 2; ++i) {
    t[i][0] = __float_as_uint(x[i]) & 0xffff0000u;
    const float r1 = x[i] - __uint_as_float(t[i][0]);
    t[i][1] = __float_as_uint(r1) & 0xffff0000u;
    t[i][2] = __float_as_uint(r1 - __uint_as_float(t[i][1]));
  }
  p0 = __builtin_amdgcn_perm(t[1][0], t[0][0], 0x07060302u); p1 = __builtin_amdgcn_perm(t[1][1], t[0][1], 0x07060302u);
  p2 = __builtin_amdgcn_perm(t[1][2], t[0][2], 0x07060302u);
}

template <bool RNE>
__global__ void split_kernel(int64_t n_pairs, const float *x, uint32_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  uint32_t p0, p1, p2;
  if (RNE) split_pair_rne(x[2 * i], x[2 * i + 1], p0, p1, p2); else split_pair_trunc(x[2 * i], x[2 * i + 1], p0, p1, p2);
  out[3 * i] = p0; out[3 * i + 1] = p1; out[3 * i + 2] = p2;
}
// rate: a dependent-free stream of splits on register values
template <bool RNE>
__global__ void rate_kernel(int iters, const float *x, uint32_t *out) {
  float a = x[threadIdx.x], b = x[threadIdx.x + 64];
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      uint32_t p0, p1, p2;
      if (RNE) split_pair_rne(a + (float)u, b - (float)u, p0, p1, p2); else split_pair_trunc(a + (float)u, b - (float)u, p0, p1, p2);
      acc ^= p0 + p1 * 3u + p2 * 5u;
    }
    a = __uint_as_float((acc & 0x007fffffu) | 0x3f800000u);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static float bf(uint32_t h) { uint32_t w = h << 16; float f; memcpy(&f, &w, 4); return f; }

int main() {
  const int64_t n = 1 << 24;
  std::vector<float> h(n);
  srand(1);
  for (int64_t i = 0; i < n; ++i) {
    uint32_t w = ((uint32_t)rand() << 16) ^ (uint32_t)rand() ^ ((uint32_t)rand() << 31);
    uint32_t e = (i % 3 == 0) ? 100 + rand() % 56 : 1 + rand() % 253;   // a third around 1.0, the rest anywhere normal
    w = (w & 0x807fffffu) | (e << 23);
    if (i % 1000 == 7) w |= 0x007fffffu;    // all-ones mantissas: the conversion rounds up into the next binade
    if (i % 1000 == 9) w &= 0xff800000u;    // powers of two
    memcpy(&h[i], &w, 4);
  }
  float *dx; uint32_t *dout;
  CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dout, n / 2 * 3 * 4));
  CK(hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> o(n / 2 * 3);
  for (int mode = 0; mode < 2; ++mode) {
    if (mode) split_kernel<true><<<(unsigned)((n / 2 + 255) / 256), 256>>>(n / 2, dx, dout); else split_kernel<false><<<(unsigned)((n / 2 + 255) / 256), 256>>>(n / 2, dx, dout);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0, big1 = 0, big2 = 0;
    double worst = 0;
    for (int64_t i = 0; i < n / 2; ++i)
      for (int k = 0; k < 2; ++k) {
        const float x = h[2 * i + k];
        const float t0 = bf((o[3 * i] >> (16 * k)) & 0xffffu), t1 = bf((o[3 * i + 1] >> (16 * k)) & 0xffffu), t2 = bf((o[3 * i + 2] >> (16 * k)) & 0xffffu);
        const double s = (double)t0 + (double)t1 + (double)t2;
        if (s != (double)x) { ++bad; const double e = fabs(s - x) / fabs(x); if (e > worst) worst = e; }
        if (fabs(t1) > ldexp(fabs(x), -7)) ++big1;
        if (fabs(t2) > ldexp(fabs(x), -15)) ++big2;
      }
    printf("%s: %lld of %lld values with t0+t1+t2 != x (worst relative error %.3g); |t1| > 2^-7 |x|: %lld, |t2| > 2^-15 |x|: %lld\n", mode ? "rne  " : "trunc",
           (long long)bad, (long long)n, worst, (long long)big1, (long long)big2);
  }
  // issue rate
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000, blocks = 256 * 8;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (mode) rate_kernel<true><<<blocks, 256>>>(iters, dx, dout); else rate_kernel<false><<<blocks, 256>>>(iters, dx, dout);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%s: %.3f ms for %.3g value splits -> %.1f G values/s chip-wide\n", mode ? "rne  " : "trunc", ms, (double)blocks * 256 * iters * 32, (double)blocks * 256 * iters * 32 / ms * 1e-6);
    }
  }
  return 0;
}
