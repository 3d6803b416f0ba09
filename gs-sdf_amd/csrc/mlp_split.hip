// mlp_split.hip — S2 / S2' on the bf16 MFMA pipe at fp32 accuracy.
// Same operator and same tensors as mlp.hip (TCNNNetwork::forward / the torch Sequential decoder,
// /root/reference/include/neural_net/local_map.cpp:29-55, :94); what changes is how a dot product is evaluated.
//
// Why: v_mfma_f32_32x32x2_f32 retires 2 k per 64 cycles, v_mfma_f32_32x32x16_bf16 16 k per 32 cycles (measured on MI355X,
// tools/ubench/mfma_split.hip: 116 against 1848 TFLOP/s at the clock the matrix pipe sustains).  mlp.hip's forward already
// sits on the fp32 pipe's bound (0.93 ms for 3.3 M points), so the decoder can only get faster on the other pipe.
//
// Numerics: every fp32 operand x is split EXACTLY into three bf16 terms x = x0 + x1 + x2 (x0 = the top 16 bits of the word,
// x1 = the top 16 bits of x - x0, x2 = x - x0 - x1: 8 + 8 + 8 mantissa bits, both residuals exact in fp32), and a product
// a*b is the six partial products of order >= 2^-16, a0b2 + a2b0 + a1b1 + a0b1 + a1b0 + a0b0, accumulated in fp32 smallest
// first.  What is dropped (a1b2 + a2b1 + a2b2) is below 2^-23 |ab|: measured error of a K = 64 contraction against fp64
// 1.2e-7 rel-L2, the fp32 MFMA's own is 1.4e-7 (same ubench).  Six bf16 MFMAs cost 192 cycles per 16 k, the fp32 pipe
// 512: 2.67x.
//
// MI355X mapping: the register-chaining of mlp.hip carries over — a wave owns a 32-point tile, D's layout (lane = point,
// 16 registers = neurons d_row(r, half)) is the next layer's B operand after a permutation of the K order: k-step s of
// a 64-wide hidden operand is the 8 registers 8 (s & 1) .. + 7 of accumulator tile s >> 1 from both half-waves; the weights
// are split once per workgroup into the same order in LDS (one conflict-free ds_read_b128 per A operand).
#include "mlp_common.h"

namespace gsdf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int SPLIT_FWD_THREADS = 512;  // 8 waves share one 98 KiB weight image: 2 waves per SIMD

__device__ __forceinline__ v16f mfma_bf16(const uint4 &a, const uint4 &b, v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the three bf16 terms of x as the TOP halves of t0, t1, t2 (exact: t0 + t1 + t2 == x)
__device__ __forceinline__ void split3(float x, uint32_t &t0, uint32_t &t1, uint32_t &t2) {
  t0 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(t0);
  t1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(t1);
  t2 = __float_as_uint(r2);   // 8 significant bits left: its low half is zero
}
// bf16 pair {e0 (low half), e1 (high half)} from the top halves of two words: one v_perm_b32
__device__ __forceinline__ uint32_t pack_top(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

struct Split8 {   // one MFMA operand (8 consecutive k of a lane) in its three terms
  uint4 s[3];
};
__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
  uint32_t t[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3(x[e], t[0][e], t[1][e], t[2][e]);
  Split8 o;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    o.s[j] = make_uint4(pack_top(t[j][0], t[j][1]), pack_top(t[j][2], t[j][3]), pack_top(t[j][4], t[j][5]), pack_top(t[j][6], t[j][7]));
  return o;
}

// the six partial products of two accumulator tiles that share the B operand, interleaved so that consecutive MFMAs never
// depend on each other
__device__ __forceinline__ void mfma6x2(const Split8 &a0, const Split8 &a1, const Split8 &b, v16f &c0, v16f &c1) {
  c0 = mfma_bf16(a0.s[0], b.s[2], c0); c1 = mfma_bf16(a1.s[0], b.s[2], c1);
  c0 = mfma_bf16(a0.s[2], b.s[0], c0); c1 = mfma_bf16(a1.s[2], b.s[0], c1);
  c0 = mfma_bf16(a0.s[1], b.s[1], c0); c1 = mfma_bf16(a1.s[1], b.s[1], c1);
  c0 = mfma_bf16(a0.s[0], b.s[1], c0); c1 = mfma_bf16(a1.s[0], b.s[1], c1);
  c0 = mfma_bf16(a0.s[1], b.s[0], c0); c1 = mfma_bf16(a1.s[1], b.s[0], c1);
  c0 = mfma_bf16(a0.s[0], b.s[0], c0); c1 = mfma_bf16(a1.s[0], b.s[0], c1);
}
__device__ __forceinline__ void mfma6(const Split8 &a, const Split8 &b, v16f &c) {
  c = mfma_bf16(a.s[0], b.s[2], c);
  c = mfma_bf16(a.s[2], b.s[0], c);
  c = mfma_bf16(a.s[1], b.s[1], c);
  c = mfma_bf16(a.s[0], b.s[1], c);
  c = mfma_bf16(a.s[1], b.s[0], c);
  c = mfma_bf16(a.s[0], b.s[0], c);
}

// k order of a hidden (64-wide) operand: element e of half h in k-step s  ->  neuron
__device__ __forceinline__ int split_k_hidden(int s, int h, int e) { return 32 * (s >> 1) + d_row(8 * (s & 1) + e, h); }

struct SplitLds {
  int off4[MAX_LAYERS];   // uint4 offset of layer l's operand image
};

// ----------------------------------------------------------------------------------------------
// forward
// LDS image of layer l: [o_tile][k_step][term][64 lanes] x uint4 = the A operand of lane (row o = 32 o_tile + (lane & 31), half)
// ----------------------------------------------------------------------------------------------
__device__ void stage_split_fwd(const MlpDesc &d, const SplitLds &sl, const float *__restrict__ W, const float *__restrict__ bias,
                                uint4 *lds_w, float *lds_b) {
  for (int l = 0; l < d.n_layers; ++l) {
    const int I = l == 0 ? d.d_in : HID;
    const int O = l == d.n_layers - 1 ? d.d_out : HID;
    const int otiles = l == d.n_layers - 1 ? 1 : 2;
    const int ksteps = I / 16;
    const float *Wl = W + d.w_off[l];
    uint4 *dst = lds_w + sl.off4[l];
    for (int u = threadIdx.x; u < otiles * ksteps * 64; u += blockDim.x) {
      const int lane = u & 63, s = (u >> 6) % ksteps, t = (u >> 6) / ksteps;
      const int o = 32 * t + (lane & 31), h = lane >> 5;
      // the 8 k of this operand are two runs of 4 consecutive input neurons (one run of 8 for the natural order of layer 0)
      const int k0 = l == 0 ? 16 * s + 8 * h : split_k_hidden(s, h, 0);
      const int k1 = l == 0 ? k0 + 4 : split_k_hidden(s, h, 4);
      float w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (o < O) {
        const float4 a = *reinterpret_cast<const float4 *>(Wl + o * I + k0), b = *reinterpret_cast<const float4 *>(Wl + o * I + k1);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
      }
      const Split8 sp = split8(w);
#pragma unroll
      for (int j = 0; j < 3; ++j) dst[((t * ksteps + s) * 3 + j) * 64 + lane] = sp.s[j];
    }
    for (int e = threadIdx.x; e < HID; e += blockDim.x) lds_b[l * HID + e] = (d.has_bias && e < O) ? bias[d.b_off[l] + e] : 0.f;
  }
}

__device__ __forceinline__ Split8 lds_operand(const uint4 *img, int idx, int lane) {
  Split8 a;
#pragma unroll
  for (int j = 0; j < 3; ++j) a.s[j] = img[(idx * 3 + j) * 64 + lane];
  return a;
}

template <int D_IN>
__global__ void __launch_bounds__(SPLIT_FWD_THREADS)
    mlp_fwd_split_kernel(int64_t B, MlpDesc d, SplitLds sl, const float *__restrict__ W, const float *__restrict__ bias,
                         const float *__restrict__ in, float *__restrict__ out, float *__restrict__ acts) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  float *lds_b = reinterpret_cast<float *>(smem4);   // [MAX_LAYERS][64]
  uint4 *lds_w = smem4 + MAX_LAYERS * HID / 4;
  stage_split_fwd(d, sl, W, bias, lds_w, lds_b);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  constexpr int KS0 = D_IN / 16;
  constexpr int WAVES = SPLIT_FWD_THREADS / 64;
  const int64_t n_tiles = (B + 31) / 32;
  uint16_t *masks = acts == nullptr ? nullptr : reinterpret_cast<uint16_t *>(acts + img_off(d.n_layers - 1, n_tiles, 0, 0, 0));
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
    const int64_t p = tile * 32 + pl;
    const bool live = p < B;
    v16f cur[2];
    {  // ---- layer 0: k-step s of this lane = input features 16 s + 8 h .. + 7
      Split8 xb[KS0];
      const float4 *src = reinterpret_cast<const float4 *>(in + (live ? p : 0) * D_IN + 8 * h);
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        const float4 u = live ? src[4 * s] : make_float4(0.f, 0.f, 0.f, 0.f), v = live ? src[4 * s + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        xb[s] = split8(x);
      }
      const uint4 *w = lds_w + sl.off4[0];
      v16f acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = lds_b[d_row(r, h)]; acc1[r] = lds_b[32 + d_row(r, h)]; }
#pragma unroll
      for (int s = 0; s < KS0; ++s) mfma6x2(lds_operand(w, s, lane), lds_operand(w, KS0 + s, lane), xb[s], acc0, acc1);
#pragma unroll
      for (int r = 0; r < 16; ++r) { cur[0][r] = fmaxf(acc0[r], 0.f); cur[1][r] = fmaxf(acc1[r], 0.f); }
    }
    for (int l = 1; l < d.n_layers; ++l) {
      if (acts != nullptr) {  // post-ReLU activations of layer l-1 as a register image (dead lanes of the last tile too)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float4 *a = reinterpret_cast<float4 *>(acts + img_off(l - 1, n_tiles, tile, t, lane));
          unsigned m = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = make_float4(cur[t][4 * q], cur[t][4 * q + 1], cur[t][4 * q + 2], cur[t][4 * q + 3]);
#pragma unroll
          for (int r = 0; r < 16; ++r) m |= (cur[t][r] > 0.f ? 1u : 0u) << r;
          masks[mask_off(l - 1, n_tiles, tile, t, lane)] = (uint16_t)m;
        }
      }
      Split8 hb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = cur[s >> 1][8 * (s & 1) + e];
        hb[s] = split8(x);
      }
      const uint4 *w = lds_w + sl.off4[l];
      const bool last = l == d.n_layers - 1;
      v16f acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = lds_b[l * HID + d_row(r, h)]; acc1[r] = lds_b[l * HID + 32 + d_row(r, h)]; }
      if (!last) {
#pragma unroll
        for (int s = 0; s < 4; ++s) mfma6x2(lds_operand(w, s, lane), lds_operand(w, 4 + s, lane), hb[s], acc0, acc1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { cur[0][r] = fmaxf(acc0[r], 0.f); cur[1][r] = fmaxf(acc1[r], 0.f); }
      } else {   // one output tile: two k-steps in flight instead
        v16f accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[r] = 0.f;
        mfma6(lds_operand(w, 0, lane), hb[0], acc0); mfma6(lds_operand(w, 1, lane), hb[1], accb);
        mfma6(lds_operand(w, 2, lane), hb[2], acc0); mfma6(lds_operand(w, 3, lane), hb[3], accb);
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[0][r] = acc0[r] + accb[r];
      }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = d_row(r, h);
        if (o < d.d_out) out[p * d.d_out + o] = cur[0][r];
      }
    }
  }
}

// LDS bytes of the forward image; 0 if the topology is not covered
static size_t split_fwd_lds(const MlpDesc &d, SplitLds *sl) {
  int off = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    sl->off4[l] = off;
    const int I = l == 0 ? d.d_in : HID;
    off += (l == d.n_layers - 1 ? 1 : 2) * (I / 16) * 3 * 64;
  }
  return (size_t)off * 16 + MAX_LAYERS * HID * sizeof(float);
}

static bool split_enabled() {
  static const bool on = [] { const char *e = getenv("GSDF_MLP_MFMA"); return !(e && (e[0] == 'f' || e[0] == 'F')); }();   // GSDF_MLP_MFMA=f32: the fp32 pipe
  return on;
}

static unsigned split_grid(int64_t B, int waves) {
  static const int64_t cap = [] { const char *e = getenv("GSDF_MLP_SPLIT_WG"); return e ? (int64_t)atoi(e) : (int64_t)256; }();   // the image fills most of a CU's LDS: one workgroup per CU
  const int64_t wg = ((B + 31) / 32 + waves - 1) / waves;
  return (unsigned)(wg < 1 ? 1 : (wg > cap ? cap : wg));
}

// returns 1 if launched, 0 if this path does not cover the call (the caller falls back to the fp32 pipe), < 0 on error
int mlp_fwd_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *bias, const float *in, float *out, float *acts,
                         hipStream_t stream) {
  if (!split_enabled()) return 0;
  SplitLds sl;
  const size_t lds = split_fwd_lds(d, &sl);
  if (lds > 160 * 1024) return 0;
  const unsigned grid = split_grid(B, SPLIT_FWD_THREADS / 64);
  if (d.d_in == 32) {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_split attr");
    mlp_fwd_split_kernel<32><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, bias, in, out, acts);
  } else {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_split attr");
    mlp_fwd_split_kernel<64><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, bias, in, out, acts);
  }
  GSDF_CHECK_LAUNCH("mlp_fwd_split_kernel");
  return 1;
}

}  // namespace gsdf
