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

static bool split_enabled();
static unsigned split_grid(int64_t B, int waves);

struct SplitLds {
  int off4[MAX_LAYERS];   // uint4 offset of layer l's operand image
};

// ----------------------------------------------------------------------------------------------
// forward
// LDS image of layer l: [o_tile][k_step][term][64 lanes] x uint4 = the A operand of lane (row o = 32 o_tile + (lane & 31), half)
// ----------------------------------------------------------------------------------------------
// valu_last (round 6): an output layer of <= 4 neurons (the SDF head: sdf, isigma) is 32 fp32 FMAs per lane and output instead of a 32-row MFMA tile
// of which 2 rows are used (24 of the 192 MFMAs of a tile, and the operand split of a whole hidden layer in front of them): its weights are staged
// as plain floats [o][half][tile t][register r] = W_last[o][32 t + d_row(r, half)] — the neuron order of a lane's accumulator registers.
__device__ void stage_split_fwd(const MlpDesc &d, const SplitLds &sl, const float *__restrict__ W, const float *__restrict__ bias,
                                uint4 *lds_w, float *lds_b, bool valu_last = false) {
  for (int l = 0; l < d.n_layers; ++l) {
    const int I = l == 0 ? d.d_in : HID;
    const int O = l == d.n_layers - 1 ? d.d_out : HID;
    if (valu_last && l == d.n_layers - 1) {
      float *dst = reinterpret_cast<float *>(lds_w + sl.off4[l]);
      for (int u = threadIdx.x; u < 4 * 64; u += blockDim.x) {
        const int o = u >> 6, h = (u >> 5) & 1, t = (u >> 4) & 1, r = u & 15;
        dst[u] = o < O ? W[d.w_off[l] + o * I + 32 * t + d_row(r, h)] : 0.f;
      }
      for (int e = threadIdx.x; e < HID; e += blockDim.x) lds_b[l * HID + e] = (d.has_bias && e < O) ? bias[d.b_off[l] + e] : 0.f;
      continue;
    }
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

// accumulator tiles initialised with the layer's biases: registers 4k .. 4k+3 of a lane are 4 consecutive neurons (one
// ds_read_b128 each); a bias-free network starts from zero without touching LDS
__device__ __forceinline__ void load_bias(const float *lds_b, int l, int h, int has_bias, v16f &acc0, v16f &acc1) {
  if (!has_bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 b0 = *reinterpret_cast<const float4 *>(lds_b + l * HID + 8 * k + 4 * h);
    const float4 b1 = *reinterpret_cast<const float4 *>(lds_b + l * HID + 32 + 8 * k + 4 * h);
    acc0[4 * k] = b0.x; acc0[4 * k + 1] = b0.y; acc0[4 * k + 2] = b0.z; acc0[4 * k + 3] = b0.w;
    acc1[4 * k] = b1.x; acc1[4 * k + 1] = b1.y; acc1[4 * k + 2] = b1.z; acc1[4 * k + 3] = b1.w;
  }
}

// MASKED: the ReLUs are replaced by the saved masks of an earlier forward pass (`mask_acts` = that pass's acts buffer), no biases:
// y = W_{n-1} D_{n-2} ... D_0 W_0 x, the tangent pass of the decoder's DOUBLE backward (gsdf_mlp_bwd_bwd); `acts` receives the tangent
// images t_0 .. t_{n-2} (no masks of its own).
template <int D_IN, int THREADS, bool MASKED = false, bool VALU_LAST = false>
__global__ void __launch_bounds__(THREADS)
    mlp_fwd_split_kernel(int64_t B, MlpDesc d, SplitLds sl, const float *__restrict__ W, const float *__restrict__ bias,
                         const float *__restrict__ in, float *__restrict__ out, float *__restrict__ acts,
                         const float *__restrict__ mask_acts = nullptr) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  float *lds_b = reinterpret_cast<float *>(smem4);   // [MAX_LAYERS][64]
  uint4 *lds_w = smem4 + MAX_LAYERS * HID / 4;
  stage_split_fwd(d, sl, W, bias, lds_w, lds_b, VALU_LAST);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  constexpr int KS0 = D_IN / 16;
  constexpr int WAVES = THREADS / 64;
  const int64_t n_tiles = (B + 31) / 32;
  uint16_t *masks = acts == nullptr ? nullptr : reinterpret_cast<uint16_t *>(acts + img_off(d.n_layers - 1, n_tiles, 0, 0, 0));
  const uint16_t *msrc = MASKED ? reinterpret_cast<const uint16_t *>(mask_acts + img_off(d.n_layers - 1, n_tiles, 0, 0, 0)) : nullptr;
  // the input rows of a tile are loaded one tile ahead (branch-free: clamped address, dead lanes zeroed by a select)
  float4 xin[2 * KS0];
  auto load_rows = [&](int64_t tile) {
    const int64_t p = tile * 32 + pl;
    const bool ok = tile < n_tiles && p < B;
    const float4 *src = reinterpret_cast<const float4 *>(in + (ok ? p : 0) * D_IN + 8 * h);
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const float4 u = src[4 * s], v = src[4 * s + 1];
      xin[2 * s] = ok ? u : make_float4(0.f, 0.f, 0.f, 0.f);
      xin[2 * s + 1] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const int64_t tile0 = (int64_t)blockIdx.x * WAVES + wave, tstride = (int64_t)gridDim.x * WAVES;
  load_rows(tile0);
  for (int64_t tile = tile0; tile < n_tiles; tile += tstride) {
    const int64_t p = tile * 32 + pl;
    const bool live = p < B;
    v16f cur[2];
    {  // ---- layer 0: k-step s of this lane = input features 16 s + 8 h .. + 7
      Split8 xb[KS0];
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        const float x[8] = {xin[2 * s].x, xin[2 * s].y, xin[2 * s].z, xin[2 * s].w, xin[2 * s + 1].x, xin[2 * s + 1].y, xin[2 * s + 1].z, xin[2 * s + 1].w};
        xb[s] = split8(x);
      }
      load_rows(tile + tstride);
      const uint4 *w = lds_w + sl.off4[0];
      v16f acc0, acc1;
      load_bias(lds_b, 0, h, MASKED ? 0 : d.has_bias, acc0, acc1);
      unsigned m0 = 0, m1 = 0;
      if (MASKED) { m0 = msrc[mask_off(0, n_tiles, tile, 0, lane)]; m1 = msrc[mask_off(0, n_tiles, tile, 1, lane)]; }
#pragma unroll
      for (int s = 0; s < KS0; ++s) mfma6x2(lds_operand(w, s, lane), lds_operand(w, KS0 + s, lane), xb[s], acc0, acc1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cur[0][r] = MASKED ? ((m0 >> r) & 1u ? acc0[r] : 0.f) : fmaxf(acc0[r], 0.f);
        cur[1][r] = MASKED ? ((m1 >> r) & 1u ? acc1[r] : 0.f) : fmaxf(acc1[r], 0.f);
      }
    }
    for (int l = 1; l < d.n_layers; ++l) {
      if (acts != nullptr) {  // post-ReLU activations of layer l-1 as a register image (dead lanes of the last tile too)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float4 *a = reinterpret_cast<float4 *>(acts + img_off(l - 1, n_tiles, tile, t, lane));
          unsigned m = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q * (IMG_Q / 4)] = make_float4(cur[t][4 * q], cur[t][4 * q + 1], cur[t][4 * q + 2], cur[t][4 * q + 3]);
          if (!MASKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m |= (cur[t][r] > 0.f ? 1u : 0u) << r;
            masks[mask_off(l - 1, n_tiles, tile, t, lane)] = (uint16_t)m;
          }
        }
      }
      // B operand of k-step s = 8 registers of tile s >> 1, split right before its MFMAs (12 live registers instead of 48)
      auto hb = [&](int s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = cur[s >> 1][8 * (s & 1) + e];
        return split8(x);
      };
      const uint4 *w = lds_w + sl.off4[l];
      const bool last = l == d.n_layers - 1;
      if (VALU_LAST && last) {
        // the output layer on the vector pipe: this lane's 32 hidden values against their weights (one broadcast ds_read_b128 per four), the two
        // halves of a point added through the crossbar; fp32 FMAs in neuron order of the registers
        const float4 *wl = reinterpret_cast<const float4 *>(w);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (o < d.d_out) {
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 wv = wl[((o * 2 + h) * 2 + t) * 4 + q];
                sacc = fmaf(wv.x, cur[t][4 * q], sacc); sacc = fmaf(wv.y, cur[t][4 * q + 1], sacc);
                sacc = fmaf(wv.z, cur[t][4 * q + 2], sacc); sacc = fmaf(wv.w, cur[t][4 * q + 3], sacc);
              }
            sacc += __shfl_xor(sacc, 32, 64);
            if (!MASKED && d.has_bias) sacc += lds_b[l * HID + o];
            if (live && h == (o & 1)) out[p * d.d_out + o] = sacc;
          }
        }
        continue;
      }
      v16f acc0, acc1;
      load_bias(lds_b, l, h, MASKED ? 0 : d.has_bias, acc0, acc1);
      unsigned m0 = 0, m1 = 0;
      if (MASKED && !last) { m0 = msrc[mask_off(l, n_tiles, tile, 0, lane)]; m1 = msrc[mask_off(l, n_tiles, tile, 1, lane)]; }
      if (!last) {
        // A operands one k-step ahead: their LDS latency hides behind the 12 MFMAs of the current k-step
        Split8 a0 = lds_operand(w, 0, lane), a1 = lds_operand(w, 4, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          Split8 n0 = a0, n1 = a1;
          if (s < 3) { n0 = lds_operand(w, s + 1, lane); n1 = lds_operand(w, 4 + s + 1, lane); }
          mfma6x2(a0, a1, hb(s), acc0, acc1);
          __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from hoisting all four operand splits (48 registers) to the top
          a0 = n0; a1 = n1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          cur[0][r] = MASKED ? ((m0 >> r) & 1u ? acc0[r] : 0.f) : fmaxf(acc0[r], 0.f);
          cur[1][r] = MASKED ? ((m1 >> r) & 1u ? acc1[r] : 0.f) : fmaxf(acc1[r], 0.f);
        }
      } else {   // one output tile: two k-steps in flight instead
        v16f accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[r] = 0.f;
        mfma6(lds_operand(w, 0, lane), hb(0), acc0); mfma6(lds_operand(w, 1, lane), hb(1), accb);
        __builtin_amdgcn_sched_barrier(0);
        mfma6(lds_operand(w, 2, lane), hb(2), acc0); mfma6(lds_operand(w, 3, lane), hb(3), accb);
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[0][r] = acc0[r] + accb[r];
      }
    }
    if (live && !VALU_LAST) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = d_row(r, h);
        if (o < d.d_out) out[p * d.d_out + o] = cur[0][r];
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// backward: data path and weight path in ONE pass over the points.
//
// mlp.hip's backward is two kernels with the per-layer gradients v_pre (256 B per point and layer) written to HBM by the first
// and read back, next to the saved activations, by the second: 7 GB per 3.3 M points, which is what bounds it (5 TB/s).  Here
// a wave keeps v_pre in registers: per layer it (1) splits g = v_pre_l once, (2) accumulates dW_l += v_pre_l^T a_{l-1} over
// its 32 points and (3) chains g <- relu'(a_{l-1}) (W_l^T g).  The weight-gradient tiles of EVERY layer (14.7 k values = 16
// 32x32 tiles, + 2 for the biases) live in the accumulator registers of each wave for the whole kernel — 288 of the 512
// registers a lane owns at one wave per SIMD — and leave through one round of atomics per wave at the very end.
//
// (2) contracts over points, which D's layout has on LANES while an MFMA operand wants them on ELEMENTS: both operands go
// through a wave-private LDS block (tr_write8 / tr_read below: packed pairs written with ds_write_b64, read back with the
// gfx950 transpose read ds_read_b64_tr_b16).  The bias gradient is the same A operand against a one-hot B column (column l of
// a shared tile collects layer l's sums).
// ----------------------------------------------------------------------------------------------
static constexpr int SPLIT_BWD_THREADS = 256;
// Modes of the one-pass backward kernel (round 4: the analytic configuration's e0 backward and double backward on this pipe too)
//   BWD_FULL    : v_in and the weight / bias gradients from v_out, the saved activations and the network input
//   BWD_DATA    : v_in only (the e0 backward dsdf/dfeat before the loss): the chain alone, ReLU derivatives from the saved MASKS (2 B per
//                 lane and tile instead of the 64 B of activations), nothing saved, no transposition, two waves per SIMD
//   BWD_TANGENT : the double backward's weight term dL/dW_l += v_pre_l (x) t_{l-1}: the chain from v_out is RECOMPUTED in registers
//                 (masks) and paired with the tangent images t (written by the masked forward) in place of the activations, the network
//                 input replaced by vv_in; no v_in, no bias term
enum { BWD_FULL = 0, BWD_DATA = 1, BWD_TANGENT = 2 };
static constexpr int TR_BLOCK = 256;           // 8-byte chunks per term of a transposition block: 32 points x 8 neuron quads
static constexpr int TR_WAVE_BYTES = 3 * TR_BLOCK * 8;

// LDS image of layer l: [i_tile][k_step over o][term][64 lanes] x uint4 = the A operand of lane (row i = 32 i_tile + (lane & 31), half)
__device__ void stage_split_bwd(const MlpDesc &d, const SplitLds &sl, const float *__restrict__ W, uint4 *lds_w, bool want_in) {
  for (int l = want_in ? 0 : 1; l < d.n_layers; ++l) {
    const int I = l == 0 ? d.d_in : HID;
    const int O = l == d.n_layers - 1 ? d.d_out : HID;
    const int itiles = I / 32;
    const int ksteps = l == d.n_layers - 1 ? 2 : 4;   // o padded to 32 on the last layer
    const float *Wl = W + d.w_off[l];
    uint4 *dst = lds_w + sl.off4[l];
    for (int u = threadIdx.x; u < itiles * ksteps * 64; u += blockDim.x) {
      const int lane = u & 63, s = (u >> 6) % ksteps, t = (u >> 6) / ksteps;
      const int i = 32 * t + (lane & 31), h = lane >> 5;
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int o = split_k_hidden(s, h, e);
        w[e] = o < O ? Wl[o * I + i] : 0.f;
      }
      const Split8 sp = split8(w);
#pragma unroll
      for (int j = 0; j < 3; ++j) dst[((t * ksteps + s) * 3 + j) * 64 + lane] = sp.s[j];
    }
  }
}

// the three terms of registers 8 g .. 8 g + 7 of a tile
__device__ __forceinline__ void split8regs(const v16f &x, int g, uint32_t (&t)[3][8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) split3(x[8 * g + e], t[0][e], t[1][e], t[2][e]);
}
__device__ __forceinline__ Split8 pack8(const uint32_t (&t)[3][8]) {
  Split8 o;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    o.s[j] = make_uint4(pack_top(t[j][0], t[j][1]), pack_top(t[j][2], t[j][3]), pack_top(t[j][4], t[j][5]), pack_top(t[j][6], t[j][7]));
  return o;
}
// Transposition block (one per wave): per term 256 chunks of 8 bytes, chunk (point p, quad kq) = the bf16 terms of neurons
// 4 kq .. 4 kq + 3 of point p.  A lane (point, half) owns whole chunks (its registers 4k .. 4k+3 are 4 consecutive neurons),
// so it WRITES packed pairs with ds_write_b64; the consumer lane (neuron, half) READS with the gfx950 transpose read
// ds_read_b64_tr_b16: the 16 lanes of a group each name one chunk (4 points x 4 quads) and lane c receives, for each of the
// 4 points, the c-th of the group's 16 neurons -> 4 consecutive points of ITS neuron; two reads make one MFMA operand term.
// Chunk placement 8 p + ((kq + (p >> 1)) & 7) keeps both sides conflict-free: the 16 consecutive points of a write group
// fall on 16 different bank pairs, the 4 points x 8 quads of a 32-lane read group on all 32.
__device__ __forceinline__ int tr_chunk(int p, int kq) { return 8 * p + ((kq + (p >> 1)) & 7); }

// lane (point, half) writes the packed terms of its registers 8 g .. 8 g + 7 (two chunks per term).  NATURAL: the registers
// are network-input features 16 half + r (quad 4 half + r / 4), else neurons d_row(r, half) (quad 2 (r / 4) + half)
template <bool NATURAL>
__device__ __forceinline__ void tr_write8(uint2 *tb, const Split8 &pk, int g, int lane) {
  const int pt = lane & 31, h = lane >> 5;
  const int kq0 = NATURAL ? 4 * h + 2 * g : 4 * g + h, kq1 = NATURAL ? kq0 + 1 : kq0 + 2;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    tb[j * TR_BLOCK + tr_chunk(pt, kq0)] = make_uint2(pk.s[j].x, pk.s[j].y);
    tb[j * TR_BLOCK + tr_chunk(pt, kq1)] = make_uint2(pk.s[j].z, pk.s[j].w);
  }
}
// operand of lane (neuron = lane & 31, half) for k-step ks = points 16 ks + 8 half .. + 7
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Split8 tr_read(const uint2 *tb, int ks, int lane) {
  const int p0 = 16 * ks + 8 * (lane >> 5) + ((lane & 15) >> 2), kq = 4 * ((lane >> 4) & 1) + (lane & 3);
  Split8 a;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    typedef __attribute__((address_space(3))) short4v *lds_p;
    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tb + j * TR_BLOCK + tr_chunk(p0, kq))));
    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tb + j * TR_BLOCK + tr_chunk(p0 + 4, kq))));
    a.s[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
  return a;
}

template <int NL>
struct BwdAcc {
  v16f w0[2];                           // layer 0 (input width 32): [o tile]
  v16f wh[NL > 2 ? NL - 2 : 1][2][2];   // layers 1 .. NL-2: [o tile][i tile]
  v16f wl[2];                           // last layer (outputs padded to 32): [i tile]
  v16f b;                               // bias sums: column l = output tile 0 of layer l, column 8 + l = tile 1
};

struct BwdCtx {
  int64_t B, n_tiles;
  const uint16_t *masks;   // modes DATA / TANGENT: the ReLU masks of the first forward
  const float *in, *acts, *v_out;
  float *v_in;
  float *g_img;            // two-range backward: v_pre below the upper range's bottom layer, register image [tile][t] (written by the upper, read by the lower pass)
  const uint4 *lds_w;
  uint2 *tb;
  int lane, d_out;
};

// Everything a tile's backward loads from HBM is read through branch-free code (clamped addresses + selects) so that a
// whole tile is ONE basic block the scheduler can interleave.
// input of layer L for a tile: the network input rows (lane (p, half) holds features 16 half + r) or the activation image L-1
template <int L>
__device__ __forceinline__ void load_layer_input(const BwdCtx &c, int64_t tile, v16f (&x)[2]) {
  const int64_t tc = tile < c.n_tiles ? tile : c.n_tiles - 1;   // past the end: any valid tile, the values are never used
  if (L == 0) {
    const int64_t p = tc * 32 + (c.lane & 31);
    const bool live = p < c.B;
    const float4 *src = reinterpret_cast<const float4 *>(c.in + (live ? p : c.B - 1) * 32 + 16 * (c.lane >> 5));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = src[q];
      x[0][4 * q] = live ? v.x : 0.f; x[0][4 * q + 1] = live ? v.y : 0.f; x[0][4 * q + 2] = live ? v.z : 0.f; x[0][4 * q + 3] = live ? v.w : 0.f;
    }
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 *a = reinterpret_cast<const float4 *>(c.acts + img_off(L - 1, c.n_tiles, tc, t, c.lane));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = a[q * (IMG_Q / 4)];
        x[t][4 * q] = v.x; x[t][4 * q + 1] = v.y; x[t][4 * q + 2] = v.z; x[t][4 * q + 3] = v.w;
      }
    }
  }
}

// ReLU masks of the output of layer L-1 (slot L-1) for a tile, both halves
template <int L>
__device__ __forceinline__ void load_masks(const BwdCtx &c, int64_t tile, unsigned (&hm)[2]) {
  const int64_t tc = tile < c.n_tiles ? tile : c.n_tiles - 1;
  hm[0] = c.masks[mask_off(L - 1, c.n_tiles, tc, 0, c.lane)];
  hm[1] = c.masks[mask_off(L - 1, c.n_tiles, tc, 1, c.lane)];
}

__device__ __forceinline__ void load_v_out(const BwdCtx &c, int64_t tile, v16f &g) {
  const int64_t p = tile * 32 + (c.lane & 31);
  const bool live = tile < c.n_tiles && p < c.B;
  const float *row = c.v_out + (live ? p : 0) * c.d_out;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = d_row(r, c.lane >> 5);
    const bool ok = live && o < c.d_out;
    const float v = row[ok ? o : 0];
    g[r] = ok ? v : 0.f;
  }
}

// hand-over of the two-range backward: the chain's state (v_pre of the layer below the upper range, ReLU derivative applied) as a register image
__device__ __forceinline__ void store_g(const BwdCtx &c, int64_t tile, const v16f (&g)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float4 *dst = reinterpret_cast<float4 *>(c.g_img + img_off(0, c.n_tiles, tile, t, c.lane));
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * (IMG_Q / 4)] = make_float4(g[t][4 * q], g[t][4 * q + 1], g[t][4 * q + 2], g[t][4 * q + 3]);
  }
}
__device__ __forceinline__ void load_g(const BwdCtx &c, int64_t tile, v16f (&g)[2]) {
  const int64_t tc = tile < c.n_tiles ? tile : c.n_tiles - 1;   // past the end: any valid tile, the values are never used
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 *src = reinterpret_cast<const float4 *>(c.g_img + img_off(0, c.n_tiles, tc, t, c.lane));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = src[q * (IMG_Q / 4)];
      g[t][4 * q] = v.x; g[t][4 * q + 1] = v.y; g[t][4 * q + 2] = v.z; g[t][4 * q + 3] = v.w;
    }
  }
}

// terms of the MT tiles of g: packed chain operands gb (lane = point) and, through the LDS block, the dW operands at (lane =
// output neuron)
template <int MT, bool TR = true>
__device__ __forceinline__ void prep_g(const BwdCtx &c, const v16f (&g)[2], Split8 (&gb)[4], Split8 (&at)[2][2]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int hg = 0; hg < 2; ++hg) {
      uint32_t t[3][8];
      split8regs(g[mt], hg, t);
      gb[2 * mt + hg] = pack8(t);
      if (TR) tr_write8<false>(c.tb, gb[2 * mt + hg], hg, c.lane);
    }
    if (TR) {
      at[mt][0] = tr_read(c.tb, 0, c.lane);
      at[mt][1] = tr_read(c.tb, 1, c.lane);
    }
  }
}

// One layer of the backward of one tile, in two phases ordered so that the loads of the next step are in flight early:
//   phase 1:  chain MFMAs  g' = W_L^T g   (operands gb),  terms of the layer input x = a_{L-1} -> LDS block -> bt
//   phase 2:  dW_L += v_pre_L^T a_{L-1} (+ bias column),   g' <- relu'(a_{L-1}) g', its terms -> gb', at' (next layer)
// In:  gb/at = terms of v_pre_L, x = a_{L-1} (loaded one layer ahead).  Out: gb/at of layer L-1, x = a_{L-2} (or, from layer 0,
// the terms of the next tile's v_out and its first input).
// Measured and NOT adopted (tools/ubench/mfma_shadow.hip): zipping the vector work of one phase into the MFMA gaps
// of the other by hand.  At one wave per SIMD only plain VALU instructions hide behind an MFMA (~5 per MFMA, and only when
// the next MFMA on the same accumulator is >= 4 MFMAs away); LDS instructions (8-16 cycles of issue each) and
// v_accvgpr_read (4-6) do not, and this kernel's vector work is a third LDS / accumulator traffic: the zipped stream ran
// 1.16 ms against 1.17 ms (4 layers) and 1.96 against 1.64 ms (5 layers + biases, more live registers -> spills).
// Layer ranges (round 5): a launch runs the layers LT .. LB of every tile.  LT = NL - 1, LB = 0 is the whole backward in one launch; the
// 5-layer instances run as TWO launches (upper range down to LB > 0, which leaves the chain's state in c.g_img; lower range from there),
// so that a wave holds the weight-gradient tiles of its range only (7 + 11 instead of 17) and nothing spills.
template <int NL, bool BIAS, int L, int MODE, int LT, int LB>
__device__ __forceinline__ void bwd_layer(const BwdCtx &c, const SplitLds &sl, BwdAcc<NL> &acc, int64_t tile, int64_t next_tile,
                                          Split8 (&gb)[4], Split8 (&at)[2][2], v16f (&x)[2], unsigned (&hm)[2]) {
  constexpr bool last = L == NL - 1, first = L == 0;
  constexpr bool bottom = L == LB;            // the range ends here: what follows is the next tile's top layer
  constexpr bool top_is_out = LT == NL - 1;   // the range starts from v_out (else from the hand-over image)
  constexpr bool DW = MODE != BWD_DATA;       // weight-gradient tiles are accumulated
  constexpr bool MASKS = MODE != BWD_FULL;    // relu'(a_{L-1}) from the saved masks: x is not the activation (or not loaded at all)
  constexpr int MT = last ? 1 : 2;   // 32-row tiles of outputs o
  constexpr int NT = first ? 1 : 2;  // 32-row tiles of inputs i
  constexpr int KS = last ? 1 : 4;   // k-steps of the chain (over o; the last layer's outputs fit the first k-step)
  constexpr int LH = first || last ? 0 : L - 1;
  const int lane = c.lane;
  // prefetch: the input of layer L-1, or the first input and v_out of the next tile
  // (with 5 layers' accumulators the prefetch is issued between the phases instead: 32 fewer live registers in phase 1)
  v16f xn[2], gn[2];
  unsigned hmn[2] = {0u, 0u};
  constexpr bool early = NL <= 4 || !DW || LT - LB + 1 < NL;
  auto prefetch = [&] {
    if (bottom) {
      if (DW) load_layer_input<LT>(c, next_tile, xn);
      if (top_is_out) load_v_out(c, next_tile, gn[0]);
      else load_g(c, next_tile, gn);
    } else if (DW) {
      load_layer_input<first ? 0 : L - 1>(c, tile, xn);
    }
  };
  if (early) prefetch();
  if (MASKS) {   // masks of the layer below (slot L-2), or of the next tile's top layer
    if (bottom) load_masks<(LT >= 1 ? LT : 1)>(c, next_tile, hmn);
    else if (L >= 2) load_masks<(L >= 2 ? L - 1 : 1)>(c, tile, hmn);
  }

  // ---- phase 1
  const uint4 *w = c.lds_w + sl.off4[L];
  v16f ng[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { ng[0][r] = 0.f; ng[1][r] = 0.f; }
  constexpr int KSC = (first && MODE == BWD_TANGENT) ? 0 : KS;   // the tangent pass has no use for v_in = W_0^T v_pre_0
#pragma unroll
  for (int s = 0; s < KSC; ++s) {
    if (NT == 2) mfma6x2(lds_operand(w, s, lane), lds_operand(w, (last ? 2 : 4) + s, lane), gb[s], ng[0], ng[1]);
    else mfma6(lds_operand(w, s, lane), gb[s], ng[0]);
  }
  Split8 bt[NT][2];
  if (DW) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int hg = 0; hg < 2; ++hg) {
        uint32_t t[3][8];
        split8regs(x[nt], hg, t);
        tr_write8<first>(c.tb, pack8(t), hg, lane);
      }
      bt[nt][0] = tr_read(c.tb, 0, lane);
      bt[nt][1] = tr_read(c.tb, 1, lane);
    }
  }
  v16f g[2];   // g' = relu'(a_{L-1}) (W_L^T g): x and the chain's accumulators end here
  if (!first) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = (MASKS ? ((hm[t] >> r) & 1u) != 0u : x[t][r] > 0.f) ? ng[t][r] : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!early) prefetch();

  // ---- phase 2
  if (BIAS && DW) {   // ONE shared tile: column L collects the sums of output tile 0 of layer L, column 8 + L those of tile 1
    const uint32_t one0 = (lane & 31) == L ? 0x3f803f80u : 0u, one1 = (lane & 31) == 8 + L ? 0x3f803f80u : 0u;
    const uint4 oh[2] = {make_uint4(one0, one0, one0, one0), make_uint4(one1, one1, one1, one1)};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 2; j >= 0; --j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc.b = mfma_bf16(at[mt][ks].s[j], oh[mt], acc.b);
  }
#pragma unroll
  for (int nt = 0; nt < (DW ? NT : 0); ++nt) {
    if (last) {
      mfma6(at[0][0], bt[nt][0], acc.wl[nt]);
      mfma6(at[0][1], bt[nt][1], acc.wl[nt]);
    } else if (first) {
      mfma6x2(at[0][0], at[MT - 1][0], bt[nt][0], acc.w0[0], acc.w0[1]);
      mfma6x2(at[0][1], at[MT - 1][1], bt[nt][1], acc.w0[0], acc.w0[1]);
    } else {
      mfma6x2(at[0][0], at[MT - 1][0], bt[nt][0], acc.wh[LH][0][nt], acc.wh[LH][1][nt]);
      mfma6x2(at[0][1], at[MT - 1][1], bt[nt][1], acc.wh[LH][0][nt], acc.wh[LH][1][nt]);
    }
  }
  Split8 gbn[4], atn[2][2];
  if (!bottom) {
    prep_g<2, DW>(c, g, gbn, atn);
  } else {
    if (first) {
      const int64_t p = tile * 32 + (lane & 31);
      if (c.v_in != nullptr && p < c.B) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4 *>(c.v_in + p * 32 + 8 * q + 4 * (lane >> 5)) = make_float4(ng[0][4 * q], ng[0][4 * q + 1], ng[0][4 * q + 2], ng[0][4 * q + 3]);
      }
    } else {
      store_g(c, tile, g);   // the lower range's launch continues from here
    }
    if (top_is_out) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gn[1][r] = 0.f;
      prep_g<1, DW>(c, gn, gbn, atn);   // the next tile's v_out
    } else {
      prep_g<2, DW>(c, gn, gbn, atn);   // the next tile's hand-over state
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 4; ++s) gb[s] = gbn[s];
  if (DW) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { at[mt][0] = atn[mt][0]; at[mt][1] = atn[mt][1]; }
    x[0] = xn[0]; x[1] = xn[1];
  }
  hm[0] = hmn[0]; hm[1] = hmn[1];
}

template <int NL, bool BIAS, int L, int MODE, int LT, int LB, bool DONE = (L < LB)>
struct BwdLayers {
  static __device__ __forceinline__ void run(const BwdCtx &c, const SplitLds &sl, BwdAcc<NL> &acc, int64_t tile, int64_t next_tile,
                                             Split8 (&gb)[4], Split8 (&at)[2][2], v16f (&x)[2], unsigned (&hm)[2]) {
    bwd_layer<NL, BIAS, L, MODE, LT, LB>(c, sl, acc, tile, next_tile, gb, at, x, hm);
    BwdLayers<NL, BIAS, L - 1, MODE, LT, LB>::run(c, sl, acc, tile, next_tile, gb, at, x, hm);
  }
};
template <int NL, bool BIAS, int L, int MODE, int LT, int LB>
struct BwdLayers<NL, BIAS, L, MODE, LT, LB, true> {
  static __device__ __forceinline__ void run(const BwdCtx &, const SplitLds &, BwdAcc<NL> &, int64_t, int64_t, Split8 (&)[4], Split8 (&)[2][2], v16f (&)[2],
                                             unsigned (&)[2]) {}
};

// a 32x32 tile leaves: rows o = o0 + d_row(r, half) < O, column i = i0 + (lane & 31).  PART: plain stores into the wave's own partial
// buffer (every element of the blob is written by exactly one tile); else one round of atomics on the gradient itself
template <bool PART>
__device__ __forceinline__ void flush_tile(const v16f &a, float *vw, int I, int O, int o0, int i0, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = o0 + d_row(r, lane >> 5);
    if (PART) { if (o < O) vw[o * I + i0 + (lane & 31)] = a[r]; }
    else if (o < O && a[r] != 0.f) atomicAdd(vw + o * I + i0 + (lane & 31), a[r]);
  }
}

// partial buffers [n_part][n_elem] -> chunk sums [RED_CHUNKS][n_elem] -> v_W / v_b (one atomic per element and launch: other launches
// may be accumulating into the same gradient from another stream)
static constexpr int RED_CHUNKS = 16;
__global__ void __launch_bounds__(256) mlp_partials_sum_kernel(int n_part, int n_elem, const float *__restrict__ part, float *__restrict__ chunk) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  const int per = (n_part + RED_CHUNKS - 1) / RED_CHUNKS;
  const int p0 = blockIdx.y * per, p1 = min(n_part, p0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = p0;
  for (; p + 3 < p1; p += 4) {
    s0 += part[(int64_t)p * n_elem + e]; s1 += part[(int64_t)(p + 1) * n_elem + e];
    s2 += part[(int64_t)(p + 2) * n_elem + e]; s3 += part[(int64_t)(p + 3) * n_elem + e];
  }
  for (; p < p1; ++p) s0 += part[(int64_t)p * n_elem + e];
  chunk[(int64_t)blockIdx.y * n_elem + e] = (s0 + s1) + (s2 + s3);
}
__global__ void __launch_bounds__(256) mlp_partials_apply_kernel(int n_elem, int n_w, const float *__restrict__ chunk, float *__restrict__ v_W,
                                                                 float *__restrict__ v_b) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < RED_CHUNKS; ++c) s += chunk[(int64_t)c * n_elem + e];
  if (s == 0.f) return;
  if (e < n_w) atomicAdd(v_W + e, s);
  else if (v_b != nullptr) atomicAdd(v_b + (e - n_w), s);
}

template <int NL, bool BIAS, bool PART, int MODE = BWD_FULL, int THREADS = SPLIT_BWD_THREADS, int LT = NL - 1, int LB = 0>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS / 256, THREADS / 256)))
    mlp_bwd_split_kernel(int64_t B, MlpDesc d, SplitLds sl, int lds_w4, const float *__restrict__ W, const float *__restrict__ in,
                         const float *__restrict__ acts, const float *__restrict__ v_out, float *__restrict__ v_in,
                         float *__restrict__ v_W, float *__restrict__ v_b, int64_t part_stride, const float *__restrict__ mask_acts = nullptr,
                         float *g_img = nullptr) {
  static_assert(LT <= NL - 1 && LB >= 0 && LB <= LT && (LT >= 1 || NL == 1), "layer range");
  static_assert(MODE != BWD_DATA || (LT == NL - 1 && LB == 0), "the data pass has no accumulators to divide");
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  stage_split_bwd(d, sl, W, smem4, MODE != BWD_TANGENT);   // the tangent pass has no use for W_0^T (no v_in)
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  BwdCtx c;
  c.B = B; c.n_tiles = (B + 31) / 32;
  c.in = in; c.acts = acts; c.v_out = v_out; c.v_in = v_in; c.g_img = g_img;
  c.masks = MODE != BWD_FULL ? reinterpret_cast<const uint16_t *>(mask_acts + img_off(d.n_layers - 1, c.n_tiles, 0, 0, 0)) : nullptr;
  c.lds_w = smem4;
  // the transposition blocks are an object of their own: the compiler then knows that they never alias the weight image
  __shared__ __attribute__((aligned(16))) uint2 tr_blocks[MODE == BWD_DATA ? 1 : THREADS / 64][MODE == BWD_DATA ? 1 : TR_WAVE_BYTES / 8];
  c.tb = tr_blocks[MODE == BWD_DATA ? 0 : wave];
  c.lane = lane; c.d_out = d.d_out;
  BwdAcc<NL> acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      acc.w0[a][r] = 0.f; acc.wl[a][r] = 0.f; acc.b[r] = 0.f;
#pragma unroll
      for (int l = 0; l < (NL > 2 ? NL - 2 : 1); ++l) { acc.wh[l][a][0][r] = 0.f; acc.wh[l][a][1][r] = 0.f; }
    }
  }
  constexpr int WAVES = THREADS / 64;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  v16f g[2], x[2];
  unsigned hm[2] = {0u, 0u};
  Split8 gb[4], at[2][2];
  if (MODE != BWD_DATA) load_layer_input<LT>(c, tile, x);
  if (MODE != BWD_FULL) load_masks<(LT >= 1 ? LT : 1)>(c, tile, hm);
  if (LT == NL - 1) {
    load_v_out(c, tile, g[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) { g[1][r] = 0.f; x[1][r] = (NL == 1 || MODE == BWD_DATA) ? 0.f : x[1][r]; if (MODE == BWD_DATA) x[0][r] = 0.f; }
    prep_g<1, MODE != BWD_DATA>(c, g, gb, at);
#pragma unroll
    for (int s = 2; s < 4; ++s) gb[s] = gb[0];
    if (MODE != BWD_DATA) { at[1][0] = at[0][0]; at[1][1] = at[0][1]; }
  } else {   // lower range: the chain's state as the upper range's launch left it
    load_g(c, tile, g);
    prep_g<2, true>(c, g, gb, at);
  }
  for (; tile < c.n_tiles; tile += stride) BwdLayers<NL, BIAS, LT, MODE, LT, LB>::run(c, sl, acc, tile, tile + stride, gb, at, x, hm);
  if (MODE == BWD_DATA) return;
  // ---- the wave's weight-gradient tiles leave: PART = into its own partial buffer (v_W / v_b point at the buffers' base: [wave][blob]),
  //      else one round of atomics on the gradient
  if (PART) {
    const int64_t mine = ((int64_t)blockIdx.x * WAVES + wave) * part_stride;
    v_W += mine;
    if (BIAS) v_b += mine;
  }
  // (only the tiles of the layers LB .. LT: with PART, the two ranges' launches fill disjoint parts of the same per-wave blob)
  if (LB == 0) {
    flush_tile<PART>(acc.w0[0], v_W + d.w_off[0], 32, HID, 0, 0, lane);
    flush_tile<PART>(acc.w0[1], v_W + d.w_off[0], 32, HID, 32, 0, lane);
  }
#pragma unroll
  for (int l = (LB > 1 ? LB : 1); l < (LT < NL - 1 ? LT + 1 : NL - 1); ++l)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) flush_tile<PART>(acc.wh[l - 1][mt][nt], v_W + d.w_off[l], HID, HID, 32 * mt, 32 * nt, lane);
  if (LT == NL - 1) {
    flush_tile<PART>(acc.wl[0], v_W + d.w_off[NL - 1], HID, d.d_out, 0, 0, lane);
    flush_tile<PART>(acc.wl[1], v_W + d.w_off[NL - 1], HID, d.d_out, 0, 32, lane);
  }
  if (BIAS) {
#pragma unroll
    for (int l = LB; l <= LT; ++l) {
      const int O = l == NL - 1 ? d.d_out : HID;
#pragma unroll
      for (int mt = 0; mt < (l == NL - 1 ? 1 : 2); ++mt) {
        if ((lane & 31) != 8 * mt + l) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * mt + d_row(r, lane >> 5);
          if (o < O) { if (PART) v_b[d.b_off[l] + o] = acc.b[r]; else atomicAdd(v_b + d.b_off[l] + o, acc.b[r]); }
        }
      }
    }
  }
}

static size_t split_bwd_lds(const MlpDesc &d, SplitLds *sl, int *lds_w4) {
  int off = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    sl->off4[l] = off;
    const int I = l == 0 ? d.d_in : HID;
    off += (I / 32) * (l == d.n_layers - 1 ? 2 : 4) * 3 * 64;
  }
  *lds_w4 = off;
  return (size_t)off * 16;   // dynamic part; the transposition blocks are static
}

// grid of the one-pass backward: a wave must own enough 32-point tiles to amortise its exit (measured with the atomic exit: 32768 points
// on 256 workgroups = 1 tile per wave took 0.37 ms, all of it the flush)
static unsigned split_bwd_grid(int64_t B) {
  constexpr int waves = SPLIT_BWD_THREADS / 64, min_tiles_per_wave = 8;
  unsigned grid = split_grid(B, waves);
  const int64_t want = ((B + 31) / 32 + (int64_t)waves * min_tiles_per_wave - 1) / ((int64_t)waves * min_tiles_per_wave);
  if ((int64_t)grid > want) grid = (unsigned)(want < 1 ? 1 : want);
  return grid;
}
static int blob_floats(const MlpDesc &d, int *n_w) {
  const int O = d.d_out, nl = d.n_layers;
  *n_w = d.w_off[nl - 1] + O * HID;
  return *n_w + (d.has_bias ? d.b_off[nl - 1] + O : 0);
}
// the hand-over image of the two-range backward: 64 floats per point, rows padded to whole tiles; it sits behind the partial buffers
static size_t g_img_bytes(int64_t B) { return (size_t)((B + 31) / 32) * 2048 * sizeof(float) + 256; }
static size_t partials_bytes(int64_t B, int64_t n_elem) {
  return (size_t)(((int64_t)split_bwd_grid(B) * (SPLIT_BWD_THREADS / 64) + RED_CHUNKS) * n_elem) * sizeof(float) + 256;
}
size_t mlp_bwd_split_ws_bytes(int64_t B, const MlpDesc &d) {
  int n_w;
  const int64_t n_elem = blob_floats(d, &n_w);
  return partials_bytes(B, n_elem) + (d.n_layers == 5 ? g_img_bytes(B) : 0);
}
size_t mlp_bwd_split_ws_bytes_bound(int64_t B, int n_layers) {
  const int64_t n_elem = (int64_t)n_layers * (HID * HID + HID);
  return partials_bytes(B, n_elem) + g_img_bytes(B);
}

// The 5-layer backward as two launches over layer ranges (bwd_layer): layers {4, 3}, then {2, 1, 0}: 7 + 11 accumulator tiles, 428 / 487
// registers, no scratch (one launch: 17 tiles, 124-229 spilled registers).  Measured at the headline batch (0.44 M base rows): one-pass
// backward 261 -> 216 us, recomputing pass 365 -> 303 us, step 200 -> 208 it/s.  The 4-layer net (tcnn topology: 13 tiles, 1-25 spilled)
// stays one launch: split {3, 2} + {1, 0} it compiles without scratch too but its step gets 2 % slower (123 -> 121 it/s: 3 M rows through
// the 256 B/row hand-over image cost more than the few spills).  Without a workspace (ws == NULL) the 5-layer net runs as one launch too.
template <int NL, bool BIAS, bool PART, int MODE>
static int launch_bwd_kernels(unsigned grid, size_t lds, float *g_img, int64_t B, const MlpDesc &d, const SplitLds &sl, int lds_w4, const float *W,
                              const float *in, const float *acts, const float *v_out, float *v_in, float *v_W, float *v_b, int64_t part_stride,
                              const float *mask_acts, hipStream_t stream) {
  if constexpr (NL == 5) {
    if (g_img != nullptr) {
      constexpr int RL = 3;
      auto upper = mlp_bwd_split_kernel<NL, BIAS, PART, MODE, SPLIT_BWD_THREADS, NL - 1, RL>;
      auto lower = mlp_bwd_split_kernel<NL, BIAS, PART, MODE, SPLIT_BWD_THREADS, RL - 1, 0>;
      GSDF_HIP(hipFuncSetAttribute((const void *)upper, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd_split attr");
      GSDF_HIP(hipFuncSetAttribute((const void *)lower, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd_split attr");
      upper<<<grid, SPLIT_BWD_THREADS, lds, stream>>>(B, d, sl, lds_w4, W, in, acts, v_out, v_in, v_W, v_b, part_stride, mask_acts, g_img);
      GSDF_CHECK_LAUNCH("mlp_bwd_split_kernel (upper layers)");
      lower<<<grid, SPLIT_BWD_THREADS, lds, stream>>>(B, d, sl, lds_w4, W, in, acts, v_out, v_in, v_W, v_b, part_stride, mask_acts, g_img);
      GSDF_CHECK_LAUNCH("mlp_bwd_split_kernel (lower layers)");
      return 1;
    }
  }
  auto whole = mlp_bwd_split_kernel<NL, BIAS, PART, MODE>;
  GSDF_HIP(hipFuncSetAttribute((const void *)whole, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd_split attr");
  whole<<<grid, SPLIT_BWD_THREADS, lds, stream>>>(B, d, sl, lds_w4, W, in, acts, v_out, v_in, v_W, v_b, part_stride, mask_acts, nullptr);
  GSDF_CHECK_LAUNCH("mlp_bwd_split_kernel");
  return 1;
}

template <int NL, bool BIAS, int MODE = BWD_FULL>
static int launch_bwd_split(int64_t B, const MlpDesc &d, const SplitLds &sl, int lds_w4, size_t lds, const float *W, const float *in,
                            const float *acts, const float *v_out, float *v_in, float *v_W, float *v_b, void *ws, hipStream_t stream,
                            const float *mask_acts = nullptr) {
  const unsigned grid = split_bwd_grid(B);
  // beyond ~48 tiles per wave the waves drift apart and their atomic exits hide behind each other's tiles; the partial buffers then only
  // add their two reduction launches (3.29 M points: 1.95 ms atomic, 2.05 ms partial; 0.49 M: 0.55 / 0.36; 0.1 M: 0.24 / 0.15)
  const bool many_tiles = (B + 31) / 32 > (int64_t)grid * (SPLIT_BWD_THREADS / 64) * 48;
  int n_w;
  const int n_elem = blob_floats(d, &n_w);
  const int n_part = (int)grid * (SPLIT_BWD_THREADS / 64);
  float *part = nullptr, *chunk = nullptr, *g_img = nullptr;   // the workspace: per-wave partial blobs, their chunk sums, the two ranges' hand-over image
  if (ws != nullptr) {
    part = (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    chunk = part + (int64_t)n_part * n_elem;
    if (NL == 5) g_img = (float *)(((uintptr_t)(chunk + (int64_t)RED_CHUNKS * n_elem) + 255) & ~(uintptr_t)255);
  }
  GSDF_REQUIRE(ws != nullptr || !deterministic(), "mlp_bwd (one pass): the deterministic mode needs the workspace of the per-wave partial buffers");
  if (ws == nullptr || (many_tiles && !deterministic()))   // (deterministic mode: no atomic exit whatever the batch size)
    return launch_bwd_kernels<NL, BIAS, false, MODE>(grid, lds, g_img, B, d, sl, lds_w4, W, in, acts, v_out, v_in, v_W, v_b, 0, mask_acts, stream);
  int rc = launch_bwd_kernels<NL, BIAS, true, MODE>(grid, lds, g_img, B, d, sl, lds_w4, W, in, acts, v_out, v_in, part, part + n_w, n_elem, mask_acts, stream);
  if (rc < 0) return rc;
  const unsigned eb = (unsigned)((n_elem + 255) / 256);
  mlp_partials_sum_kernel<<<dim3(eb, RED_CHUNKS), 256, 0, stream>>>(n_part, n_elem, part, chunk);
  GSDF_CHECK_LAUNCH("mlp_partials_sum_kernel");
  mlp_partials_apply_kernel<<<eb, 256, 0, stream>>>(n_elem, n_w, chunk, v_W, BIAS ? v_b : nullptr);
  GSDF_CHECK_LAUNCH("mlp_partials_apply_kernel");
  return 1;
}

bool mlp_bwd_split_covers(const MlpDesc &d) {
  if (!split_enabled() || d.d_in != 32 || d.d_out > 16 || (d.n_layers != 4 && d.n_layers != 5)) return false;
  SplitLds sl;
  int lds_w4;
  return split_bwd_lds(d, &sl, &lds_w4) + (size_t)(SPLIT_BWD_THREADS / 64) * TR_WAVE_BYTES <= 160 * 1024;
}

int mlp_bwd_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *in, const float *acts, const float *v_out,
                         float *v_in, float *v_W, float *v_b, void *ws, hipStream_t stream) {
  if (v_W == nullptr || !mlp_bwd_split_covers(d)) return 0;
  SplitLds sl;
  int lds_w4;
  const size_t lds = split_bwd_lds(d, &sl, &lds_w4);
  const bool bias = d.has_bias && v_b != nullptr;
  if (d.n_layers == 5) return bias ? launch_bwd_split<5, true>(B, d, sl, lds_w4, lds, W, in, acts, v_out, v_in, v_W, v_b, ws, stream)
                                   : launch_bwd_split<5, false>(B, d, sl, lds_w4, lds, W, in, acts, v_out, v_in, v_W, v_b, ws, stream);
  if (d.n_layers == 4) return bias ? launch_bwd_split<4, true>(B, d, sl, lds_w4, lds, W, in, acts, v_out, v_in, v_W, v_b, ws, stream)
                                   : launch_bwd_split<4, false>(B, d, sl, lds_w4, lds, W, in, acts, v_out, v_in, v_W, v_b, ws, stream);
  return 0;
}

// v_in only (the chain, masks instead of activations, nothing saved): 8 waves per workgroup, two per SIMD
int mlp_bwd_data_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *acts, const float *v_out, float *v_in, hipStream_t stream) {
  if (v_in == nullptr || !mlp_bwd_split_covers(d)) return 0;
  SplitLds sl;
  int lds_w4;
  const size_t lds = split_bwd_lds(d, &sl, &lds_w4);
  constexpr int THREADS = 512;
  const unsigned grid = split_grid(B, THREADS / 64);
#define LAUNCH_DATA(NL)                                                                                                                  \
  {                                                                                                                                      \
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_bwd_split_kernel<NL, false, false, BWD_DATA, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), \
             "mlp_bwd_data_split attr");                                                                                                 \
    mlp_bwd_split_kernel<NL, false, false, BWD_DATA, THREADS><<<grid, THREADS, lds, stream>>>(B, d, sl, lds_w4, W, nullptr, nullptr, v_out, v_in, nullptr, nullptr, \
                                                                                               0, acts);                                 \
  }
  if (d.n_layers == 5) LAUNCH_DATA(5) else LAUNCH_DATA(4)
#undef LAUNCH_DATA
  GSDF_CHECK_LAUNCH("mlp_bwd_split_kernel<data>");
  return 1;
}

// the double backward's weight term with the chain recomputed (BWD_TANGENT): in = vv_in, tangent = the masked forward's images
int mlp_bwd_tangent_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *vv_in, const float *tangent, const float *mask_acts,
                                 const float *v_out, float *g_W, void *ws, hipStream_t stream) {
  if (g_W == nullptr || !mlp_bwd_split_covers(d)) return 0;
  SplitLds sl;
  int lds_w4;
  const size_t lds = split_bwd_lds(d, &sl, &lds_w4);
  if (d.n_layers == 5) return launch_bwd_split<5, false, BWD_TANGENT>(B, d, sl, lds_w4, lds, W, vv_in, tangent, v_out, nullptr, g_W, nullptr, ws, stream, mask_acts);
  return launch_bwd_split<4, false, BWD_TANGENT>(B, d, sl, lds_w4, lds, W, vv_in, tangent, v_out, nullptr, g_W, nullptr, ws, stream, mask_acts);
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
  const int64_t cap = 256;   // the image fills most of a CU's LDS: one workgroup per CU
  const int64_t wg = ((B + 31) / 32 + waves - 1) / waves;
  return (unsigned)(wg < 1 ? 1 : (wg > cap ? cap : wg));
}

// the tangent pass of the double backward: masked, bias-free forward of vv_in; images of t_0 .. t_{n-2} into `tangent`
int mlp_fwd_masked_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *vv_in, float *out, float *tangent, const float *mask_acts,
                                hipStream_t stream) {
  if (!split_enabled() || d.d_in != 32) return 0;
  SplitLds sl;
  const size_t lds = split_fwd_lds(d, &sl);
  if (lds > 160 * 1024) return 0;
  const unsigned grid = split_grid(B, SPLIT_FWD_THREADS / 64);
  if (d.d_out <= 4) {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_masked_split attr");
    mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS, true, true><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, nullptr, vv_in, out, tangent, mask_acts);
  } else {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_masked_split attr");
    mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS, true><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, nullptr, vv_in, out, tangent, mask_acts);
  }
  GSDF_CHECK_LAUNCH("mlp_fwd_split_kernel<masked>");
  return 1;
}

// returns 1 if launched, 0 if this path does not cover the call (the caller falls back to the fp32 pipe), < 0 on error
int mlp_fwd_split_launch(int64_t B, const MlpDesc &d, const float *W, const float *bias, const float *in, float *out, float *acts,
                         hipStream_t stream) {
  if (!split_enabled()) return 0;
  SplitLds sl;
  const size_t lds = split_fwd_lds(d, &sl);
  if (lds > 160 * 1024) return 0;
  // (measured: 12 and 16 waves per workgroup — 3 / 4 waves per SIMD, a few spilled registers — run 0.72 and 0.90 ms against
  //  0.70 ms at 8; without the saved activations all three take 0.52 ms: occupancy is not what bounds this kernel; round 5: 4 waves per
  //  workgroup — one per SIMD, so that the compositing backward's waves could share the CU in the two-stream step — 203 against 209 it/s)
  const unsigned grid = split_grid(B, SPLIT_FWD_THREADS / 64);
  if (d.d_in == 32 && d.d_out <= 4) {   // the SDF head: output layer on the vector pipe
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_split attr");
    mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS, false, true><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, bias, in, out, acts);
  } else if (d.d_in == 32) {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_split attr");
    mlp_fwd_split_kernel<32, SPLIT_FWD_THREADS><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, bias, in, out, acts);
  } else {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_split_kernel<64, SPLIT_FWD_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd_split attr");
    mlp_fwd_split_kernel<64, SPLIT_FWD_THREADS><<<grid, SPLIT_FWD_THREADS, lds, stream>>>(B, d, sl, W, bias, in, out, acts);
  }
  GSDF_CHECK_LAUNCH("mlp_fwd_split_kernel");
  return 1;
}

}  // namespace gsdf
