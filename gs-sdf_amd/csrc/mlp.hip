// mlp.hip — S2 / S2': fully fused width-64 ReLU decoder MLP on the fp32 MFMA pipe.
// Replaces tiny-cuda-nn's FullyFusedMLP behind TCNNNetwork::forward
// (/root/reference/include/neural_net/local_map.cpp:44-55, :94) and offers the same fused path for the
// reference's default torch::nn::Sequential topology with biases (local_map.cpp:29-42).
//
// Numerics: v_mfma_f32_32x32x2_f32 — f32 in, f32 accumulate, bit-equal to an fmaf chain — because the
// parity bar is 1e-4 on an SDF whose eikonal/curvature terms amplify error (no bf16/fp16 here).
//
// MI355X mapping (wave64, one 32-point tile per wave, whole network in registers):
//   Y^T = W X^T :  A = weights [out i][k],  B = activations [k][point j],  D[out][point].
//   D's lane layout (col = lane&31 = point, 16 regs = out rows (r&3)+8(r>>2)+4(lane>>5)) is fed straight
//   back as the next layer's B operand by PERMUTING THE K ORDER: k-step s of the next layer pairs the
//   neurons the two half-waves already hold in register s, and the weights are staged into LDS in that
//   permuted order once per workgroup.  No transposes, no LDS round trip of activations, one
//   conflict-free ds_read_b32 per MFMA.  ReLU/bias are applied on the accumulator registers.
//   Weight gradients are a separate "tiny-MN, huge-K" MFMA GEMM (K = points) fed by coalesced row loads.
#include "mlp_common.h"

namespace gsdf {

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
// LDS image of layer l: [o_tile][k_step][64 lanes] with value W[32*o_tile + (lane&31)][perm(k_step, lane>>5)]
__device__ void stage_weights_fwd(const MlpDesc &d, const float *__restrict__ W, const float *__restrict__ bias,
                                  float *lds_w, float *lds_b) {
  for (int l = 0; l < d.n_layers; ++l) {
    const int I = l == 0 ? d.d_in : HID;
    const int O = l == d.n_layers - 1 ? d.d_out : HID;
    const int otiles = l == d.n_layers - 1 ? 1 : 2;
    const int ksteps = I / 2;
    const float *Wl = W + d.w_off[l];
    float *dst = lds_w + d.lds_off[l];
    // four consecutive k-steps of one half-wave are four CONSECUTIVE input neurons (both for the natural order of
    // layer 0 and for perm_hidden), so every lane fetches one float4 of its weight row and scatters it into the
    // image with 4 conflict-free ds_write_b32 (lanes = 32 consecutive output rows)
    const int sq_n = ksteps / 4;
    for (int u = threadIdx.x; u < otiles * sq_n * 64; u += MLP_THREADS) {
      const int ol = u & 31, h = (u >> 5) & 1, sq = (u >> 6) % sq_n, t = (u >> 6) / sq_n;
      const int o = 32 * t + ol;
      const int k0 = l == 0 ? h * ksteps + 4 * sq : perm_hidden(4 * sq, h);
      const float4 v = o < O ? *reinterpret_cast<const float4 *>(Wl + o * I + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      float *d4 = dst + (t * ksteps + 4 * sq) * 64 + 32 * h + ol;
      d4[0] = v.x; d4[64] = v.y; d4[128] = v.z; d4[192] = v.w;
    }
    for (int e = threadIdx.x; e < HID; e += MLP_THREADS)
      lds_b[l * HID + e] = (d.has_bias && e < O) ? bias[d.b_off[l] + e] : 0.f;
  }
}

// MASKED: the ReLUs are replaced by the saved masks of an earlier forward pass (`mask_acts` = that pass's acts buffer) and there
// are no biases: y = W_{n-1} D_{n-2} ... D_0 W_0 x, the forward half of the decoder's DOUBLE backward (gsdf_mlp_bwd_bwd).
template <int D_IN, bool MASKED = false>
__global__ void __launch_bounds__(MLP_THREADS)
    mlp_fwd_kernel(int64_t B, MlpDesc d, const float *__restrict__ W, const float *__restrict__ bias,
                   const float *__restrict__ in, float *__restrict__ out, float *__restrict__ acts,
                   const float *__restrict__ mask_acts = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *lds_b = smem;                         // [MAX_LAYERS][64]
  float *lds_w = smem + MAX_LAYERS * HID;      // permuted weights
  stage_weights_fwd(d, W, bias, lds_w, lds_b);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  constexpr int K0 = D_IN / 2;
  const int64_t n_tiles = (B + 31) / 32;
  uint16_t *masks = acts == nullptr ? nullptr : reinterpret_cast<uint16_t *>(acts + img_off(d.n_layers - 1, n_tiles, 0, 0, 0));
  const uint16_t *msrc = MASKED ? reinterpret_cast<const uint16_t *>(mask_acts + img_off(d.n_layers - 1, n_tiles, 0, 0, 0)) : nullptr;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t p = tile * 32 + pl;
    const bool live = p < B;
    // B operand of layer 0: this lane's K0 input features h*K0 .. h*K0+K0-1 of point p
    float x[K0];
    const float4 *src = reinterpret_cast<const float4 *>(in + (live ? p : 0) * D_IN + h * K0);
#pragma unroll
    for (int q = 0; q < K0 / 4; ++q) {
      const float4 v = live ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    v16f cur[2];
    // ---- layer 0
    {
      const float *w = lds_w + d.lds_off[0];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = lds_b[32 * t + d_row(r, h)];
        float aw[K0];
#pragma unroll
        for (int s = 0; s < K0; ++s) aw[s] = w[(t * K0 + s) * 64 + lane];
#pragma unroll
        for (int s = 0; s < K0; ++s) acc = mfma32(aw[s], x[s], acc);
        if (MASKED) {
          const unsigned m = msrc[mask_off(0, n_tiles, tile, t, lane)];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = (m >> r) & 1u ? acc[r] : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
        }
        cur[t] = acc;
      }
    }
    // ---- hidden layers and output layer
    for (int l = 1; l < d.n_layers; ++l) {
      if (acts != nullptr) {  // post-ReLU activations of layer l-1 as a register image (dead lanes of the last tile too)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float4 *a = reinterpret_cast<float4 *>(acts + img_off(l - 1, n_tiles, tile, t, lane));
          unsigned m = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a[q * (IMG_Q / 4)] = make_float4(cur[t][4 * q], cur[t][4 * q + 1], cur[t][4 * q + 2], cur[t][4 * q + 3]);
#pragma unroll
          for (int r = 0; r < 16; ++r) m |= (cur[t][r] > 0.f ? 1u : 0u) << r;
          masks[mask_off(l - 1, n_tiles, tile, t, lane)] = (uint16_t)m;
        }
      }
      const float *w = lds_w + d.lds_off[l];
      const bool last = l == d.n_layers - 1;
      const int otiles = last ? 1 : 2;
      v16f nxt[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t < otiles) {  // wave-uniform
          v16f acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = lds_b[l * HID + 32 * t + d_row(r, h)];
          float aw[32];  // all A operands of this tile first: the 32 dependent MFMAs then issue back to back
#pragma unroll
          for (int s = 0; s < 32; ++s) aw[s] = w[(t * 32 + s) * 64 + lane];
#pragma unroll
          for (int s = 0; s < 32; ++s) acc = mfma32(aw[s], cur[s >> 4][s & 15], acc);
          if (!last) {
            if (MASKED) {
              const unsigned m = msrc[mask_off(l, n_tiles, tile, t, lane)];
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[r] = (m >> r) & 1u ? acc[r] : 0.f;
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
            }
          }
          nxt[t] = acc;
        }
      }
      cur[0] = nxt[0];
      if (!last) cur[1] = nxt[1];
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

// ----------------------------------------------------------------------------------------------
// backward, data path: v_out -> v_pre of every layer (workspace) -> v_in
// LDS image of layer l: [i_tile][k_step][64 lanes] = W_l[o = perm(k_step, lane>>5)][i = 32*i_tile + (lane&31)]
// ----------------------------------------------------------------------------------------------
__device__ void stage_weights_bwd(const MlpDesc &d, const float *__restrict__ W, float *lds_w) {
  for (int l = 0; l < d.n_layers; ++l) {
    const int I = l == 0 ? d.d_in : HID;
    const int O = l == d.n_layers - 1 ? d.d_out : HID;
    const int itiles = I / 32;
    const int ksteps = l == d.n_layers - 1 ? 16 : 32;  // o padded to 32 on the last layer
    const float *Wl = W + d.w_off[l];
    float *dst = lds_w + d.lds_off[l];
    for (int e = threadIdx.x; e < itiles * ksteps * 64; e += MLP_THREADS) {
      const int lane = e & 63, s = (e >> 6) % ksteps, t = (e >> 6) / ksteps;
      const int o = perm_hidden(s, lane >> 5), i = 32 * t + (lane & 31);
      dst[e] = o < O ? Wl[o * I + i] : 0.f;
    }
  }
}

template <int D_IN>
__global__ void __launch_bounds__(MLP_THREADS)
    mlp_bwd_data_kernel(int64_t B, MlpDesc d, const float *__restrict__ W, const float *__restrict__ acts,
                        const float *__restrict__ v_out, float *__restrict__ v_pre_ws, float *__restrict__ v_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *lds_w = smem;
  stage_weights_bwd(d, W, lds_w);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, pl = lane & 31;
  const int64_t n_tiles = (B + 31) / 32;
  const uint16_t *masks = reinterpret_cast<const uint16_t *>(acts + img_off(d.n_layers - 1, n_tiles, 0, 0, 0));
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t p = tile * 32 + pl;
    const bool live = p < B;
    v16f g[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = d_row(r, h);
      g[0][r] = (live && o < d.d_out) ? v_out[p * d.d_out + o] : 0.f;
      g[1][r] = 0.f;
    }
    for (int l = d.n_layers - 1; l >= 1; --l) {
      const float *w = lds_w + d.lds_off[l];
      const bool last = l == d.n_layers - 1;
      // the ReLU mask of layer l-1 (its saved activations) does not depend on the MFMAs below: issue its loads first so
      // that their latency hides behind the 32-64 MFMAs of this layer
      unsigned hm[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) hm[t] = masks[mask_off(l - 1, n_tiles, tile, t, lane)];
      v16f ng[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (last) {
          float aw[16];
#pragma unroll
          for (int s = 0; s < 16; ++s) aw[s] = w[(t * 16 + s) * 64 + lane];
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = mfma32(aw[s], g[0][s], acc);
        } else {
          float aw[32];
#pragma unroll
          for (int s = 0; s < 32; ++s) aw[s] = w[(t * 32 + s) * 64 + lane];
#pragma unroll
          for (int s = 0; s < 32; ++s) acc = mfma32(aw[s], g[s >> 4][s & 15], acc);
        }
        ng[t] = acc;
      }
      // ReLU mask of layer l-1's output, then persist v_pre_{l-1} for the weight-gradient GEMM
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float4 *vp = reinterpret_cast<float4 *>(v_pre_ws + img_off(l - 1, n_tiles, tile, t, lane));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 o4;   // dead lanes carry g = 0 all the way down: their image rows are zeros
          o4.x = (hm[t] >> (4 * q)) & 1u ? ng[t][4 * q] : 0.f;
          o4.y = (hm[t] >> (4 * q + 1)) & 1u ? ng[t][4 * q + 1] : 0.f;
          o4.z = (hm[t] >> (4 * q + 2)) & 1u ? ng[t][4 * q + 2] : 0.f;
          o4.w = (hm[t] >> (4 * q + 3)) & 1u ? ng[t][4 * q + 3] : 0.f;
          ng[t][4 * q] = o4.x; ng[t][4 * q + 1] = o4.y; ng[t][4 * q + 2] = o4.z; ng[t][4 * q + 3] = o4.w;
          vp[q * (IMG_Q / 4)] = o4;
        }
      }
      g[0] = ng[0]; g[1] = ng[1];
    }
    if (v_in != nullptr) {  // layer 0: v_in = W_0^T v_pre_0
      const float *w = lds_w + d.lds_off[0];
      constexpr int ITILES = D_IN / 32;
#pragma unroll
      for (int t = 0; t < ITILES; ++t) {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float aw[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) aw[s] = w[(t * 32 + s) * 64 + lane];
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = mfma32(aw[s], g[s >> 4][s & 15], acc);
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(v_in + p * D_IN + 32 * t + 8 * q + 4 * h) =
                make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// backward, weight path: v_W_l[o][i] += sum_p v_pre_l[p][o] * h_in_l[p][i]   (K = points)
// grid (k_chunks, n_layers); wave w owns output tile (mt = w>>1, nt = w&1)
// ----------------------------------------------------------------------------------------------
static constexpr int WG_KCHUNK = 512;  // points per workgroup (one round of weight atomics per chunk)

__global__ void __launch_bounds__(MLP_THREADS)
    mlp_bwd_weights_kernel(int64_t B, MlpDesc d, const float *__restrict__ in, const float *__restrict__ acts,
                           const float *__restrict__ v_out, const float *__restrict__ v_pre_ws,
                           float *__restrict__ v_W, float *__restrict__ v_b) {
  const int l = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = wave >> 1, nt = wave & 1;
  const int I = l == 0 ? d.d_in : HID;
  const int O = l == d.n_layers - 1 ? d.d_out : HID;
  if (32 * mt >= O || 32 * nt >= I) return;
  const bool last = l == d.n_layers - 1;
  const int64_t n_tiles = (B + 31) / 32;
  const int n = lane & 31, hh = lane >> 5;
  // A operand (rows = output neurons): v_out rows [p][d_out] for the last layer, else the v_pre image of slot l.
  // B operand (cols = input neurons): the network input rows [p][d_in] for layer 0, else the activation image of slot l-1.
  // For an image operand lane n serves neuron pi(n) = d_row(n & 15, n >> 4) of its 32-neuron tile: the 32 neurons of a
  // point are then two runs of 16 consecutive floats (the registers of the point's two lanes in the image).
  const bool a_img = !last, b_img = l != 0;
  const int o = 32 * mt + (a_img ? d_row(n & 15, n >> 4) : n);   // output neuron this lane supplies as A row n
  const int i = 32 * nt + (b_img ? d_row(n & 15, n >> 4) : n);   // input neuron this lane supplies as B column n
  const bool o_ok = o < O;
  v16f acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * WG_KCHUNK;
  const int64_t p1 = min(B, p0 + WG_KCHUNK);
  // Operand addresses of a 16-point group pb..pb+15 (pb is a multiple of 16): base(pb) + s * stride, s = 0..7, so that the
  // 8 loads of a group are one base computation + immediate offsets.  Image: the group lies in one tile, neuron n of a point
  // is register n & 15 of the point's lane in half n >> 4: element = tile base + IMG_Q ((n & 15) >> 2) + 4 lane + (n & 3)
  // with lane = (pb & 31) + 2 s + hh + 32 (n >> 4).
  const int img_lane0 = IMG_Q * ((n & 15) >> 2) + 4 * (32 * (n >> 4) + hh) + (n & 3);
  const float *a0 = a_img ? v_pre_ws + img_off(l, n_tiles, 0, mt, 0) + img_lane0 : v_out + (int64_t)hh * d.d_out + o;
  const float *b0 = b_img ? acts + img_off(l - 1, n_tiles, 0, nt, 0) + img_lane0 : in + (int64_t)hh * d.d_in + i;
  const int a_stride = a_img ? 8 : 2 * d.d_out, b_stride = b_img ? 8 : 2 * d.d_in;
  auto load16 = [&](int64_t pb, float (&ra)[8], float (&rb)[8]) {
    const int64_t img_base = (pb >> 5) * 2048 + (pb & 31) * 4;
    const float *ap = a0 + (a_img ? img_base : pb * d.d_out);
    const float *bp = b0 + (b_img ? img_base : pb * d.d_in);
    const int64_t left = p1 - pb - hh;   // point pb + 2 s + hh is valid while 2 s < left
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bool ok = 2 * s < left;
      ra[s] = (ok && o_ok) ? ap[s * a_stride] : 0.f;
      rb[s] = ok ? bp[s * b_stride] : 0.f;
    }
  };
  // software pipeline: the operands of the next 16 points are in flight while the 8 MFMAs of the current ones issue
  float a[8], b[8], na[8], nb[8];
  load16(p0, a, b);
  for (int64_t pb = p0; pb < p1; pb += 16) {
    load16(pb + 16, na, nb);   // past p1: predicated off, zeros
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      acc = mfma32(a[s], b[s], acc);
      bsum += a[s];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) { a[s] = na[s]; b[s] = nb[s]; }
  }
  // D[m][n']: row m = d_row(r, hh) is the A row supplied by lane m, column n' = this lane's B column
  float *vw = v_W + d.w_off[l];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = d_row(r, hh);
    const int oo = 32 * mt + (a_img ? d_row(m & 15, m >> 4) : m);
    if (oo < O && acc[r] != 0.f) atomicAdd(vw + oo * I + i, acc[r]);
  }
  if (v_b != nullptr && nt == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (hh == 0 && o_ok) atomicAdd(v_b + d.b_off[l] + o, bsum);
  }
}

static int make_desc(int n_layers, const int *dims, int has_bias, bool bwd, MlpDesc *d, size_t *lds_floats,
                     const char *who) {
  GSDF_REQUIRE(n_layers >= 2 && n_layers <= MAX_LAYERS, "%s: n_layers %d not in [2,%d]", who, n_layers, MAX_LAYERS);
  GSDF_REQUIRE(dims[0] == 32 || dims[0] == 64, "%s: input width %d unsupported (32 or 64)", who, dims[0]);
  GSDF_REQUIRE(dims[n_layers] >= 1 && dims[n_layers] <= 32, "%s: output width %d unsupported (<= 32)", who, dims[n_layers]);
  for (int l = 1; l < n_layers; ++l)
    GSDF_REQUIRE(dims[l] == HID, "%s: hidden width %d unsupported (64 only, as the reference configures)", who, dims[l]);
  d->n_layers = n_layers; d->d_in = dims[0]; d->d_out = dims[n_layers]; d->has_bias = has_bias;
  int w = 0, b = 0, lo = 0;
  for (int l = 0; l < n_layers; ++l) {
    d->w_off[l] = w; d->b_off[l] = b; d->lds_off[l] = lo;
    w += dims[l] * dims[l + 1]; b += dims[l + 1];
    const int I = dims[l];
    const bool last = l == n_layers - 1;
    lo += bwd ? (I / 32) * (last ? 16 : 32) * 64 : (last ? 1 : 2) * (I / 2) * 64;
  }
  *lds_floats = (size_t)lo + (bwd ? 0 : MAX_LAYERS * HID);
  return GSDF_OK;
}

}  // namespace gsdf

using namespace gsdf;

static unsigned mlp_grid(int64_t B) {
  const int64_t cap = 768;   // 3 per CU on 256 CUs, 4 per CU on a 6-XCD queue
  const int64_t wg = (B + 127) / 128;  // 4 tiles of 32 points per workgroup
  return (unsigned)(wg < 1 ? 1 : (wg > cap ? cap : wg));
}

static size_t image_floats(int64_t B, int n_layers) {   // register images, rows padded to whole 32-point tiles
  return (size_t)(n_layers > 1 ? n_layers - 1 : 0) * (size_t)((B + 31) / 32 * 32) * HID;
}
extern "C" size_t gsdf_mlp_acts_floats(int64_t B, int n_layers) {   // images + the uint16 ReLU masks (1/32 of the images)
  return image_floats(B, n_layers) + image_floats(B, n_layers) / 32 + 64;
}
extern "C" size_t gsdf_mlp_bwd_ws_bytes(int64_t B, int n_layers) {
  // the per-layer gradient images of the two-kernel path, or the per-wave partial weight gradients of the one-pass path
  const size_t img = image_floats(B, n_layers) * sizeof(float) + 256, part = mlp_bwd_split_ws_bytes_bound(B, n_layers);
  return img > part ? img : part;
}

extern "C" size_t gsdf_mlp_bwd_ws_bytes_for(int64_t B, int n_layers, const int *dims_host, int want_weights) {
  MlpDesc d;
  size_t lds_floats;
  if (dims_host && want_weights && make_desc(n_layers, dims_host, 0, true, &d, &lds_floats, "mlp_bwd_ws_bytes_for") == GSDF_OK &&
      mlp_bwd_split_covers(d))
    return mlp_bwd_split_ws_bytes_bound(B, n_layers);   // one-pass path: partial weight gradients only (with or without biases)
  return gsdf_mlp_bwd_ws_bytes(B, n_layers);
}

extern "C" int gsdf_mlp_bwd_is_one_pass(int n_layers, const int *dims_host) {
  MlpDesc d;
  size_t lds_floats;
  return dims_host && make_desc(n_layers, dims_host, 0, true, &d, &lds_floats, "mlp_bwd_is_one_pass") == GSDF_OK && mlp_bwd_split_covers(d) ? 1 : 0;
}

extern "C" int gsdf_mlp_fwd(int64_t B, int n_layers, const int *dims_host, const float *weights, const float *biases,
                            const float *in, float *out, float *acts, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_mlp_fwd");
  GSDF_REQUIRE(dims_host, "mlp_fwd: null dims");
  MlpDesc d;
  size_t lds_floats;
  int rc = make_desc(n_layers, dims_host, biases != nullptr, false, &d, &lds_floats, "mlp_fwd");
  if (rc) return rc;
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(weights && in && out, "mlp_fwd: null buffer");
  rc = mlp_fwd_split_launch(B, d, weights, biases, in, out, acts, stream);
  if (rc != 0) return rc < 0 ? rc : GSDF_OK;
  const size_t lds = lds_floats * sizeof(float);
  GSDF_REQUIRE(lds <= 160 * 1024, "mlp_fwd: %zu bytes of weights do not fit the 160 KiB LDS", lds);
  if (d.d_in == 32) {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd attr");
    mlp_fwd_kernel<32><<<mlp_grid(B), MLP_THREADS, lds, stream>>>(B, d, weights, biases, in, out, acts);
  } else {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_fwd attr");
    mlp_fwd_kernel<64><<<mlp_grid(B), MLP_THREADS, lds, stream>>>(B, d, weights, biases, in, out, acts);
  }
  GSDF_CHECK_LAUNCH("mlp_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_mlp_bwd(int64_t B, int n_layers, const int *dims_host, const float *weights, const float *biases,
                            const float *in, const float *acts, const float *v_out, float *v_in, float *v_weights,
                            float *v_biases, void *ws, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED(v_weights != nullptr ? "gsdf_mlp_bwd" : "gsdf_mlp_bwd_data");   // both gradients (one pass) / input gradient only
  GSDF_REQUIRE(dims_host, "mlp_bwd: null dims");
  MlpDesc d;
  size_t lds_floats;
  int rc = make_desc(n_layers, dims_host, biases != nullptr, true, &d, &lds_floats, "mlp_bwd");
  if (rc) return rc;
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(weights && in && v_out, "mlp_bwd: null buffer");
  GSDF_REQUIRE(acts, "mlp_bwd: the saved activations of mlp_fwd are required");
  if (v_weights != nullptr) {   // both gradients wanted: one pass on the bf16 pipe, v_pre never leaves the registers
    rc = mlp_bwd_split_launch(B, d, weights, in, acts, v_out, v_in, v_weights, biases != nullptr ? v_biases : nullptr, ws, stream);
    if (rc != 0) return rc < 0 ? rc : GSDF_OK;
  } else if (ws == nullptr) {   // input gradient only and NOTHING saved (round 4): the chain alone on the bf16 pipe, ReLU masks instead of activations
    rc = mlp_bwd_data_split_launch(B, d, weights, acts, v_out, v_in, stream);
    if (rc != 0) return rc < 0 ? rc : GSDF_OK;
  }
  GSDF_REQUIRE(ws, "mlp_bwd: null workspace (only the topologies gsdf_mlp_bwd_is_one_pass covers run without one)");
  const size_t lds = lds_floats * sizeof(float);
  GSDF_REQUIRE(lds <= 160 * 1024, "mlp_bwd: %zu bytes of weights do not fit the 160 KiB LDS", lds);
  float *v_pre = (float *)ws;
  if (d.d_in == 32) {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_bwd_data_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd attr");
    mlp_bwd_data_kernel<32><<<mlp_grid(B), MLP_THREADS, lds, stream>>>(B, d, weights, acts, v_out, v_pre, v_in);
  } else {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_bwd_data_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd attr");
    mlp_bwd_data_kernel<64><<<mlp_grid(B), MLP_THREADS, lds, stream>>>(B, d, weights, acts, v_out, v_pre, v_in);
  }
  GSDF_CHECK_LAUNCH("mlp_bwd_data_kernel");
  if (v_weights != nullptr) {
    dim3 grid((unsigned)((B + WG_KCHUNK - 1) / WG_KCHUNK), (unsigned)n_layers);
    mlp_bwd_weights_kernel<<<grid, MLP_THREADS, 0, stream>>>(B, d, in, acts, v_out, v_pre, v_weights,
                                                             biases != nullptr ? v_biases : nullptr);
    GSDF_CHECK_LAUNCH("mlp_bwd_weights_kernel");
  }
  return GSDF_OK;
}

extern "C" size_t gsdf_mlp_bwd_bwd_ws_bytes(int64_t B, int n_layers) {   // tangent images, then the per-wave partial weight gradients of the recompute path
  return align_up(gsdf_mlp_acts_floats(B, n_layers) * sizeof(float) + 256, 256) + mlp_bwd_split_ws_bytes_bound(B, n_layers);
}

extern "C" int gsdf_mlp_bwd_bwd(int64_t B, int n_layers, const int *dims_host, const float *weights, const float *acts,
                                const float *v_out, const void *bwd_ws, const float *vv_in, float *g_vout, float *g_weights,
                                void *ws, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_mlp_bwd_bwd");
  GSDF_REQUIRE(dims_host, "mlp_bwd_bwd: null dims");
  MlpDesc d;
  size_t lds_floats;
  int rc = make_desc(n_layers, dims_host, 0, false, &d, &lds_floats, "mlp_bwd_bwd");
  if (rc) return rc;
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(weights && acts && v_out && vv_in && ws, "mlp_bwd_bwd: null buffer");
  GSDF_REQUIRE(g_vout != nullptr, "mlp_bwd_bwd: g_vout [B, d_out] is required (it is the masked forward's output buffer)");
  if (bwd_ws == nullptr) {
    // Round 4, no saved v_pre images: (1) the tangent pass on the bf16 pipe (masked forward of vv_in, images t_0 .. t_{n-2} into ws),
    // (2) ONE pass that recomputes the chain v_out -> v_pre_l in registers from the ReLU masks and accumulates dL/dW_l += v_pre_l (x) t_{l-1}
    // (mlp_split.hip: BWD_TANGENT).  Only for the topologies the one-pass backward covers (gsdf_mlp_bwd_is_one_pass).
    MlpDesc db;
    rc = make_desc(n_layers, dims_host, 0, true, &db, &lds_floats, "mlp_bwd_bwd");
    if (rc) return rc;
    GSDF_REQUIRE(mlp_bwd_split_covers(db), "mlp_bwd_bwd: bwd_ws = NULL (recompute) needs a topology gsdf_mlp_bwd_is_one_pass covers");
    float *tangent = (float *)ws;
    rc = mlp_fwd_masked_split_launch(B, d, weights, vv_in, g_vout, tangent, acts, stream);
    GSDF_REQUIRE(rc != 0, "mlp_bwd_bwd: the masked forward does not cover this topology");
    if (rc < 0) return rc;
    if (g_weights != nullptr) {
      void *part = (char *)ws + align_up(gsdf_mlp_acts_floats(B, n_layers) * sizeof(float) + 256, 256);
      rc = mlp_bwd_tangent_split_launch(B, db, weights, vv_in, tangent, acts, v_out, g_weights, part, stream);
      GSDF_REQUIRE(rc != 0, "mlp_bwd_bwd: the recompute pass does not cover this topology");
      if (rc < 0) return rc;
    }
    return GSDF_OK;
  }
  // (1) masked bias-free forward of vv_in: u_l = D_l W_l u_{l-1} saved as a register image in ws, last layer -> dL/d v_out
  const size_t lds = lds_floats * sizeof(float);
  GSDF_REQUIRE(lds <= 160 * 1024, "mlp_bwd_bwd: %zu bytes of weights do not fit the 160 KiB LDS", lds);
  float *u_img = (float *)ws;
  float *sink = g_vout;
  if (d.d_in == 32) {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd_bwd attr");
    mlp_fwd_kernel<32, true><<<mlp_grid(B), MLP_THREADS, lds, stream>>>(B, d, weights, nullptr, vv_in, sink, u_img, acts);
  } else {
    GSDF_HIP(hipFuncSetAttribute((const void *)mlp_fwd_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "mlp_bwd_bwd attr");
    mlp_fwd_kernel<64, true><<<mlp_grid(B), MLP_THREADS, lds, stream>>>(B, d, weights, nullptr, vv_in, sink, u_img, acts);
  }
  GSDF_CHECK_LAUNCH("mlp_fwd_kernel<masked>");
  // (2) dL/dW_0 = delta_0 (x) vv_in, dL/dW_l = delta_l (x) u_{l-1}: the weight-gradient GEMM of the first order backward with
  //     (network input, activations) := (vv_in, u images) and the first backward's own v_pre images / v_out as the A operand
  if (g_weights != nullptr) {
    MlpDesc db;
    rc = make_desc(n_layers, dims_host, 0, true, &db, &lds_floats, "mlp_bwd_bwd");
    if (rc) return rc;
    dim3 grid((unsigned)((B + WG_KCHUNK - 1) / WG_KCHUNK), (unsigned)n_layers);
    mlp_bwd_weights_kernel<<<grid, MLP_THREADS, 0, stream>>>(B, db, vv_in, u_img, v_out, (const float *)bwd_ws, g_weights, nullptr);
    GSDF_CHECK_LAUNCH("mlp_bwd_weights_kernel<bwd_bwd>");
  }
  return GSDF_OK;
}

extern "C" int gsdf_mlp_bwd_weights(int64_t B, int n_layers, const int *dims_host, int has_biases, const float *in,
                                    const float *acts, const float *v_out, const void *ws, float *v_weights,
                                    float *v_biases, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_mlp_bwd_weights");
  GSDF_REQUIRE(dims_host, "mlp_bwd_weights: null dims");
  MlpDesc d;
  size_t lds_floats;
  int rc = make_desc(n_layers, dims_host, has_biases != 0, true, &d, &lds_floats, "mlp_bwd_weights");
  if (rc) return rc;
  if (B == 0) return GSDF_OK;
  GSDF_REQUIRE(in && acts && v_out && ws && v_weights, "mlp_bwd_weights: null buffer");
  GSDF_REQUIRE(!has_biases || v_biases, "mlp_bwd_weights: null v_biases");
  dim3 grid((unsigned)((B + WG_KCHUNK - 1) / WG_KCHUNK), (unsigned)n_layers);
  mlp_bwd_weights_kernel<<<grid, MLP_THREADS, 0, stream>>>(B, d, in, acts, v_out, (const float *)ws, v_weights,
                                                           has_biases ? v_biases : nullptr);
  GSDF_CHECK_LAUNCH("mlp_bwd_weights_kernel");
  return GSDF_OK;
}
