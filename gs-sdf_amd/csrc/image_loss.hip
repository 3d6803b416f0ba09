// image_loss.hip — O2: fused photometric loss  L = w_l1 * mean|I - G| + w_ssim * (1 - mean SSIM(I, G)), forward and
// backward, as the reference's trainer evaluates it every iteration on the rendered colour image
// (/root/reference/include/neural_mapping/neural_mapping.cpp:237-240; loss::rgb_loss / dssim_loss
// include/optimizer/loss/loss.cpp:22-47; loss_utils::ssim include/optimizer/loss_utils/loss_utils.cpp:71-117:
// five depthwise 11x11 conv2d with zero padding 5 on x, y, x^2, y^2, xy + the SSIM map, ~40 libtorch launches fwd+bwd).
// The 11-tap window is an argument: the reference's gaussian() (loss_utils.cpp:6-14) is NOT the symmetric Gaussian
// (floor((x-11)/2)), the host mirror reproduces it exactly.
//
// MI355X mapping: workgroup = 32x32 pixel tile of one channel, separable convolution staged through LDS
// (42x42 input halo tile -> 42x32 row pass -> 32x32 column pass), 256 lanes x 4 pixels.  HBM-bound: forward reads
// 2 x 4 B, writes 3 x 4 B of saved partial-derivative maps per pixel-channel; backward reads 5 x 4 B, writes 4 B.
#include "common.h"

namespace gsdf {

static constexpr int IL_T = 32, IL_R = 5, IL_H = IL_T + 2 * IL_R;  // tile, radius, halo tile
struct Win11 { float w[11]; };

__device__ __forceinline__ float il_at(const float *__restrict__ img, int H, int W, int y, int x, int c) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[((int64_t)y * W + x) * 3 + c] : 0.f;
}

// One workgroup = one 32x32 pixel tile, ALL three channels: the halo tile is fetched once as contiguous rows of
// interleaved RGB (42 px x 12 B) instead of three stride-3 passes by three workgroups, and every pixel's nine saved
// derivatives leave as three contiguous float3.
__global__ void __launch_bounds__(256)
    l1_dssim_fwd_kernel(int H, int W, const float *__restrict__ img, const float *__restrict__ gt, Win11 win,
                        float *__restrict__ sums, float *__restrict__ maps) {
  __shared__ float sx[3][IL_H][IL_H + 1], sy[3][IL_H][IL_H + 1];
  __shared__ float row[5][IL_H][IL_T + 1];
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  // the two sums are ONE line: capped grid + tile loop, two atomics per workgroup instead of per tile (atomics on one line serialise at
  // ~88 per microsecond, DESIGN 6.2: 4080 of them at 1080p)
  const int tiles_x = (W + IL_T - 1) / IL_T, n_tiles = tiles_x * ((H + IL_T - 1) / IL_T);
  float l1 = 0.f, ss = 0.f;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const int x0 = (tile % tiles_x) * IL_T, y0 = (tile / tiles_x) * IL_T;
  if (tile != (int)blockIdx.x) __syncthreads();   // the previous tile's readers of sx / sy are done
  for (int e = tid; e < IL_H * IL_H * 3; e += 256) {
    const int px = e / 3, c = e - 3 * px;
    const int yy = px / IL_H, xx = px - yy * IL_H;
    sx[c][yy][xx] = il_at(img, H, W, y0 + yy - IL_R, x0 + xx - IL_R, c);
    sy[c][yy][xx] = il_at(gt, H, W, y0 + yy - IL_R, x0 + xx - IL_R, c);
  }
  __syncthreads();
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float keep[4][3][3];  // [pixel of this lane][map][channel]
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (int e = tid; e < IL_H * IL_T; e += 256) {  // horizontal pass
      const int yy = e / IL_T, xx = e - yy * IL_T;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wv = win.w[k], u = sx[c][yy][xx + k], v = sy[c][yy][xx + k];
        a0 += wv * u; a1 += wv * v; a2 += wv * u * u; a3 += wv * v * v; a4 += wv * u * v;
      }
      row[0][yy][xx] = a0; row[1][yy][xx] = a1; row[2][yy][xx] = a2; row[3][yy][xx] = a3; row[4][yy][xx] = a4;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // vertical pass + SSIM: pixel e = tid + 256 j of the tile
      const int e = tid + 256 * j;
      const int yy = e / IL_T, xx = e - yy * IL_T;
      float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wv = win.w[k];
        mu1 += wv * row[0][yy + k][xx]; mu2 += wv * row[1][yy + k][xx]; e11 += wv * row[2][yy + k][xx];
        e22 += wv * row[3][yy + k][xx]; e12 += wv * row[4][yy + k][xx];
      }
      const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
      const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
      const float iB = 1.f / (B1 * B2), S = A1 * A2 * iB;
      if (y0 + yy < H && x0 + xx < W) {
        ss += S;
        l1 += fabsf(sx[c][yy + IL_R][xx + IL_R] - sy[c][yy + IL_R][xx + IL_R]);
      }
      keep[j][0][c] = (2.f * mu2 * A2 - 2.f * mu2 * A1) * iB - S * (2.f * mu1 / B1 - 2.f * mu1 / B2);  // dS/dmu1
      keep[j][1][c] = -S / B2;                                                                         // dS/dE[x^2]
      keep[j][2][c] = 2.f * A1 * iB;                                                                   // dS/dE[xy]
    }
    __syncthreads();  // `row` is reused by the next channel
  }
  if (maps != nullptr) {
    const int64_t P3 = (int64_t)H * W * 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j;
      const int gy = y0 + e / IL_T, gx = x0 + (e & (IL_T - 1));
      if (gy >= H || gx >= W) continue;
      const int64_t p = ((int64_t)gy * W + gx) * 3;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        float *dst = maps + m * P3 + p;
        dst[0] = keep[j][m][0]; dst[1] = keep[j][m][1]; dst[2] = keep[j][m][2];
      }
    }
  }
  }   // tile loop
  for (int s = 32; s >= 1; s >>= 1) { l1 += __shfl_xor(l1, s, 64); ss += __shfl_xor(ss, s, 64); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ss; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(sums, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(sums + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// v_img(p) = g_l1 * sign(I - G) + g_ss * [ T(dmu1) + 2 I T(dE11) + G T(dE12) ](p),  T = correlation with the FLIPPED window
// Same tiling as the forward: one workgroup (512 lanes) = one 32x32 tile, all three channels; the nine saved planes
// (3 maps x 3 channels) of the halo tile are fetched once with contiguous row reads.
static constexpr int ILB_THREADS = 512;
__global__ void __launch_bounds__(ILB_THREADS)
    l1_dssim_bwd_kernel(int H, int W, const float *__restrict__ img, const float *__restrict__ gt, Win11 win,
                        const float *__restrict__ maps, const float *__restrict__ v_loss, float w_l1, float w_ssim,
                        float *__restrict__ v_img) {
  extern __shared__ __attribute__((aligned(16))) float il_smem[];
  float (*sm)[3][IL_H][IL_H + 1] = reinterpret_cast<float (*)[3][IL_H][IL_H + 1]>(il_smem);             // [map][ch]
  float (*row)[IL_H][IL_T + 1] = reinterpret_cast<float (*)[IL_H][IL_T + 1]>(il_smem + 9 * IL_H * (IL_H + 1));  // [map]
  const int x0 = blockIdx.x * IL_T, y0 = blockIdx.y * IL_T, tid = threadIdx.x;
  const int64_t P3 = (int64_t)H * W * 3;
  for (int e = tid; e < IL_H * IL_H * 3; e += ILB_THREADS) {
    const int px = e / 3, c = e - 3 * px;
    const int yy = px / IL_H, xx = px - yy * IL_H;
    const int gy = y0 + yy - IL_R, gx = x0 + xx - IL_R;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
    const int64_t p = ((int64_t)gy * W + gx) * 3 + c;
    sm[0][c][yy][xx] = in ? maps[p] : 0.f; sm[1][c][yy][xx] = in ? maps[P3 + p] : 0.f; sm[2][c][yy][xx] = in ? maps[2 * P3 + p] : 0.f;
  }
  __syncthreads();
  float t[2][3][3];  // [pixel of this lane][map][channel]
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (int e = tid; e < IL_H * IL_T; e += ILB_THREADS) {
      const int yy = e / IL_T, xx = e - yy * IL_T;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wv = win.w[10 - k];
        a0 += wv * sm[0][c][yy][xx + k]; a1 += wv * sm[1][c][yy][xx + k]; a2 += wv * sm[2][c][yy][xx + k];
      }
      row[0][yy][xx] = a0; row[1][yy][xx] = a1; row[2][yy][xx] = a2;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + ILB_THREADS * j;
      const int yy = e / IL_T, xx = e - yy * IL_T;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wv = win.w[10 - k];
        t0 += wv * row[0][yy + k][xx]; t1 += wv * row[1][yy + k][xx]; t2 += wv * row[2][yy + k][xx];
      }
      t[j][0][c] = t0; t[j][1][c] = t1; t[j][2][c] = t2;
    }
    __syncthreads();
  }
  const float n = 1.0f / (float)P3, vl = *v_loss;
  const float g_l1 = vl * w_l1 * n, g_ss = -vl * w_ssim * n;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + ILB_THREADS * j;
    const int gy = y0 + e / IL_T, gx = x0 + (e & (IL_T - 1));
    if (gy >= H || gx >= W) continue;
    const int64_t p = ((int64_t)gy * W + gx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float u = img[p + c], v = gt[p + c], d = u - v;
      v_img[p + c] = g_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + g_ss * (t[j][0][c] + 2.f * u * t[j][1][c] + v * t[j][2][c]);
    }
  }
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_l1_dssim_fwd(int height, int width, const float *img, const float *gt, const float *window11_host,
                                 float *sums, float *maps, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_l1_dssim_fwd");
  GSDF_REQUIRE(height > 0 && width > 0, "l1_dssim_fwd: bad image size");
  GSDF_REQUIRE(img && gt && window11_host && sums, "l1_dssim_fwd: null buffer");
  Win11 w;
  for (int k = 0; k < 11; ++k) w.w[k] = window11_host[k];
  GSDF_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(float), stream), "l1_dssim_fwd memset");
  const int n_tiles = ((width + IL_T - 1) / IL_T) * ((height + IL_T - 1) / IL_T);
  l1_dssim_fwd_kernel<<<n_tiles < 512 ? n_tiles : 512, 256, 0, stream>>>(height, width, img, gt, w, sums, maps);
  GSDF_CHECK_LAUNCH("l1_dssim_fwd_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_l1_dssim_bwd(int height, int width, const float *img, const float *gt, const float *window11_host,
                                 const float *maps, const float *v_loss, float w_l1, float w_ssim, float *v_img,
                                 gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_l1_dssim_bwd");
  GSDF_REQUIRE(height > 0 && width > 0, "l1_dssim_bwd: bad image size");
  GSDF_REQUIRE(img && gt && window11_host && maps && v_loss && v_img, "l1_dssim_bwd: null buffer");
  Win11 w;
  for (int k = 0; k < 11; ++k) w.w[k] = window11_host[k];
  dim3 grid((width + IL_T - 1) / IL_T, (height + IL_T - 1) / IL_T, 1);
  const size_t lds = (size_t)(9 * IL_H * (IL_H + 1) + 3 * IL_H * (IL_T + 1)) * sizeof(float);   // 81.6 KB
  GSDF_HIP(hipFuncSetAttribute((const void *)l1_dssim_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
           "l1_dssim_bwd attr");
  l1_dssim_bwd_kernel<<<grid, ILB_THREADS, lds, stream>>>(height, width, img, gt, w, maps, v_loss, w_l1, w_ssim, v_img);
  GSDF_CHECK_LAUNCH("l1_dssim_bwd_kernel");
  return GSDF_OK;
}
