// image_loss.hip — O2: fused photometric loss  L = w_l1 * mean|I - G| + w_ssim * (1 - mean SSIM(I, G)), forward and
// backward, as the reference's trainer evaluates it every iteration on the rendered colour image
// (/root/reference/include/neural_mapping/neural_mapping.cpp:237-240; loss::rgb_loss / dssim_loss
// include/optimizer/loss/loss.cpp:22-47; loss_utils::ssim include/optimizer/loss_utils/loss_utils.cpp:71-117:
// five depthwise 11x11 conv2d with zero padding 5 on x, y, x^2, y^2, xy + the SSIM map, ~40 libtorch launches fwd+bwd).
// The 11-tap window is an argument: the reference's gaussian() (loss_utils.cpp:6-14) is NOT the symmetric Gaussian
// (floor((x-11)/2)), the host mirror reproduces it exactly.
//
// MI355X mapping (round 5): workgroup = 32x16 pixel tile, 256 lanes x 2 pixels, ONE channel at a time through LDS (26x42 halo tile ->
// 26x32 row pass -> 16x32 column pass), the other channels' results waiting in registers.  Footprint: 26 KB (forward) / 24 KB (backward) of
// LDS and <= 64 VGPRs, 4 waves per workgroup — so that a workgroup fits into whatever a retiring workgroup of the OTHER leg's kernel frees on a
// CU.  Round 4's kernels (71 / 82 KB of LDS, the backward 512 lanes x 209 VGPRs) needed whole CUs: beside the hash-grid forward the backward
// (0.11 ms alone) waited 1.9 ms for them, on the splat leg's critical chain (profiles/r04_bench_cfg3_step_timeline.txt).
// HBM side: forward reads 2 x 4 B, writes 3 x 4 B of saved partial-derivative maps per pixel-channel; backward reads 5 x 4 B, writes 4 B
// (halo re-reads, 2.1x, are served by the L2).
#include "common.h"

namespace gsdf {

static constexpr int IL_TX = 32, IL_TY = 16, IL_R = 5, IL_HX = IL_TX + 2 * IL_R, IL_HY = IL_TY + 2 * IL_R;  // tile, radius, halo tile
struct Win11 { float w[11]; };

__device__ __forceinline__ float il_at(const float *__restrict__ img, int H, int W, int y, int x, int c) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[((int64_t)y * W + x) * 3 + c] : 0.f;
}

static __device__ DetScalarSlot g_det_l1_dssim;   // deterministic mode: the ordered finish of the two sums
__global__ void __launch_bounds__(256, 6)
    l1_dssim_fwd_kernel(int H, int W, const float *__restrict__ img, const float *__restrict__ gt, Win11 win,
                        float *__restrict__ sums, float *__restrict__ maps, bool det) {
  __shared__ float sx[IL_HY][IL_HX + 1], sy[IL_HY][IL_HX + 1];
  __shared__ float row[5][IL_HY][IL_TX + 1];
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  // the two sums are ONE line: capped grid + tile loop, two atomics per workgroup instead of per tile (atomics on one line serialise at
  // ~88 per microsecond)
  const int tiles_x = (W + IL_TX - 1) / IL_TX, n_tiles = tiles_x * ((H + IL_TY - 1) / IL_TY);
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float l1 = 0.f, ss = 0.f;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int x0 = (tile % tiles_x) * IL_TX, y0 = (tile / tiles_x) * IL_TY;
    const int64_t P3 = (int64_t)H * W * 3;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {      // rolled: one channel's registers live at a time
      __syncthreads();   // the previous channel's (tile's) readers of sx / sy / row are done
#pragma unroll 1
      for (int e = tid; e < IL_HY * IL_HX; e += 256) {
        const int yy = e / IL_HX, xx = e - yy * IL_HX;
        sx[yy][xx] = il_at(img, H, W, y0 + yy - IL_R, x0 + xx - IL_R, c);
        sy[yy][xx] = il_at(gt, H, W, y0 + yy - IL_R, x0 + xx - IL_R, c);
      }
      __syncthreads();
#pragma unroll 1
      for (int e = tid; e < IL_HY * IL_TX; e += 256) {  // horizontal pass
        const int yy = e / IL_TX, xx = e - yy * IL_TX;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const float wv = win.w[k], u = sx[yy][xx + k], v = sy[yy][xx + k];
          a0 += wv * u; a1 += wv * v; a2 += wv * u * u; a3 += wv * v * v; a4 += wv * u * v;
        }
        row[0][yy][xx] = a0; row[1][yy][xx] = a1; row[2][yy][xx] = a2; row[3][yy][xx] = a3; row[4][yy][xx] = a4;
      }
      __syncthreads();
#pragma unroll 1
      for (int j = 0; j < 2; ++j) {  // vertical pass + SSIM: pixel e = tid + 256 j of the tile
        const int e = tid + 256 * j;
        const int yy = e / IL_TX, xx = e - yy * IL_TX;
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
          l1 += fabsf(sx[yy + IL_R][xx + IL_R] - sy[yy + IL_R][xx + IL_R]);
          if (maps != nullptr) {
            const int64_t p = ((int64_t)(y0 + yy) * W + (x0 + xx)) * 3 + c;
            maps[p] = (2.f * mu2 * A2 - 2.f * mu2 * A1) * iB - S * (2.f * mu1 / B1 - 2.f * mu1 / B2);  // dS/dmu1
            maps[P3 + p] = -S / B2;                                                                     // dS/dE[x^2]
            maps[2 * P3 + p] = 2.f * A1 * iB;                                                           // dS/dE[xy]
          }
        }
      }
    }
  }   // tile loop
  for (int s = 32; s >= 1; s >>= 1) { l1 += __shfl_xor(l1, s, 64); ss += __shfl_xor(ss, s, 64); }
  __syncthreads();
  if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ss; }
  __syncthreads();
  finish_scalars(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3], sums, sums + 1,
                 det ? &g_det_l1_dssim : nullptr);
}

// v_img(p) = g_l1 * sign(I - G) + g_ss * [ T(dmu1) + 2 I T(dE11) + G T(dE12) ](p),  T = correlation with the FLIPPED window.
// Same tiling as the forward: the three saved planes of ONE channel of the halo tile in LDS at a time.
__global__ void __launch_bounds__(256, 6)
    l1_dssim_bwd_kernel(int H, int W, const float *__restrict__ img, const float *__restrict__ gt, Win11 win,
                        const float *__restrict__ maps, const float *__restrict__ v_loss, float w_l1, float w_ssim,
                        float *__restrict__ v_img) {
  __shared__ float sm[3][IL_HY][IL_HX + 1];    // [map]
  __shared__ float row[3][IL_HY][IL_TX + 1];
  const int x0 = blockIdx.x * IL_TX, y0 = blockIdx.y * IL_TY, tid = threadIdx.x;
  const int64_t P3 = (int64_t)H * W * 3;
  const float n = 1.0f / (float)P3, vl = *v_loss;
  const float g_l1 = vl * w_l1 * n, g_ss = -vl * w_ssim * n;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {      // rolled: one channel's registers live at a time (<= 64 VGPRs)
    if (c) __syncthreads();
#pragma unroll 1
    for (int e = tid; e < IL_HY * IL_HX; e += 256) {
      const int yy = e / IL_HX, xx = e - yy * IL_HX;
      const int gy = y0 + yy - IL_R, gx = x0 + xx - IL_R;
      const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
      const int64_t p = ((int64_t)gy * W + gx) * 3 + c;
      sm[0][yy][xx] = in ? maps[p] : 0.f; sm[1][yy][xx] = in ? maps[P3 + p] : 0.f; sm[2][yy][xx] = in ? maps[2 * P3 + p] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int e = tid; e < IL_HY * IL_TX; e += 256) {
      const int yy = e / IL_TX, xx = e - yy * IL_TX;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wv = win.w[10 - k];
        a0 += wv * sm[0][yy][xx + k]; a1 += wv * sm[1][yy][xx + k]; a2 += wv * sm[2][yy][xx + k];
      }
      row[0][yy][xx] = a0; row[1][yy][xx] = a1; row[2][yy][xx] = a2;
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 256 * j;
      const int yy = e / IL_TX, xx = e - yy * IL_TX;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float wv = win.w[10 - k];
        t0 += wv * row[0][yy + k][xx]; t1 += wv * row[1][yy + k][xx]; t2 += wv * row[2][yy + k][xx];
      }
      const int gy = y0 + yy, gx = x0 + xx;
      if (gy < H && gx < W) {
        const int64_t p = ((int64_t)gy * W + gx) * 3 + c;
        const float u = img[p], v = gt[p], d = u - v;
        v_img[p] = g_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + g_ss * (t0 + 2.f * u * t1 + v * t2);
      }
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
  const int n_tiles = ((width + IL_TX - 1) / IL_TX) * ((height + IL_TY - 1) / IL_TY);
  l1_dssim_fwd_kernel<<<n_tiles < 1024 ? n_tiles : 1024, 256, 0, stream>>>(height, width, img, gt, w, sums, maps, deterministic());
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
  dim3 grid((width + IL_TX - 1) / IL_TX, (height + IL_TY - 1) / IL_TY, 1);
  l1_dssim_bwd_kernel<<<grid, 256, 0, stream>>>(height, width, img, gt, w, maps, v_loss, w_l1, w_ssim, v_img);
  GSDF_CHECK_LAUNCH("l1_dssim_bwd_kernel");
  return GSDF_OK;
}
