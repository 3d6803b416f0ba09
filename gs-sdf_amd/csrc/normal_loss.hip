// normal_loss.hip — O3: fused depth->normal + normal-consistency loss (SURVEY 8f "next #3").
// Reference (libtorch ops, ~25 launches fwd+bwd, every iteration after refine_gs_struct_start_iter):
//   sensor::depth_to_normal  /root/reference/include/utils/sensor_utils/cameras.hpp:176-226
//     P = (zdir @ R_c2w^T) * depth + t,  zdir = ((x+.5-cx)/fx, (y+.5-cy)/fy, 1)   (cameras.hpp:15-29)
//     dx = P[i+1,j] - P[i-1,j], dy = P[i,j+1] - P[i,j-1], n = normalize(cross(dx,dy)) on the interior, 0 on the border
//   normal_error = mean( alpha^2 - nan_to_num( (n*alpha) . render_normal ) )       neural_mapping.cpp:243-266
// (alpha is detached there).  One 16x16 pixel tile per workgroup; forward = 5-point stencil on the back-projected
// points; backward gathers v_P(q) = v_dx(q-(1,0)) - v_dx(q+(1,0)) + v_dy(q-(0,1)) - v_dy(q+(0,1)) from a halo-1 ring
// recomputed in LDS (no atomics), v_depth = v_P . dir.  HBM-bound: 20 B read + 16 B written per pixel.
#include "common.h"

namespace gsdf {

struct NlCam { float fx, fy, cx, cy; float R[9]; float t[3]; };  // R = cam->world rotation (row-major), t = position

static constexpr int NL_T = 16;

__device__ __forceinline__ void nl_dir(const NlCam &c, int x, int y, float d[3]) {
  const float zx = ((float)x + 0.5f - c.cx) / c.fx, zy = ((float)y + 0.5f - c.cy) / c.fy;
  d[0] = zx * c.R[0] + zy * c.R[1] + c.R[2];
  d[1] = zx * c.R[3] + zy * c.R[4] + c.R[5];
  d[2] = zx * c.R[6] + zy * c.R[7] + c.R[8];
}

// normal (and its pre-normalisation cross product) at interior pixel from the 4 neighbours' points
struct NlN { float n[3], c[3], len; bool interior; };

__device__ __forceinline__ void nl_normal(const float *pu, const float *pd, const float *pl, const float *pr, NlN &o) {
  const float dx[3] = {pd[0] - pu[0], pd[1] - pu[1], pd[2] - pu[2]};   // P[i+1,j] - P[i-1,j]
  const float dy[3] = {pr[0] - pl[0], pr[1] - pl[1], pr[2] - pl[2]};   // P[i,j+1] - P[i,j-1]
  o.c[0] = dx[1] * dy[2] - dx[2] * dy[1]; o.c[1] = dx[2] * dy[0] - dx[0] * dy[2]; o.c[2] = dx[0] * dy[1] - dx[1] * dy[0];
  o.len = sqrtf(o.c[0] * o.c[0] + o.c[1] * o.c[1] + o.c[2] * o.c[2]);
  const float inv = 1.0f / fmaxf(o.len, 1e-12f);
  o.n[0] = o.c[0] * inv; o.n[1] = o.c[1] * inv; o.n[2] = o.c[2] * inv;
}

static __device__ DetScalarSlot g_det_normal_fwd, g_det_normal_bwd;   // deterministic mode: the ordered finish of the loss value

template <bool BWD>
__global__ void __launch_bounds__(256)
    normal_loss_kernel(int H, int W, NlCam cam, const float *__restrict__ depth, const float *__restrict__ alpha,
                       const float *__restrict__ rnormal, float *__restrict__ sum, const float *__restrict__ v_loss,
                       float *__restrict__ v_depth, float *__restrict__ v_rnormal, bool det) {
  constexpr int HALO = BWD ? 2 : 1, S = NL_T + 2 * HALO;
  __shared__ float P[S][S][3];
  __shared__ float G[BWD ? NL_T + 2 : 1][BWD ? NL_T + 2 : 1][6];  // (v_dx, v_dy) on the halo-1 ring (backward only)
  __shared__ float red[4];
  const int tid = threadIdx.x;
  // The forward's value is ONE address and atomics on one line serialise at ~88 per microsecond: a workgroup per tile meant
  // 8160 of them at 1080p = 93 us of a 108 us kernel.  Capped grid + tile loop, one atomic per workgroup (round 4: 108 -> ~25 us).
  const int tiles_x = (W + NL_T - 1) / NL_T, n_tiles = tiles_x * ((H + NL_T - 1) / NL_T);
  float wg_term = 0.f;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const int x0 = (tile % tiles_x) * NL_T, y0 = (tile / tiles_x) * NL_T;
  if (tile != (int)blockIdx.x) __syncthreads();   // the previous tile's readers of P are done
  for (int e = tid; e < S * S; e += 256) {
    const int yy = e / S, xx = e - yy * S, gy = y0 + yy - HALO, gx = x0 + xx - HALO;
    float p[3] = {0.f, 0.f, 0.f};
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
      float d[3];
      nl_dir(cam, gx, gy, d);
      const float z = depth[(int64_t)gy * W + gx];
      p[0] = d[0] * z + cam.t[0]; p[1] = d[1] * z + cam.t[1]; p[2] = d[2] * z + cam.t[2];
    }
    P[yy][xx][0] = p[0]; P[yy][xx][1] = p[1]; P[yy][xx][2] = p[2];
  }
  __syncthreads();
  const float inv_n = 1.0f / ((float)H * (float)W);
  if (!BWD) {
    const int yy = tid / NL_T, xx = tid - yy * NL_T, gy = y0 + yy, gx = x0 + xx;
    float term = 0.f;
    if (gx < W && gy < H) {
      const float a = alpha[(int64_t)gy * W + gx];
      float dot = 0.f;
      if (gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1) {
        NlN nn;
        nl_normal(P[yy][xx + 1], P[yy + 2][xx + 1], P[yy + 1][xx], P[yy + 1][xx + 2], nn);
        const float *rn = rnormal + ((int64_t)gy * W + gx) * 3;
        dot = a * (nn.n[0] * rn[0] + nn.n[1] * rn[1] + nn.n[2] * rn[2]);
        if (dot != dot) dot = 0.f;                                         // nan_to_num
        else dot = fminf(fmaxf(dot, -3.4028234663852886e38f), 3.4028234663852886e38f);
      }
      term = a * a - dot;
    }
    wg_term += term;
  } else {
    const float vl = *v_loss * inv_n;
    // ring: v_dx, v_dy of every pixel of the tile + 1 halo
    for (int e = tid; e < (NL_T + 2) * (NL_T + 2); e += 256) {
      const int yy = e / (NL_T + 2), xx = e - yy * (NL_T + 2), gy = y0 + yy - 1, gx = x0 + xx - 1;
      float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1) {
        // P index of pixel (gy,gx) is [yy+1][xx+1] (halo 2)
        const float *pu = P[yy][xx + 1], *pd = P[yy + 2][xx + 1], *pl = P[yy + 1][xx], *pr = P[yy + 1][xx + 2];
        NlN nn;
        nl_normal(pu, pd, pl, pr, nn);
        const float a = alpha[(int64_t)gy * W + gx];
        const float *rn = rnormal + ((int64_t)gy * W + gx) * 3;
        const float dot = a * (nn.n[0] * rn[0] + nn.n[1] * rn[1] + nn.n[2] * rn[2]);
        const bool fin = (dot == dot) && fabsf(dot) <= 3.4028234663852886e38f;
        if (fin) {
          // d(-dot)/dn = -a rn ; through normalize: v_c = (v_n - (v_n.n) n)/|c|   (clamped norm: zero when |c| < 1e-12)
          float vn[3] = {-vl * a * rn[0], -vl * a * rn[1], -vl * a * rn[2]};
          float vc[3] = {0.f, 0.f, 0.f};
          if (nn.len >= 1e-12f) {
            const float dnn = vn[0] * nn.n[0] + vn[1] * nn.n[1] + vn[2] * nn.n[2], il = 1.0f / nn.len;
            vc[0] = (vn[0] - dnn * nn.n[0]) * il; vc[1] = (vn[1] - dnn * nn.n[1]) * il; vc[2] = (vn[2] - dnn * nn.n[2]) * il;
          } else {
            vc[0] = vn[0] * 1e12f; vc[1] = vn[1] * 1e12f; vc[2] = vn[2] * 1e12f;
          }
          const float dx[3] = {pd[0] - pu[0], pd[1] - pu[1], pd[2] - pu[2]}, dy[3] = {pr[0] - pl[0], pr[1] - pl[1], pr[2] - pl[2]};
          // c = dx x dy : v_dx = dy x v_c, v_dy = v_c x dx
          g6[0] = dy[1] * vc[2] - dy[2] * vc[1]; g6[1] = dy[2] * vc[0] - dy[0] * vc[2]; g6[2] = dy[0] * vc[1] - dy[1] * vc[0];
          g6[3] = vc[1] * dx[2] - vc[2] * dx[1]; g6[4] = vc[2] * dx[0] - vc[0] * dx[2]; g6[5] = vc[0] * dx[1] - vc[1] * dx[0];
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) G[yy][xx][k] = g6[k];
    }
    __syncthreads();
    const int yy = tid / NL_T, xx = tid - yy * NL_T, gy = y0 + yy, gx = x0 + xx;
    if (gx < W && gy < H) {
      // G index of pixel (gy,gx) is [yy+1][xx+1]
      float vp[3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        vp[k] = G[yy][xx + 1][k] - G[yy + 2][xx + 1][k] + G[yy + 1][xx][3 + k] - G[yy + 1][xx + 2][3 + k];
      float d[3];
      nl_dir(cam, gx, gy, d);
      v_depth[(int64_t)gy * W + gx] = vp[0] * d[0] + vp[1] * d[1] + vp[2] * d[2];
      float o[3] = {0.f, 0.f, 0.f};
      if (gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1) {
        NlN nn;
        nl_normal(P[yy + 1][xx + 2], P[yy + 3][xx + 2], P[yy + 2][xx + 1], P[yy + 2][xx + 3], nn);
        const float a = alpha[(int64_t)gy * W + gx];
        const float *rn = rnormal + ((int64_t)gy * W + gx) * 3;
        const float dot = a * (nn.n[0] * rn[0] + nn.n[1] * rn[1] + nn.n[2] * rn[2]);
        if ((dot == dot) && fabsf(dot) <= 3.4028234663852886e38f) { o[0] = -vl * a * nn.n[0]; o[1] = -vl * a * nn.n[1]; o[2] = -vl * a * nn.n[2]; }
      }
      float *vr = v_rnormal + ((int64_t)gy * W + gx) * 3;
      vr[0] = o[0]; vr[1] = o[1]; vr[2] = o[2];
      if (sum != nullptr) {   // the value next to the gradient (gsdf_normal_consistency_fwd_bwd): the forward's term of this pixel
        const float a = alpha[(int64_t)gy * W + gx];
        float dot = 0.f;
        if (gx >= 1 && gx < W - 1 && gy >= 1 && gy < H - 1) {
          NlN nn;
          nl_normal(P[yy + 1][xx + 2], P[yy + 3][xx + 2], P[yy + 2][xx + 1], P[yy + 2][xx + 3], nn);
          const float *rn = rnormal + ((int64_t)gy * W + gx) * 3;
          dot = a * (nn.n[0] * rn[0] + nn.n[1] * rn[1] + nn.n[2] * rn[2]);
          if (dot != dot) dot = 0.f;
          else dot = fminf(fmaxf(dot, -3.4028234663852886e38f), 3.4028234663852886e38f);
        }
        wg_term += a * a - dot;
      }
    }
  }
  }   // tile loop
  if (!BWD || sum != nullptr) {
    for (int s = 32; s >= 1; s >>= 1) wg_term += __shfl_xor(wg_term, s, 64);
    if ((tid & 63) == 0) red[tid >> 6] = wg_term;
    __syncthreads();
    finish_scalars((red[0] + red[1] + red[2] + red[3]) * (1.0f / ((float)H * (float)W)), 0.f, sum, nullptr,
                   det ? (BWD ? &g_det_normal_bwd : &g_det_normal_fwd) : nullptr);
  }
}

static int nl_cam(const float *intr4, const float *pose34, NlCam *c) {
  c->fx = intr4[0]; c->fy = intr4[1]; c->cx = intr4[2]; c->cy = intr4[3];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c->R[3 * i + j] = pose34[4 * i + j]; c->t[i] = pose34[4 * i + 3]; }
  return 0;
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_normal_consistency_fwd(int height, int width, const float *intrinsics4_host, const float *pose_c2w_host,
                                           const float *depth, const float *alpha, const float *render_normal, float *loss,
                                           gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_normal_consistency_fwd");
  GSDF_REQUIRE(height > 2 && width > 2, "normal_consistency_fwd: image must be at least 3x3");
  GSDF_REQUIRE(intrinsics4_host && pose_c2w_host && depth && alpha && render_normal && loss, "normal_consistency_fwd: null buffer");
  NlCam cam;
  nl_cam(intrinsics4_host, pose_c2w_host, &cam);
  GSDF_HIP(hipMemsetAsync(loss, 0, sizeof(float), stream), "normal_consistency memset");
  const int n_tiles = ((width + NL_T - 1) / NL_T) * ((height + NL_T - 1) / NL_T);
  normal_loss_kernel<false><<<n_tiles < 1024 ? n_tiles : 1024, 256, 0, stream>>>(height, width, cam, depth, alpha, render_normal, loss, nullptr, nullptr, nullptr, deterministic());
  GSDF_CHECK_LAUNCH("normal_loss_kernel<fwd>");
  return GSDF_OK;
}

// value and gradients in one launch; `loss` ACCUMULATES (the caller zeroes it, e.g. with the buffer it lives in)
extern "C" int gsdf_normal_consistency_fwd_bwd(int height, int width, const float *intrinsics4_host, const float *pose_c2w_host,
                                               const float *depth, const float *alpha, const float *render_normal, const float *v_loss,
                                               float *loss, float *v_depth, float *v_render_normal, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_normal_consistency_fwd_bwd");
  GSDF_REQUIRE(height > 2 && width > 2, "normal_consistency_fwd_bwd: image must be at least 3x3");
  GSDF_REQUIRE(intrinsics4_host && pose_c2w_host && depth && alpha && render_normal && v_loss && loss && v_depth && v_render_normal,
               "normal_consistency_fwd_bwd: null buffer");
  NlCam cam;
  nl_cam(intrinsics4_host, pose_c2w_host, &cam);
  const int n_tiles = ((width + NL_T - 1) / NL_T) * ((height + NL_T - 1) / NL_T);
  // capped grid: the value is one address (one atomic per workgroup)
  normal_loss_kernel<true><<<n_tiles < 2048 ? n_tiles : 2048, 256, 0, stream>>>(height, width, cam, depth, alpha, render_normal, loss, v_loss, v_depth,
                                                                               v_render_normal, deterministic());
  GSDF_CHECK_LAUNCH("normal_loss_kernel<fwd+bwd>");
  return GSDF_OK;
}

extern "C" int gsdf_normal_consistency_bwd(int height, int width, const float *intrinsics4_host, const float *pose_c2w_host,
                                           const float *depth, const float *alpha, const float *render_normal,
                                           const float *v_loss, float *v_depth, float *v_render_normal, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_normal_consistency_bwd");
  GSDF_REQUIRE(height > 2 && width > 2, "normal_consistency_bwd: image must be at least 3x3");
  GSDF_REQUIRE(intrinsics4_host && pose_c2w_host && depth && alpha && render_normal && v_loss && v_depth && v_render_normal,
               "normal_consistency_bwd: null buffer");
  NlCam cam;
  nl_cam(intrinsics4_host, pose_c2w_host, &cam);
  const int n_tiles = ((width + NL_T - 1) / NL_T) * ((height + NL_T - 1) / NL_T);
  normal_loss_kernel<true><<<n_tiles, 256, 0, stream>>>(height, width, cam, depth, alpha, render_normal, nullptr, v_loss, v_depth,
                                                      v_render_normal, false);
  GSDF_CHECK_LAUNCH("normal_loss_kernel<bwd>");
  return GSDF_OK;
}
