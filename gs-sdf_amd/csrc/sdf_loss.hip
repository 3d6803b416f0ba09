// sdf_loss.hip — S3: the per-ray SDF batch's two elementwise ends, fused.
//   query points: world points (+ the 6 central-difference stencil points of LocalMap::get_gradient,
//                 /root/reference/include/neural_net/local_map.cpp:110-131) -> unit-cube encoder inputs
//                 (SubMap::xyz_to_zp1_pts, /root/reference/include/neural_net/sub_map.cpp:82-97)
//   ray loss:     loss::sdf_loss (BCE with logits against the sigmoid-squashed target, isigma clamped at 5e2) on the
//                 base points + w_eik * loss::eikonal_loss of the numerical gradient
//                 (/root/reference/include/optimizer/loss.cpp:49-83; isigma = 1 + softplus(raw, beta 100) * bce_isigma,
//                 local_map.cpp:87-103), value AND gradient w.r.t. the decoder output in one pass.
// In the reference these are ~50 eager libtorch kernels forward and as many backward; they are pure launch latency.
#include "common.h"

namespace gsdf {

__global__ void __launch_bounds__(256)
    sdf_query_points_kernel(int64_t n, int K, const float *__restrict__ xyz, float delta, float px, float py, float pz,
                            float inv, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * K) return;
  const int64_t k = i / n, j = i - k * n;  // row 0..n-1: base points; then stencil-major (+x,-x,+y,-y,+z,-z)
  float x[3] = {xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
  if (k > 0) {
    const int axis = (int)(k - 1) >> 1;
    x[axis] = x[axis] + (((k - 1) & 1) ? -delta : delta);
  }
  const float p[3] = {px, py, pz};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float m = ((x[c] - p[c]) * 2.0f) * inv;  // scale_to_m1p1
    out[3 * i + c] = 0.5f * m + 0.5f;
  }
}

// the same for a batch that is the concatenation of n_a rows of `a` and n_b rows of `b` (rows ids[j] of it when ids != NULL): the
// gather and the concatenation of the joint iteration's SDF batch (ray points + visible splats' samples) happen in the load
__global__ void __launch_bounds__(256)
    sdf_query_points2_kernel(int64_t n_a, const float *__restrict__ a, int64_t n_b, const float *__restrict__ b,
                             const int64_t *__restrict__ ids, int K, float delta, float px, float py, float pz, float inv,
                             float *__restrict__ out) {
  const int64_t n = n_a + n_b;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * K) return;
  const int64_t k = i / n, j = i - k * n;
  const float *src = j < n_a ? a + 3 * j : b + 3 * (ids != nullptr ? ids[j - n_a] : j - n_a);
  float x[3] = {src[0], src[1], src[2]};
  if (k > 0) {
    const int axis = (int)(k - 1) >> 1;
    x[axis] = x[axis] + (((k - 1) & 1) ? -delta : delta);
  }
  const float p[3] = {px, py, pz};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float m = ((x[c] - p[c]) * 2.0f) * inv;  // scale_to_m1p1
    out[3 * i + c] = 0.5f * m + 0.5f;
  }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// 256-thread block sum -> ONE atomic per block on the loss value (atomics on one address serialise at ~88 per microsecond:
// one per wave cost 0.1 ms at 4.5e5 points)
static __device__ DetScalarSlot g_det_sdf_loss[3];   // deterministic mode: the ordered finish of the loss value, one slot per kernel
__device__ __forceinline__ void block_sum_to_loss(float contrib, float *__restrict__ loss, DetScalarSlot *det) {
  __shared__ float s_part[4];
  const float ws = wave_sum_to_lane63(contrib);
  if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = ws;
  __syncthreads();
  finish_scalars((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]), 0.f, loss, nullptr, det);
}

__global__ void __launch_bounds__(256)
    sdf_ray_loss_kernel(int64_t n, int stencil, const float *__restrict__ attr, int ld, const float *__restrict__ gt,
                        float bce_isigma, float delta, float w_eik, float *__restrict__ loss,
                        float *__restrict__ v_attr, bool det) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (i < n) {
    const float inv_n = 1.0f / (float)n;
    const float s = attr[i * ld], raw = attr[i * ld + 1], g = gt[i];
    // isigma = min(1 + softplus(raw, beta = 100, threshold = 20) * bce_isigma, 500)
    const float br = 100.0f * raw;
    const float sp = br > 20.0f ? raw : log1pf(expf(br)) * 0.01f;
    const float dsp = br > 20.0f ? 1.0f : sigmoidf(br);
    const float is0 = 1.0f + sp * bce_isigma;
    const float is = fminf(is0, 500.0f);
    const float dis_draw = is0 <= 500.0f ? dsp * bce_isigma : 0.0f;
    const float x = -s * is, u = -g * is;
    const float t0 = sigmoidf(u);
    const float t = fminf(fmaxf(t0, 1e-7f), 1.0f - 1e-7f);
    const float dt_du = (t0 >= 1e-7f && t0 <= 1.0f - 1e-7f) ? t0 * (1.0f - t0) : 0.0f;
    // bce = (1 - t) x + softplus(-x)
    const float bce = (1.0f - t) * x + fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
    const float dx = sigmoidf(x) - t, dt = -x;
    const float d_is = dx * (-s) + dt * dt_du * (-g);
    contrib = bce * inv_n;
    v_attr[i * ld] = dx * (-is) * inv_n;
    v_attr[i * ld + 1] = d_is * dis_draw * inv_n;
    for (int c = 2; c < ld; ++c) v_attr[i * ld + c] = 0.f;
    if (stencil) {
      float ps[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ps[k] = attr[(n + k * n + i) * ld];
      const float h = 0.5f * (1.0f / delta);
      const float gx = h * (ps[0] - ps[1]), gy = h * (ps[2] - ps[3]), gz = h * (ps[4] - ps[5]);
      const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
      const float e = nrm - 1.0f;
      contrib += w_eik * e * e * inv_n;
      const float k0 = nrm > 0.f ? w_eik * 2.0f * e / nrm * h * inv_n : 0.f;  // torch's norm backward: 0 at the origin
      const float d[3] = {k0 * gx, k0 * gy, k0 * gz};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int64_t r = (n + k * n + i) * ld;
        v_attr[r] = (k & 1) ? -d[k >> 1] : d[k >> 1];
        for (int c = 1; c < ld; ++c) v_attr[r + c] = 0.f;
      }
    }
  }
  block_sum_to_loss(contrib, loss, det ? &g_det_sdf_loss[0] : nullptr);
}

// loss::gs_sdf_loss (/root/reference/include/optimizer/loss.cpp:7-11): 0.5 * sum_i w_i * sdf_i^2, with the row gather of
// the weights (neural_mapping.cpp:436-437) folded in: w_i = weights[ids[i]] (ids == nullptr: w_i = weights[i]).
// stencil != 0: attr holds 7n rows (base points, then the 6 central-difference blocks of gsdf_sdf_query_points) and the
// eikonal regulariser of the visible splats' samples is added: w_eik * mean_i (|g_i| - 1)^2 with the numerical gradient
// g (NeuralSLAM::sdf_regularization on gs_samples, neural_mapping.cpp:448-451 -> :106-116; LocalMap::get_gradient
// local_map.cpp:110-131; loss::eikonal_loss loss.cpp:81-83).
__global__ void __launch_bounds__(256)
    gs_sdf_loss_kernel(int64_t n, int stencil, const float *__restrict__ attr, int ld, const float *__restrict__ weights,
                       const int64_t *__restrict__ ids, float scale, float delta, float w_eik, float *__restrict__ loss,
                       float *__restrict__ v_attr, bool det) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float contrib = 0.f;
  if (i < n) {
    const float s = attr[i * ld];
    const float w = weights[ids != nullptr ? ids[i] : i];
    contrib = 0.5f * scale * w * s * s;
    v_attr[i * ld] = scale * w * s;
    for (int c = 1; c < ld; ++c) v_attr[i * ld + c] = 0.f;
    if (stencil) {
      const float inv_n = 1.0f / (float)n;
      float ps[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ps[k] = attr[(n + k * n + i) * ld];
      const float h = 0.5f * (1.0f / delta);
      const float gx = h * (ps[0] - ps[1]), gy = h * (ps[2] - ps[3]), gz = h * (ps[4] - ps[5]);
      const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
      const float e = nrm - 1.0f;
      contrib += w_eik * e * e * inv_n;
      const float k0 = nrm > 0.f ? w_eik * 2.0f * e / nrm * h * inv_n : 0.f;  // torch's norm backward: 0 at the origin
      const float d[3] = {k0 * gx, k0 * gy, k0 * gz};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int64_t r = (n + k * n + i) * ld;
        v_attr[r] = (k & 1) ? -d[k >> 1] : d[k >> 1];
        for (int c = 1; c < ld; ++c) v_attr[r + c] = 0.f;
      }
    }
  }
  block_sum_to_loss(contrib, loss, det ? &g_det_sdf_loss[1] : nullptr);
}


// The data term of one base row and its gradient with respect to the decoder's outputs: rows [0, n_ray) w_sdf * loss::sdf_loss (mean over the
// per-ray batch), rows [n_ray, n) w_gs * loss::gs_sdf_loss.  Shared by sdf_analytic_loss_kernel and sdf_data_term_grad_kernel: the same operations
// in the same order, i.e. the same bits from either.
__device__ __forceinline__ void data_term(bool ray, float s, float raw, float g, float w, float bce_isigma, float w_sdf, float w_gs, float inv_n,
                                          float &c_pt, float &va0, float &va1) {
#pragma clang fp contract(off)   // (which products fuse with which sums is the compiler's choice per call site: none, so that both kernels round alike)
  if (ray) {
    const float br = 100.0f * raw;
    const float sp = br > 20.0f ? raw : log1pf(expf(br)) * 0.01f;
    const float dsp = br > 20.0f ? 1.0f : sigmoidf(br);
    const float is0 = 1.0f + sp * bce_isigma;
    const float is = fminf(is0, 500.0f);
    const float dis_draw = is0 <= 500.0f ? dsp * bce_isigma : 0.0f;
    const float x = -s * is, u = -g * is;
    const float t0 = sigmoidf(u);
    const float t = fminf(fmaxf(t0, 1e-7f), 1.0f - 1e-7f);
    const float dt_du = (t0 >= 1e-7f && t0 <= 1.0f - 1e-7f) ? t0 * (1.0f - t0) : 0.0f;
    const float bce = (1.0f - t) * x + fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
    const float dx = sigmoidf(x) - t, dt = -x;
    const float d_is = dx * (-s) + dt * dt_du * (-g);
    c_pt += w_sdf * bce * inv_n;
    va0 = w_sdf * dx * (-is) * inv_n;
    va1 = w_sdf * d_is * dis_draw * inv_n;
  } else {
    c_pt += 0.5f * w_gs * w * s * s;
    va0 = w_gs * w * s;
  }
}

// d (data terms) / d attr of the base rows ALONE (what sdf_analytic_loss_kernel writes as v_attr): it needs neither the stencil rows nor g0, so
// the joint iteration can send the samples' gradient on its way (decoder backward -> Jacobian contraction) before the regularisers' inputs exist.
__global__ void __launch_bounds__(256)
    sdf_data_term_grad_kernel(int64_t n, int64_t n_ray, const float *__restrict__ attr, int ld, const float *__restrict__ gt,
                              const float *__restrict__ weights, const int64_t *__restrict__ ids, float bce_isigma, float w_sdf, float w_gs,
                              float *__restrict__ v_attr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool ray = i < n_ray;
  const float inv_n = 1.0f / (float)(ray ? n_ray : n - n_ray);
  float c_pt = 0.f, va0 = 0.f, va1 = 0.f;
  data_term(ray, attr[i * ld], ray ? attr[i * ld + 1] : 0.f, ray ? gt[i] : 0.f, ray ? 0.f : weights[ids != nullptr ? ids[i - n_ray] : i - n_ray], bce_isigma,
            w_sdf, w_gs, inv_n, c_pt, va0, va1);
  v_attr[i * ld] = va0;
  if (ld > 1) v_attr[i * ld + 1] = va1;
  for (int c = 2; c < ld; ++c) v_attr[i * ld + c] = 0.f;
}

// ---- the reference's DEFAULT regulariser: analytic SDF gradient (numerical_grad: 0, config/base.yaml:13) ------------------
// NeuralSLAM::sdf_regularization (/root/reference/include/neural_mapping/neural_mapping.cpp:106-136) with
// LocalMap::get_gradient's autograd branch (include/neural_net/local_map.cpp:151-172):
//     g      = d sdf / d xyz            = map_size_inv * J(x)^T g0     (g0 = d sdf / d features, J = d features / d x)
//     loss  += w_eik * mean_n (|g| - 1)^2                                                   (loss::eikonal_loss, loss.cpp:81-83)
//     loss  += w_align * mean_{n,3} |g - g_num.detach()|,  g_num = central differences       (:126-134, align_weight 0.1)
// on top of the batch's data term: mode 0 = sdf_weight * loss::sdf_loss (per-ray batch, neural_mapping.cpp:165-170), mode 1 =
// scale * loss::gs_sdf_loss at the visible splats' samples (:436-457).  One thread per point: value, d/d decoder output,
// vv_x = dL/d(J^T g0) (unit-cube coordinates) and u0 = J vv_x = dL/d g0 — what the double backward of the decoder
// (gsdf_mlp_bwd_bwd) and of the encoder (gsdf_hashgrid_bwd_binned2) consume.
// Round 4: EIGHT lanes per point (NF = 32: four features per lane).  One lane per point read its own 384-byte Jacobian row and 128-byte
// g0 row at a 384-byte lane stride — 64 lines per load instruction, the rows thrashing the L1 — and read the Jacobian twice (J^T g0, then
// J vv_x).  Now a group's 8 lanes read their point's rows as 8 x 48 and 8 x 16 contiguous bytes (a wave = 8 consecutive points = 3 KB /
// 1 KB contiguous), keep the 12 Jacobian entries of their 4 features in registers for both products, reduce the three components of
// J^T g0 over the group with DPP and write u0 as one float4 each.  The per-point scalar work (data term, eikonal, align) is done by every
// lane of the group on the same operands; lane 0 writes it.  0.213 -> 0.089 ms at 380 k points (the kernel now moves its algorithmic 0.75 KB per point).
template <int NF>
__global__ void __launch_bounds__(256)
    sdf_analytic_loss_kernel(int64_t n, int64_t n_ray, int stencil, const float *__restrict__ attr, int ld, const float *__restrict__ g0,
                             const float *__restrict__ jac, const float *__restrict__ gt, const float *__restrict__ weights,
                             const int64_t *__restrict__ ids, float bce_isigma, float w_sdf, float w_gs, float map_size_inv, float delta,
                             float w_eik, float w_align, float *__restrict__ loss, float *__restrict__ v_attr,
                             float *__restrict__ vv_x, float *__restrict__ u0, bool det) {
  static_assert(NF == 32, "8 lanes x 4 features");
  __shared__ float s_part[4];
  float contrib = 0.f;
  const int q = threadIdx.x & 7;                       // lane within the point's group: features 4q .. 4q+3
  const int64_t gpb = 256 / 8;                         // points per workgroup and round
  const int64_t rounds = (n + gpb * gridDim.x - 1) / (gpb * gridDim.x);
  // capped grid + grid-stride loop: the value is ONE address, and atomics on one line serialise at ~88 per microsecond
  for (int64_t rd = 0; rd < rounds; ++rd) {
    const int64_t i = (rd * gridDim.x + blockIdx.x) * gpb + (threadIdx.x >> 3);
    const bool live = i < n;                           // whole groups are live or dead together; dead groups still take part in the DPP
    const int64_t ic = live ? i : 0;
    // rows [0, n_ray): per-ray batch (mean over n_ray); rows [n_ray, n): splat samples (mean over n - n_ray): the two
    // sdf_regularization calls of the iteration normalise separately (neural_mapping.cpp:183-186, :448-451)
    const bool ray = ic < n_ray;
    const float inv_n = 1.0f / (float)(ray ? n_ray : n - n_ray);
    const float s = attr[ic * ld];
    float c_pt = 0.f, va0 = 0.f, va1 = 0.f;
    data_term(ray, s, ray ? attr[ic * ld + 1] : 0.f, ray ? gt[ic] : 0.f, ray ? 0.f : weights[ids != nullptr ? ids[ic - n_ray] : ic - n_ray], bce_isigma,
              w_sdf, w_gs, inv_n, c_pt, va0, va1);
    // analytic gradient in world units: this lane's 4 features
    const float4 *J4 = reinterpret_cast<const float4 *>(jac + ic * (int64_t)(NF * 3) + 12 * q);
    const float4 j0 = J4[0], j1 = J4[1], j2 = J4[2];
    const float4 gq = *reinterpret_cast<const float4 *>(g0 + ic * (int64_t)NF + 4 * q);
    const float J[12] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w, j2.x, j2.y, j2.z, j2.w};
    const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) { ax = fmaf(J[3 * f], gv[f], ax); ay = fmaf(J[3 * f + 1], gv[f], ay); az = fmaf(J[3 * f + 2], gv[f], az); }
    // sum over the 8 lanes of the group (all lanes get it): lane ^ 1, ^ 2 (quad permutes), ^ 4 (row_shl / row_shr by 4 within 8-lane halves)
    ax += dpp_mov<0xB1>(ax); ay += dpp_mov<0xB1>(ay); az += dpp_mov<0xB1>(az);
    ax += dpp_mov<0x4E>(ax); ay += dpp_mov<0x4E>(ay); az += dpp_mov<0x4E>(az);
    ax += dpp_xchg<0x104, 0x114, 0x5, 0xA>(ax); ay += dpp_xchg<0x104, 0x114, 0x5, 0xA>(ay); az += dpp_xchg<0x104, 0x114, 0x5, 0xA>(az);
    ax *= map_size_inv; ay *= map_size_inv; az *= map_size_inv;
    const float nrm = sqrtf(ax * ax + ay * ay + az * az);
    const float e = nrm - 1.0f;
    c_pt += w_eik * e * e * inv_n;
    const float k0 = nrm > 0.f ? w_eik * 2.0f * e / nrm * inv_n : 0.f;    // torch's norm backward: 0 at the origin
    float vx = k0 * ax, vy = k0 * ay, vz = k0 * az;
    if (stencil && w_align != 0.f) {
      float ps[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ps[k] = attr[(n + k * n + ic) * ld];
      const float h = 0.5f * (1.0f / delta);
      const float nx = h * (ps[0] - ps[1]), ny = h * (ps[2] - ps[3]), nz = h * (ps[4] - ps[5]);
      const float c3 = w_align * inv_n * (1.0f / 3.0f);
      const float dx_ = ax - nx, dy_ = ay - ny, dz_ = az - nz;
      c_pt += c3 * (fabsf(dx_) + fabsf(dy_) + fabsf(dz_));
      // torch's abs backward: sign(0) = 0
      vx += c3 * (dx_ > 0.f ? 1.f : (dx_ < 0.f ? -1.f : 0.f));
      vy += c3 * (dy_ > 0.f ? 1.f : (dy_ < 0.f ? -1.f : 0.f));
      vz += c3 * (dz_ > 0.f ? 1.f : (dz_ < 0.f ? -1.f : 0.f));
    }
    vx *= map_size_inv; vy *= map_size_inv; vz *= map_size_inv;       // dL / d (J^T g0)
    if (live) {
      float4 uo;
      uo.x = fmaf(J[2], vz, fmaf(J[1], vy, J[0] * vx));
      uo.y = fmaf(J[5], vz, fmaf(J[4], vy, J[3] * vx));
      uo.z = fmaf(J[8], vz, fmaf(J[7], vy, J[6] * vx));
      uo.w = fmaf(J[11], vz, fmaf(J[10], vy, J[9] * vx));
      *reinterpret_cast<float4 *>(u0 + i * (int64_t)NF + 4 * q) = uo;
      if (q == 0) {
        contrib += c_pt;
        if (v_attr != nullptr) {                      // (NULL: gsdf_sdf_data_term_grad has written it already)
          v_attr[i * ld] = va0;
          if (ld > 1) v_attr[i * ld + 1] = va1;       // (0 for a splat-sample row)
          for (int c = 2; c < ld; ++c) v_attr[i * ld + c] = 0.f;
        }
        vv_x[3 * i] = vx; vv_x[3 * i + 1] = vy; vv_x[3 * i + 2] = vz;
      }
    }
  }
  const float ws = wave_sum_to_lane63(contrib);
  if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = ws;
  __syncthreads();
  finish_scalars((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]), 0.f, loss, nullptr, det ? &g_det_sdf_loss[2] : nullptr);
}

}  // namespace gsdf

using namespace gsdf;

extern "C" int gsdf_sdf_query_points(int64_t n, int stencil, const float *xyz, float delta, const float *origin_host,
                                     float map_size_inv, float *out, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_sdf_query_points");
  GSDF_REQUIRE(n >= 0 && origin_host, "sdf_query_points: bad arguments");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(xyz && out, "sdf_query_points: null buffer");
  const int K = stencil ? 7 : 1;
  sdf_query_points_kernel<<<(unsigned)((n * K + 255) / 256), 256, 0, stream>>>(n, K, xyz, delta, origin_host[0],
                                                                                origin_host[1], origin_host[2],
                                                                                map_size_inv, out);
  GSDF_CHECK_LAUNCH("sdf_query_points_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_sdf_query_points2(int64_t n_a, const float *xyz_a, int64_t n_b, const float *xyz_b, const int64_t *ids_b, int stencil,
                                      float delta, const float *origin_host, float map_size_inv, float *out, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_sdf_query_points");
  GSDF_REQUIRE(n_a >= 0 && n_b >= 0 && origin_host, "sdf_query_points2: bad arguments");
  const int64_t n = n_a + n_b;
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE((n_a == 0 || xyz_a) && (n_b == 0 || xyz_b) && out, "sdf_query_points2: null buffer");
  const int K = stencil ? 7 : 1;
  sdf_query_points2_kernel<<<(unsigned)((n * K + 255) / 256), 256, 0, stream>>>(n_a, xyz_a, n_b, xyz_b, ids_b, K, delta, origin_host[0], origin_host[1],
                                                                                 origin_host[2], map_size_inv, out);
  GSDF_CHECK_LAUNCH("sdf_query_points2_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_sdf_ray_loss(int64_t n, int stencil, const float *attr, int ld, const float *gt_sdf, float bce_isigma,
                                 float delta, float w_eik, float *loss, float *v_attr, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_sdf_ray_loss");
  GSDF_REQUIRE(n > 0 && ld >= 2 && attr && gt_sdf && loss && v_attr, "sdf_ray_loss: bad arguments");
  GSDF_REQUIRE(!stencil || delta > 0.f, "sdf_ray_loss: delta must be positive");
  GSDF_HIP(hipMemsetAsync(loss, 0, 4, stream), "sdf_ray_loss memset");
  const bool det = deterministic();
  GSDF_REQUIRE(!det || (n + 255) / 256 <= DET_MAX_BLOCKS, "sdf_ray_loss: deterministic mode takes at most %d rows", DET_MAX_BLOCKS * 256);
  sdf_ray_loss_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, stencil, attr, ld, gt_sdf, bce_isigma, delta,
                                                                       w_eik, loss, v_attr, det);
  GSDF_CHECK_LAUNCH("sdf_ray_loss_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_gs_sdf_eik_loss(int64_t n, int stencil, const float *attr, int ld, const float *weights,
                                    const int64_t *ids, float scale, float delta, float w_eik, float *loss,
                                    float *v_attr, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_gs_sdf_eik_loss");
  GSDF_REQUIRE(n >= 0 && ld >= 1 && loss, "gs_sdf_loss: bad arguments");
  GSDF_REQUIRE(!stencil || delta > 0.f, "gs_sdf_loss: delta must be positive");
  GSDF_HIP(hipMemsetAsync(loss, 0, 4, stream), "gs_sdf_loss memset");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(attr && weights && v_attr, "gs_sdf_loss: null buffer");
  const bool det = deterministic();
  GSDF_REQUIRE(!det || (n + 255) / 256 <= DET_MAX_BLOCKS, "gs_sdf_loss: deterministic mode takes at most %d rows", DET_MAX_BLOCKS * 256);
  gs_sdf_loss_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, stencil, attr, ld, weights, ids, scale, delta,
                                                                      w_eik, loss, v_attr, det);
  GSDF_CHECK_LAUNCH("gs_sdf_loss_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_gs_sdf_loss(int64_t n, const float *attr, int ld, const float *weights, const int64_t *ids, float scale,
                                float *loss, float *v_attr, gsdf_stream_t stream) {
  return gsdf_gs_sdf_eik_loss(n, 0, attr, ld, weights, ids, scale, 0.f, 0.f, loss, v_attr, stream);
}

extern "C" int gsdf_sdf_analytic_loss(int64_t n, int64_t n_ray, int stencil, const float *attr, int ld, const float *g0, int n_feat,
                                      const float *jac, const float *gt_sdf, const float *weights, const int64_t *ids,
                                      float bce_isigma, float w_sdf, float w_gs, float map_size_inv, float delta, float w_eik,
                                      float w_align, float *loss, float *v_attr, float *vv_x, float *u0, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_sdf_analytic_loss");
  GSDF_REQUIRE(n >= 0 && n_ray >= 0 && n_ray <= n && loss, "sdf_analytic_loss: bad arguments");
  GSDF_REQUIRE(n_feat == 32, "sdf_analytic_loss: %d encoder features unsupported (32: 16 levels x 2, as the reference configures)", n_feat);
  GSDF_REQUIRE(ld >= (n_ray > 0 ? 2 : 1), "sdf_analytic_loss: decoder output too narrow");
  GSDF_REQUIRE(!(stencil && w_align != 0.f) || delta > 0.f, "sdf_analytic_loss: delta must be positive");
  GSDF_HIP(hipMemsetAsync(loss, 0, 4, stream), "sdf_analytic_loss memset");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(attr && g0 && jac && vv_x && u0 && (n_ray == 0 || gt_sdf) && (n_ray == n || weights), "sdf_analytic_loss: null buffer");
  const int64_t blocks = (n + 31) / 32;          // 8 lanes per point
  sdf_analytic_loss_kernel<32><<<(unsigned)(blocks > 2048 ? 2048 : blocks), 256, 0, stream>>>(n, n_ray, stencil, attr, ld, g0, jac, gt_sdf, weights,
                                                                                           ids, bce_isigma, w_sdf, w_gs, map_size_inv, delta,
                                                                                           w_eik, w_align, loss, v_attr, vv_x, u0, deterministic());
  GSDF_CHECK_LAUNCH("sdf_analytic_loss_kernel");
  return GSDF_OK;
}

extern "C" int gsdf_sdf_data_term_grad(int64_t n, int64_t n_ray, const float *attr, int ld, const float *gt_sdf, const float *weights, const int64_t *ids,
                                       float bce_isigma, float w_sdf, float w_gs, float *v_attr, gsdf_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GSDF_TIMED("gsdf_sdf_data_term_grad");
  GSDF_REQUIRE(n >= 0 && n_ray >= 0 && n_ray <= n, "sdf_data_term_grad: bad arguments");
  GSDF_REQUIRE(ld >= (n_ray > 0 ? 2 : 1), "sdf_data_term_grad: decoder output too narrow");
  if (n == 0) return GSDF_OK;
  GSDF_REQUIRE(attr && v_attr && (n_ray == 0 || gt_sdf) && (n_ray == n || weights), "sdf_data_term_grad: null buffer");
  sdf_data_term_grad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(n, n_ray, attr, ld, gt_sdf, weights, ids, bce_isigma, w_sdf, w_gs, v_attr);
  GSDF_CHECK_LAUNCH("sdf_data_term_grad_kernel");
  return GSDF_OK;
}
