// extras.cpp — gsdf_extras/gsdf_extras.h over the C ABI (include/gsdf_hip.h sections O2, S3, a2, a8, O1).
#include "gsdf_extras/gsdf_extras.h"
#include "tcnn_binding/tcnn_binding.h"

#include <cmath>

#include "stream_gate.h"
#include "util.h"

using namespace gsdf_host;
using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

namespace {
const float *ssim_window() {  // the reference's 1-D window (loss_utils.cpp:6-14), NOT the symmetric Gaussian
  static float w[11];
  static bool init = false;
  if (!init) {
    double s = 0, g[11];
    for (int x = 0; x < 11; ++x) { g[x] = std::exp(-std::pow(std::floor((x - 11) / 2.0), 2) / (2.0 * 1.5 * 1.5)); s += g[x]; }
    for (int x = 0; x < 11; ++x) w[x] = (float)(g[x] / s);
    init = true;
  }
  return w;
}

struct L1Dssim : public torch::autograd::Function<L1Dssim> {
  static Tensor forward(AutogradContext *ctx, const Tensor &img_, const Tensor &gt_, double w1, double w2) {
    Tensor img = f32c(img_, "render"), gt = f32c(gt_, "gt");
    TORCH_CHECK(img.dim() == 3 && img.size(2) == 3 && img.sizes() == gt.sizes(), "l1_dssim_loss: expected two [H,W,3] images");
    const int H = (int)img.size(0), W = (int)img.size(1);
    Tensor sums = empty_like_opts(img, {2}, torch::kFloat32);
    Tensor maps = img_.requires_grad() ? empty_like_opts(img, {3, H, W, 3}, torch::kFloat32) : Tensor();
    check(gsdf_l1_dssim_fwd(H, W, fp(img), fp(gt), ssim_window(), fpm(sums), fpm(maps), cur_stream()), "l1_dssim_fwd");
    ctx->save_for_backward({img, gt, maps});
    ctx->saved_data["w1"] = w1;
    ctx->saved_data["w2"] = w2;
    const double n = 3.0 * H * W;
    return w1 * sums[0] / n + w2 * (1.0 - sums[1] / n);
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const int H = (int)s[0].size(0), W = (int)s[0].size(1);
    Tensor v_img = torch::empty_like(s[0]);
    Tensor v = f32c(g[0].reshape({1}), "grad");
    check(gsdf_l1_dssim_bwd(H, W, fp(s[0]), fp(s[1]), ssim_window(), fp(s[2]), fp(v), (float)ctx->saved_data["w1"].toDouble(),
                            (float)ctx->saved_data["w2"].toDouble(), fpm(v_img), cur_stream()),
          "l1_dssim_bwd");
    return {v_img, Tensor(), Tensor(), Tensor()};
  }
};

struct ValueAndGrad : public torch::autograd::Function<ValueAndGrad> {  // loss whose gradient was produced by the forward launch
  static Tensor forward(AutogradContext *ctx, const Tensor &attr, const Tensor &loss, const Tensor &v_attr) {
    ctx->save_for_backward({v_attr});
    return loss.clone();
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    return {ctx->get_saved_variables()[0] * g[0], Tensor(), Tensor()};
  }
};

struct Activate : public torch::autograd::Function<Activate> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &anchors_, const Tensor &offsets_, const Tensor &scaling_,
                             const Tensor &opacity_) {
    Tensor anchors = f32c(anchors_, "anchors"), offsets = f32c(offsets_, "offsets"), scaling = f32c(scaling_, "scaling");
    Tensor opacity = f32c(opacity_, "opacity");
    const int64_t n = anchors.size(0);
    Tensor xyz = torch::empty_like(anchors), scales = torch::empty_like(scaling), opac = empty_like_opts(anchors, {n}, torch::kFloat32);
    check(gsdf_splat_activations_fwd(n, fp(anchors), fp(offsets), fp(scaling), fp(opacity), fpm(xyz), fpm(scales), fpm(opac), cur_stream()),
          "splat_activations_fwd");
    ctx->save_for_backward({scales, opac});
    return {xyz, scales, opac};
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const int64_t n = s[1].size(0);
    auto z = [&](const Tensor &t, const Tensor &like) { return t.defined() ? f32c(t, "grad") : torch::zeros_like(like); };
    Tensor vx = g[0].defined() ? f32c(g[0], "grad") : torch::zeros({n, 3}, s[0].options());
    Tensor vs = z(g[1], s[0]), vo = z(g[2], s[1]);
    Tensor g_off = torch::zeros_like(vx), g_sc = torch::zeros_like(s[0]), g_op = torch::zeros_like(s[1]);   // the kernel accumulates
    check(gsdf_splat_activations_bwd(n, fp(s[0]), fp(s[1]), fp(vx), fp(vs), fp(vo), fpm(g_off), fpm(g_sc), fpm(g_op), cur_stream()),
          "splat_activations_bwd");
    return {Tensor(), g_off, g_sc, g_op};
  }
};
}  // namespace

namespace gsdf_extras {

Tensor l1_dssim_loss(const Tensor &render, const Tensor &gt, double rgb_weight, double dssim_weight) {
  return L1Dssim::apply(render, gt, rgb_weight, dssim_weight);
}

Tensor query_points(const Tensor &xyz_, const std::vector<float> &origin, double map_size_inv, bool with_stencil, double delta) {
  TORCH_CHECK(origin.size() == 3, "query_points: map_origin needs 3 entries");
  Tensor xyz = f32c(xyz_.detach(), "xyz");
  const int64_t n = xyz.size(0);
  Tensor out = empty_like_opts(xyz, {(with_stencil ? 7 : 1) * n, 3}, torch::kFloat32);
  check(gsdf_sdf_query_points(n, with_stencil ? 1 : 0, fp(xyz), (float)delta, origin.data(), (float)map_size_inv, fpm(out), cur_stream()),
        "sdf_query_points");
  return out;
}

Tensor sdf_ray_loss(const Tensor &attr_, const Tensor &gt_sdf, int64_t n, double bce_isigma, double delta, double w_eik) {
  Tensor attr = f32c(attr_, "attr"), gt = f32c(gt_sdf, "gt_sdf");
  const bool stencil = attr.size(0) == 7 * n;
  TORCH_CHECK(stencil || attr.size(0) == n, "sdf_ray_loss: attr must have n or 7n rows");
  Tensor loss = empty_like_opts(attr, {}, torch::kFloat32), v_attr = torch::empty_like(attr);
  check(gsdf_sdf_ray_loss(n, stencil ? 1 : 0, fp(attr), (int)attr.size(1), fp(gt), (float)bce_isigma, (float)delta, (float)w_eik,
                          fpm(loss), fpm(v_attr), cur_stream()),
        "sdf_ray_loss");
  return ValueAndGrad::apply(attr_, loss, v_attr);
}

Tensor gs_sdf_eik_loss(const Tensor &attr_, const Tensor &weights, const Tensor &ids, int64_t n, double scale, double delta, double w_eik) {
  Tensor attr = f32c(attr_, "attr"), w = f32c(weights.reshape({-1}), "weights");
  const bool stencil = attr.size(0) == 7 * n && n > 0 && delta > 0;
  Tensor idc = ids.defined() ? ids.contiguous() : Tensor();
  Tensor loss = empty_like_opts(attr, {}, torch::kFloat32), v_attr = torch::empty_like(attr);
  check(gsdf_gs_sdf_eik_loss(n, stencil ? 1 : 0, fp(attr), (int)attr.size(1), fp(w), idc.defined() ? idc.data_ptr<int64_t>() : nullptr,
                             (float)scale, (float)delta, (float)w_eik, fpm(loss), fpm(v_attr), cur_stream()),
        "gs_sdf_eik_loss");
  return ValueAndGrad::apply(attr_, loss, v_attr);
}

namespace {
struct CouplingCfg { int L, F, H, R; float S; std::vector<int> dims; float origin[3]; double map_size_inv, scale, delta, w_eik; };
// autograd contexts carry IValues only: the configuration travels as an int list + a double list
CouplingCfg coupling_cfg(const std::vector<int64_t> &iv, const std::vector<double> &dv) {
  CouplingCfg c;
  c.L = (int)iv[0]; c.F = (int)iv[1]; c.H = (int)iv[2]; c.R = (int)iv[3];
  c.dims.assign(iv.begin() + 4, iv.end());
  c.S = (float)dv[0];
  for (int d = 0; d < 3; ++d) c.origin[d] = (float)dv[1 + d];
  c.map_size_inv = dv[4]; c.scale = dv[5]; c.delta = dv[6]; c.w_eik = dv[7];
  return c;
}

// neural_mapping.cpp:420-462 as one node (Python mirror: gs_sdf_amd/sdf.py _CouplingLeg)
struct CouplingFn : public torch::autograd::Function<CouplingFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &samples, const Tensor &ids_, const Tensor &weights, const Tensor &table_,
                        const Tensor &W_, Tensor table_grad, Tensor decoder_grad, std::vector<int64_t> iv, std::vector<double> dv,
                        int64_t gate_handle) {
    const CouplingCfg c = coupling_cfg(iv, dv);
    Tensor ids = ids_.contiguous(), table = f32c(table_.detach(), "encoder params"), W = f32c(W_.detach(), "decoder params");
    Tensor smp = f32c(samples.detach(), "samples"), xs = smp;   // (xs: device / dtype donor of the buffers below)
    const int64_t n = ids.size(0), K = c.delta > 0 ? 7 : 1, nq = K * n;
    const int nf = c.L * c.F, nl = (int)c.dims.size() - 1;
    Tensor x01 = empty_like_opts(xs, {nq, 3}, torch::kFloat32);
    Tensor feat = empty_like_opts(xs, {nq, nf}, torch::kFloat32), jac = empty_like_opts(xs, {n, nf, 3}, torch::kFloat32);
    if (K == 7 && n > 0) {   // rows ids[j] of the samples, their 6 stencil points and the encoder in one launch
      check(gsdf_hashgrid_fwd_stencil_points(0, nullptr, n, fp(smp), ids.data_ptr<int64_t>(), (float)c.delta, c.origin, (float)c.map_size_inv, 1, c.L, c.F, c.H, c.R,
                                             c.S, fp(table), fpm(x01), fpm(feat), fpm(jac), cur_stream()), "hashgrid_fwd_stencil_points");
    } else {
      Tensor rows = f32c(smp.index_select(0, ids), "samples");
      check(gsdf_sdf_query_points(n, K == 7, fp(rows), (float)c.delta, c.origin, (float)c.map_size_inv, fpm(x01), cur_stream()), "sdf_query_points");
      check(gsdf_hashgrid_fwd_jac_rows(nq, n, c.L, c.F, c.H, c.R, c.S, fp(x01), fp(table), fpm(feat), fpm(jac), cur_stream()), "hashgrid_fwd_jac");
    }
    Tensor attr = empty_like_opts(xs, {nq, (int64_t)c.dims.back()}, torch::kFloat32);
    Tensor acts = empty_like_opts(xs, {(int64_t)gsdf_mlp_acts_floats(nq, nl)}, torch::kFloat32);
    check(gsdf_mlp_fwd(nq, nl, c.dims.data(), fp(W), nullptr, fp(feat), fpm(attr), fpm(acts), cur_stream()), "mlp_fwd");
    Tensor loss = empty_like_opts(xs, {}, torch::kFloat32), v_attr = torch::empty_like(attr);
    Tensor w = f32c(weights.detach().reshape({-1}), "weights");
    check(gsdf_gs_sdf_eik_loss(n, K == 7, fp(attr), (int)attr.size(1), fp(w), ids.data_ptr<int64_t>(), (float)c.scale, (float)c.delta,
                               (float)c.w_eik, fpm(loss), fpm(v_attr), cur_stream()), "gs_sdf_eik_loss");
    // the two in-place gradient sinks travel OUTSIDE save_for_backward: they are views of the flat gradient buffer, which other
    // nodes of the same backward pass legitimately write before this one runs (autograd's saved-tensor version check would throw)
    ctx->save_for_backward({ids, x01, feat, jac, acts, v_attr, table, W});
    ctx->saved_data["table_grad"] = table_grad;
    ctx->saved_data["decoder_grad"] = decoder_grad;
    ctx->saved_data["n_rows"] = samples.size(0);
    ctx->saved_data["gate"] = gate_handle;
    ctx->saved_data["iv"] = iv;
    ctx->saved_data["dv"] = dv;
    return loss;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &ids = s[0], &x01 = s[1], &feat = s[2], &jac = s[3], &acts = s[4], &v_attr = s[5], &table = s[6], &W = s[7];
    Tensor table_grad = ctx->saved_data["table_grad"].toTensor(), decoder_grad = ctx->saved_data["decoder_grad"].toTensor();
    const CouplingCfg c = coupling_cfg(ctx->saved_data["iv"].toIntVector(), ctx->saved_data["dv"].toDoubleVector());
    const int64_t nq = x01.size(0), n = jac.size(0);
    const int nl = (int)c.dims.size() - 1;
    Tensor v_out = (v_attr * g[0]).contiguous(), v_feat = torch::empty_like(feat);
    Tensor ws = empty_like_opts(feat, {(int64_t)gsdf_mlp_bwd_ws_bytes_for(nq, nl, c.dims.data(), 1)}, torch::kUInt8);
    check(gsdf_mlp_bwd(nq, nl, c.dims.data(), fp(W), nullptr, fp(feat), fp(acts), fp(v_out), fpm(v_feat), fpm(decoder_grad), nullptr,
                       ws.data_ptr(), cur_stream()), "mlp_bwd");
    Tensor v_x = empty_like_opts(feat, {n, 3}, torch::kFloat32);
    check(gsdf_hashgrid_bwd_jac(n, c.L, c.F, fp(jac), fp(v_feat), fpm(v_x), cur_stream()), "hashgrid_bwd_jac");
    Tensor v_samples = torch::zeros({ctx->saved_data["n_rows"].toInt(), 3}, feat.options());
    v_samples.index_add_(0, ids, v_x * c.map_size_inv);          // d x01 / d xyz = 0.5 * 2 * map_size_inv
    if (auto *gate = reinterpret_cast<gsdf_extras::StreamGate *>(ctx->saved_data["gate"].toInt())) gate->record_here();
    // the binned scatter does not cover every grid (log2_hashmap_size >= 20, more than 4096 tiles: ws_bytes == 0): those take
    // the atomic kernel, like TCNNEncoding's own backward and the Python mirror
    const size_t nb = (nq >= 24576 || gsdf_deterministic(-1)) ? gsdf_hashgrid_bwd_binned_ws_bytes(nq, c.L, c.F, c.H, c.R, c.S) : 0;   // (deterministic mode: no atomic scatter)
    if (nb > 0) {
      int merge = 0;
      for (int l = 0; l < c.L; ++l)
        if ((c.R * std::pow((double)c.S, l) - 1.0) * c.delta * c.map_size_inv < 1.0) ++merge;
      Tensor bws = empty_like_opts(feat, {(int64_t)nb}, torch::kUInt8);
      check(gsdf_hashgrid_bwd_binned_stencil(nq, nq == 7 * n ? n : 0, merge, c.L, c.F, c.H, c.R, c.S, fp(x01), fp(v_feat), fpm(table_grad),
                                             bws.data_ptr(), nb, cur_stream()), "hashgrid_bwd_binned_stencil");
    } else if (nq > 0) {
      check(gsdf_hashgrid_bwd(nq, c.L, c.F, c.H, c.R, c.S, fp(x01), fp(table), fp(v_feat), fpm(table_grad), nullptr, cur_stream()), "hashgrid_bwd");
    }
    return {v_samples, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
}  // namespace

namespace {
// Python mirror: gs_sdf_amd/sdf.py _SdfBatchAnalytic (same kernels, same order)
struct AnalyticFn : public torch::autograd::Function<AnalyticFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &ray_xyz, const Tensor &gt_sdf, const Tensor &samples, const Tensor &ids_,
                        const Tensor &weights, const Tensor &table_, const Tensor &W_, const Tensor &bias_, Tensor table_grad, Tensor decoder_grad,
                        Tensor bias_grad, std::vector<int64_t> iv, std::vector<double> dv, int64_t gate_handle) {
    // iv = {L, F, H, R, dims...}; dv = {S, origin x3, map_size_inv, bce_isigma, w_sdf, w_gs, delta, w_eik, w_align, unit_upstream, first_order_in_forward}
    const int L = (int)iv[0], F = (int)iv[1], H = (int)iv[2], R = (int)iv[3];
    const std::vector<int> dims(iv.begin() + 4, iv.end());
    const float S = (float)dv[0];
    const float origin[3] = {(float)dv[1], (float)dv[2], (float)dv[3]};
    const double map_size_inv = dv[4], bce_isigma = dv[5], w_sdf = dv[6], w_gs = dv[7], delta = dv[8], w_eik = dv[9], w_align = dv[10];
    Tensor table = f32c(table_.detach(), "encoder params"), W = f32c(W_.detach(), "decoder params");
    Tensor bias = (bias_.defined() && bias_.numel() > 0) ? f32c(bias_.detach(), "decoder biases") : Tensor();
    // the batch = [ray points; samples[ids]]: gathered and concatenated inside the query-points launch
    Tensor ray, smp;
    int64_t n_ray = 0, n_smp = 0;
    if (ray_xyz.defined() && ray_xyz.numel() > 0) { ray = f32c(ray_xyz.detach().reshape({-1, 3}), "ray_xyz"); n_ray = ray.size(0); }
    Tensor ids = (ids_.defined() && ids_.numel() > 0) ? ids_.contiguous() : Tensor();
    if (samples.defined() && samples.numel() > 0) { smp = f32c(samples.detach().reshape({-1, 3}), "samples"); n_smp = ids.defined() ? ids.size(0) : smp.size(0); }
    TORCH_CHECK(n_ray + n_smp > 0, "joint_sdf_loss_analytic: no points");
    const Tensor &xs = ray.defined() ? ray : smp;   // (device / dtype donor of the buffers below)
    const int64_t n = n_ray + n_smp;
    const bool stencil = w_align != 0.0 && n > 0;
    const int64_t K = stencil ? 7 : 1, nq = K * n;
    const int nf = L * F, nl = (int)dims.size() - 1;
    Tensor x01 = empty_like_opts(xs, {nq, 3}, torch::kFloat32);
    Tensor feat = empty_like_opts(xs, {nq, nf}, torch::kFloat32), jac = empty_like_opts(xs, {n, nf, 3}, torch::kFloat32);
    if (stencil && delta > 0.0) {   // the query points are made inside the encoder's launch (the same rows, bit for bit, written to x01 by it)
      check(gsdf_hashgrid_fwd_stencil_points(n_ray, fp(ray), n_smp, fp(smp), ids.defined() ? ids.data_ptr<int64_t>() : nullptr, (float)delta, origin,
                                             (float)map_size_inv, 1, L, F, H, R, S, fp(table), fpm(x01), fpm(feat), fpm(jac), cur_stream()),
            "hashgrid_fwd_stencil_points");
    } else {
      check(gsdf_sdf_query_points2(n_ray, fp(ray), n_smp, fp(smp), ids.defined() ? ids.data_ptr<int64_t>() : nullptr, stencil ? 1 : 0, (float)delta, origin,
                                   (float)map_size_inv, fpm(x01), cur_stream()), "sdf_query_points");
      if (stencil)
        check(gsdf_hashgrid_fwd_stencil(nq, n, n, L, F, H, R, S, fp(x01), fp(table), fpm(feat), fpm(jac), cur_stream()), "hashgrid_fwd_stencil");
      else
        check(gsdf_hashgrid_fwd_jac_rows(nq, n, L, F, H, R, S, fp(x01), fp(table), fpm(feat), fpm(jac), cur_stream()), "hashgrid_fwd_jac");
    }
    const int64_t d_out = dims.back();
    Tensor attr = empty_like_opts(xs, {nq, d_out}, torch::kFloat32);
    Tensor acts = empty_like_opts(xs, {(int64_t)gsdf_mlp_acts_floats(n, nl)}, torch::kFloat32);
    check(gsdf_mlp_fwd(n, nl, dims.data(), fp(W), fp(bias), fp(feat), fpm(attr), fpm(acts), cur_stream()), "mlp_fwd");
    Tensor gt = n_ray > 0 ? f32c(gt_sdf.detach().reshape({-1}), "gt_sdf") : Tensor();
    Tensor w = n > n_ray ? f32c(weights.detach().reshape({-1}), "weights") : Tensor();
    Tensor v_attr = empty_like_opts(xs, {n, d_out}, torch::kFloat32);
    // first_order_in_forward (with unit_upstream: the caller backwards this node's output directly, gradient exactly 1): the FIRST-ORDER chain runs here, ahead of everything the
    // regularisers need.  d (data terms) / d attr needs the base rows' outputs only, so the decoder's one-pass backward and the Jacobian contraction
    // that gives the splat leg its samples' gradient — what the splats' optimizer and the next render wait for — are launched before the six
    // blocks of stencil rows go through the decoder, before the e0 backward and before the loss kernel; those move to where nobody but the SDF
    // optimizer waits for them.  The parameter gradients of the first order are accumulated here in that case, not in backward().
    const bool eager = dv.size() > 12 && dv[11] != 0.0 && dv[12] != 0.0;
    Tensor v_feat, v_samples;
    if (eager) {
      check(gsdf_sdf_data_term_grad(n, n_ray, fp(attr), (int)d_out, fp(gt), fp(w), (n > n_ray && ids.defined()) ? ids.data_ptr<int64_t>() : nullptr,
                                    (float)bce_isigma, (float)w_sdf, (float)w_gs, fpm(v_attr), cur_stream()), "sdf_data_term_grad");
      v_feat = empty_like_opts(xs, {n, nf}, torch::kFloat32);
      const bool want_samples = samples.requires_grad() && n > n_ray;
      if (want_samples) v_samples = torch::zeros({samples.size(0), 3}, xs.options().dtype(torch::kFloat32).requires_grad(false));   // (its fill in front of the backward, not between it and the contraction)
      Tensor ws = empty_like_opts(xs, {(int64_t)gsdf_mlp_bwd_ws_bytes_for(n, nl, dims.data(), 1)}, torch::kUInt8);
      check(gsdf_mlp_bwd(n, nl, dims.data(), fp(W), fp(bias), fp(feat), fp(acts), fp(v_attr), fpm(v_feat), fpm(decoder_grad),
                         bias.defined() ? fpm(bias_grad) : nullptr, ws.data_ptr(), cur_stream()), "mlp_bwd");
      if (want_samples) {
        check(gsdf_hashgrid_bwd_jac_scatter(n - n_ray, L, F, fp(jac) + n_ray * nf * 3, fp(v_feat) + n_ray * nf, (float)map_size_inv,
                                            ids.defined() ? ids.data_ptr<int64_t>() : nullptr, fpm(v_samples), cur_stream()), "hashgrid_bwd_jac_scatter");
      }
      if (auto *gate = reinterpret_cast<gsdf_extras::StreamGate *>(gate_handle)) {
        gate->record_here();
        gate->payload = v_samples.defined() ? v_samples.data_ptr() : nullptr;
      }
    }
    if (stencil)   // forward-only rows: the numerical gradient of the align term is detached
      check(gsdf_mlp_fwd(6 * n, nl, dims.data(), fp(W), fp(bias), fp(feat) + n * nf, fpm(attr) + n * d_out, nullptr, cur_stream()), "mlp_fwd");
    // g0 = d sdf / d features: the decoder's backward of e_0; its per-layer gradients stay in `bws` for the double backward
    // (constant: rows (1, 0, ..); kept between calls, re-made when the batch outgrows it)
    static thread_local Tensor e0_cache;
    if (!e0_cache.defined() || e0_cache.size(0) < n || e0_cache.size(1) != d_out || e0_cache.device() != xs.device()) {
      e0_cache = zeros_like_opts(xs, {n + n / 4 + 1024, d_out}, torch::kFloat32);
      e0_cache.select(1, 0).fill_(1.0f);
    }
    Tensor e0 = e0_cache.narrow(0, 0, n);
    Tensor g0 = empty_like_opts(xs, {n, nf}, torch::kFloat32);
    // topologies of the one-pass backward (the reference's both): the lean e0 backward — the chain alone on the bf16 pipe, nothing saved; the
    // double backward recomputes it from the ReLU masks (gsdf_mlp_bwd_bwd with bwd_ws = NULL).  Else the fp32-pipe pair with its v_pre images.
    const bool lean = gsdf_mlp_bwd_is_one_pass(nl, dims.data()) == 1;
    Tensor bws = lean ? empty_like_opts(xs, {0}, torch::kUInt8) : empty_like_opts(xs, {(int64_t)gsdf_mlp_bwd_ws_bytes(n, nl)}, torch::kUInt8);
    check(gsdf_mlp_bwd(n, nl, dims.data(), fp(W), fp(bias), fp(feat), fp(acts), fp(e0), fpm(g0), nullptr, nullptr, lean ? nullptr : bws.data_ptr(),
                       cur_stream()), "mlp_bwd");
    Tensor loss = empty_like_opts(xs, {}, torch::kFloat32);
    Tensor vv_x = empty_like_opts(xs, {n, 3}, torch::kFloat32), u0 = empty_like_opts(xs, {n, nf}, torch::kFloat32);
    check(gsdf_sdf_analytic_loss(n, n_ray, stencil ? 1 : 0, fp(attr), (int)d_out, fp(g0), nf, fp(jac), fp(gt), fp(w),
                                 (n > n_ray && ids.defined()) ? ids.data_ptr<int64_t>() : nullptr, (float)bce_isigma, (float)w_sdf, (float)w_gs,
                                 (float)map_size_inv, (float)delta, (float)w_eik, (float)w_align, fpm(loss), eager ? nullptr : fpm(v_attr), fpm(vv_x), fpm(u0),
                                 cur_stream()), "sdf_analytic_loss");
    // (tensors that travel to backward without autograd's saved-tensor version check: the in-place gradient sinks)
    ctx->save_for_backward({ids.defined() ? ids : torch::zeros({0}, xs.options().dtype(torch::kInt64)), x01, feat, jac, acts, bws, e0, g0, v_attr, vv_x, u0,
                            table, W, bias.defined() ? bias : torch::zeros({0}, xs.options())});
    ctx->saved_data["eager"] = eager;
    if (eager) {
      ctx->saved_data["v_feat"] = v_feat;
      if (v_samples.defined()) ctx->saved_data["v_samples"] = v_samples;
    }
    ctx->saved_data["table_grad"] = table_grad;
    ctx->saved_data["decoder_grad"] = decoder_grad;
    ctx->saved_data["bias_grad"] = bias_grad.defined() ? bias_grad : torch::zeros({0}, xs.options());
    ctx->saved_data["n"] = n; ctx->saved_data["n_ray"] = n_ray;
    ctx->saved_data["n_rows"] = (samples.defined() && samples.numel() > 0) ? samples.size(0) : (int64_t)0;
    ctx->saved_data["has_ids"] = ids.defined();
    ctx->saved_data["gate"] = gate_handle;
    ctx->saved_data["iv"] = iv; ctx->saved_data["dv"] = dv;
    return loss;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &ids = s[0], &x01 = s[1], &feat = s[2], &jac = s[3], &acts = s[4], &bws = s[5], &e0 = s[6], &g0 = s[7], &v_attr = s[8];
    const Tensor &vv_x = s[9], &u0 = s[10], &table = s[11], &W = s[12];
    Tensor bias = s[13].numel() ? s[13] : Tensor();
    Tensor table_grad = ctx->saved_data["table_grad"].toTensor(), decoder_grad = ctx->saved_data["decoder_grad"].toTensor();
    Tensor bias_grad = ctx->saved_data["bias_grad"].toTensor();
    auto iv = ctx->saved_data["iv"].toIntVector();
    auto dv = ctx->saved_data["dv"].toDoubleVector();
    const int L = (int)iv[0], F = (int)iv[1], H = (int)iv[2], R = (int)iv[3];
    const std::vector<int> dims(iv.begin() + 4, iv.end());
    const float S = (float)dv[0];
    const double map_size_inv = dv[4];
    const int64_t n = ctx->saved_data["n"].toInt(), n_ray = ctx->saved_data["n_ray"].toInt(), n_rows = ctx->saved_data["n_rows"].toInt();
    const int nf = L * F, nl = (int)dims.size() - 1;
    tensor_list out(14);
    if (n == 0) return out;
    // unit_upstream: the caller backwards this node's output directly (gradient exactly 1): no scaling launches
    const bool unit = dv.size() > 11 && dv[11] != 0.0;
    auto scaled = [&](const Tensor &t) { return unit ? t : (t * g[0]).contiguous(); };
    const bool eager = ctx->saved_data["eager"].toBool();   // the first order ran in forward(): its gradients are where they belong already
    Tensor v_feat;
    if (eager) {
      v_feat = ctx->saved_data["v_feat"].toTensor();
      if (ctx->saved_data.count("v_samples")) {   // the only reference leaves with the result: the engine's gradient accumulator keeps the buffer instead of copying it
        out[2] = ctx->saved_data["v_samples"].toTensor();
        ctx->saved_data.erase("v_samples");
      }
    } else {
    Tensor v_out = scaled(v_attr);
    v_feat = empty_like_opts(feat, {n, nf}, torch::kFloat32);
    // first order: data terms through the decoder (one pass: input + parameter gradients)
    Tensor ws = empty_like_opts(feat, {(int64_t)gsdf_mlp_bwd_ws_bytes_for(n, nl, dims.data(), 1)}, torch::kUInt8);
    check(gsdf_mlp_bwd(n, nl, dims.data(), fp(W), fp(bias), fp(feat), fp(acts), fp(v_out), fpm(v_feat), fpm(decoder_grad),
                       bias.defined() ? fpm(bias_grad) : nullptr, ws.data_ptr(), cur_stream()), "mlp_bwd");
    if (ctx->needs_input_grad(2) && n > n_ray) {   // d (data term) / d samples from the Jacobian of the splat rows
      const int64_t ng = n - n_ray;
      Tensor v_samples = torch::zeros({n_rows, 3}, feat.options());   // contraction, chain-rule scale and row scatter in one launch
      check(gsdf_hashgrid_bwd_jac_scatter(ng, L, F, fp(jac) + n_ray * nf * 3, fp(v_feat) + n_ray * nf, (float)map_size_inv,
                                          ctx->saved_data["has_ids"].toBool() ? ids.data_ptr<int64_t>() : nullptr, fpm(v_samples), cur_stream()),
            "hashgrid_bwd_jac_scatter");
      out[2] = v_samples;
    }
    if (auto *gate = reinterpret_cast<gsdf_extras::StreamGate *>(ctx->saved_data["gate"].toInt())) gate->record_here();
    }
    // table: first-order (v_feat) and second-order (g0, vv_x) contributions of every corner in ONE scatter.  Issued BEFORE the decoder's double backward
    // (neither needs the other): in the joint iteration this call runs beside the next render forward, whose binning passes (small, LDS-light launches)
    // then share the chip with the scatter's LDS-bound kernels instead of waiting behind decoder workgroups that own whole CUs, and whose compositing
    // forward (7 x 20 KB of LDS per CU) meets the decoder instead of the scatter: 3.87 -> 3.78 ms per step (three interleaved pairs, one box)
    Tensor vvx = scaled(vv_x);
    const size_t nb = (n >= 24576 || gsdf_deterministic(-1)) ? gsdf_hashgrid_bwd_binned_ws_bytes(n, L, F, H, R, S) : 0;   // (deterministic mode: no atomic scatter)
    if (nb > 0) {
      Tensor bws2 = empty_like_opts(feat, {(int64_t)nb}, torch::kUInt8);
      check(gsdf_hashgrid_bwd_binned2(n, L, F, H, R, S, fp(x01), fp(v_feat), fp(g0), fp(vvx), fpm(table_grad), bws2.data_ptr(), nb, cur_stream()),
            "hashgrid_bwd_binned2");
    } else {
      check(gsdf_hashgrid_bwd(n, L, F, H, R, S, fp(x01), fp(table), fp(v_feat), fpm(table_grad), nullptr, cur_stream()), "hashgrid_bwd");
      check(gsdf_hashgrid_bwd_bwd(n, L, F, H, R, S, fp(x01), fp(table), fp(g0), fp(vvx), nullptr, fpm(table_grad), nullptr, cur_stream()),
            "hashgrid_bwd_bwd");
    }
    // second order: the regularisers reach the decoder weights through g0 = W_0^T D_0 ... e_0 (nobody waits for it but the optimizer)
    Tensor vv_in = scaled(u0), g_vout = torch::empty_like(e0);
    Tensor ws2 = empty_like_opts(feat, {(int64_t)gsdf_mlp_bwd_bwd_ws_bytes(n, nl)}, torch::kUInt8);
    check(gsdf_mlp_bwd_bwd(n, nl, dims.data(), fp(W), fp(acts), fp(e0), bws.numel() ? bws.data_ptr() : nullptr, fp(vv_in), fpm(g_vout), fpm(decoder_grad),
                           ws2.data_ptr(), cur_stream()), "mlp_bwd_bwd");
    return out;
  }
};

struct NormalConsistencyFn : public torch::autograd::Function<NormalConsistencyFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &depth_, const Tensor &alpha_, const Tensor &rn_, std::vector<double> cam) {
    Tensor depth = f32c(depth_, "depth"), alpha = f32c(alpha_.detach(), "alpha"), rn = f32c(rn_, "render_normal");
    const int H = (int)depth.size(0), W = (int)depth.size(1);
    float intr[4], pose[12];
    for (int k = 0; k < 4; ++k) intr[k] = (float)cam[k];
    for (int k = 0; k < 12; ++k) pose[k] = (float)cam[4 + k];
    Tensor loss = empty_like_opts(depth, {1}, torch::kFloat32);
    check(gsdf_normal_consistency_fwd(H, W, intr, pose, fp(depth), fp(alpha), fp(rn), fpm(loss), cur_stream()), "normal_consistency_fwd");
    ctx->save_for_backward({depth, alpha, rn});
    ctx->saved_data["cam"] = cam;
    return loss[0];
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    auto cam = ctx->saved_data["cam"].toDoubleVector();
    const int H = (int)s[0].size(0), W = (int)s[0].size(1);
    float intr[4], pose[12];
    for (int k = 0; k < 4; ++k) intr[k] = (float)cam[k];
    for (int k = 0; k < 12; ++k) pose[k] = (float)cam[4 + k];
    Tensor v_depth = torch::empty_like(s[0]), v_rn = torch::empty_like(s[2]);
    Tensor v = f32c(g[0].reshape({1}), "grad");
    check(gsdf_normal_consistency_bwd(H, W, intr, pose, fp(s[0]), fp(s[1]), fp(s[2]), fp(v), fpm(v_depth), fpm(v_rn), cur_stream()),
          "normal_consistency_bwd");
    return {v_depth, Tensor(), v_rn, Tensor()};
  }
};

struct IsotropicFn : public torch::autograd::Function<IsotropicFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &scales_, const Tensor &ids_) {
    Tensor scales = f32c(scales_, "scales"), ids = ids_.contiguous();
    Tensor loss = empty_like_opts(scales, {}, torch::kFloat32);
    check(gsdf_isotropic_loss_fwd(ids.size(0), fp(scales), ids.data_ptr<int64_t>(), fpm(loss), cur_stream()), "isotropic_loss_fwd");
    ctx->save_for_backward({scales, ids});
    return loss;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    Tensor v_scales = torch::zeros_like(s[0]);
    Tensor v = f32c(g[0].reshape({1}), "grad");
    check(gsdf_isotropic_loss_bwd(s[1].size(0), fp(s[0]), s[1].data_ptr<int64_t>(), fp(v), fpm(v_scales), cur_stream()), "isotropic_loss_bwd");
    return {v_scales, Tensor()};
  }
};
}  // namespace

Tensor joint_sdf_loss_analytic(const Tensor &ray_xyz, const Tensor &gt_sdf, const Tensor &samples, const Tensor &ids, const Tensor &weights,
                               ::TCNNEncoding &enc, ::TCNNNetwork &dec, const std::vector<float> &origin, double map_size_inv, double bce_isigma,
                               double w_sdf, double w_gs, double delta, double w_eik, double w_align, Tensor table_grad, Tensor decoder_grad,
                               Tensor bias_grad, StreamGate *samples_grad_ready, bool unit_upstream, bool first_order_in_forward) {
  TORCH_CHECK(origin.size() == 3, "joint_sdf_loss_analytic: map_origin needs 3 entries");
  TORCH_CHECK(unit_upstream || !first_order_in_forward, "joint_sdf_loss_analytic: first_order_in_forward needs unit_upstream (the gradients leave before backward() is called)");
  TORCH_CHECK(table_grad.defined() && table_grad.numel() == enc.params_.numel() && table_grad.is_contiguous() && decoder_grad.defined() &&
                  decoder_grad.numel() == dec.params_.numel() && decoder_grad.is_contiguous() &&
                  (!dec.biases_.defined() || (bias_grad.defined() && bias_grad.numel() == dec.biases_.numel() && bias_grad.is_contiguous())),
              "joint_sdf_loss_analytic: table_grad / decoder_grad / bias_grad must be contiguous fp32 buffers shaped like the parameters");
  std::vector<int64_t> iv = {enc.n_levels_, enc.n_feat_, enc.log2_hashmap_, enc.base_res_};
  iv.insert(iv.end(), dec.dims_.begin(), dec.dims_.end());
  std::vector<double> dv = {enc.per_level_scale_, origin[0], origin[1], origin[2], map_size_inv, bce_isigma, w_sdf, w_gs, delta, w_eik, w_align,
                            unit_upstream ? 1.0 : 0.0, first_order_in_forward ? 1.0 : 0.0};
  // autograd::Function::apply wants defined tensors: an empty tensor stands for "absent"
  const auto opt = enc.params_.options().requires_grad(false);
  auto e = [&](const Tensor &t) { return t.defined() ? t : torch::empty({0}, opt); };
  return AnalyticFn::apply(e(ray_xyz), e(gt_sdf), e(samples), ids.defined() ? ids : torch::empty({0}, opt.dtype(torch::kInt64)), e(weights), enc.params_,
                           dec.params_, e(dec.biases_), table_grad, decoder_grad, e(bias_grad), iv, dv, reinterpret_cast<int64_t>(samples_grad_ready));
}

Tensor normal_consistency_loss(const Tensor &depth, const Tensor &alpha, const Tensor &render_normal, const std::vector<float> &intrinsics4,
                               const std::vector<float> &pose_c2w) {
  TORCH_CHECK(intrinsics4.size() == 4 && pose_c2w.size() >= 12, "normal_consistency_loss: intrinsics {fx,fy,cx,cy} and a [3,4] pose expected");
  TORCH_CHECK(depth.dim() == 3 && depth.size(2) == 1 && render_normal.dim() == 3 && render_normal.size(2) == 3, "normal_consistency_loss: [H,W,1] depth / alpha, [H,W,3] normals");
  std::vector<double> cam(intrinsics4.begin(), intrinsics4.end());
  cam.insert(cam.end(), pose_c2w.begin(), pose_c2w.begin() + 12);
  return NormalConsistencyFn::apply(depth, alpha, render_normal, cam);
}

Tensor isotropic_loss(const Tensor &scales, const Tensor &gaussian_ids) {
  TORCH_CHECK(scales.dim() == 2 && scales.size(1) == 3 && gaussian_ids.scalar_type() == torch::kInt64, "isotropic_loss: scales [N,3], gaussian_ids int64");
  return IsotropicFn::apply(scales, gaussian_ids);
}

Tensor nan_rows(const Tensor &offsets, const Tensor &scaling, const Tensor &quaternion, Tensor *mask) {
  torch::NoGradGuard ng;
  Tensor o = f32c(offsets.detach(), "offsets"), sc = f32c(scaling.detach(), "scaling"), q = f32c(quaternion.detach(), "quaternion");
  const int64_t n = o.size(0);
  Tensor count = empty_like_opts(o, {1}, torch::kInt32);
  Tensor m = mask ? empty_like_opts(o, {n}, torch::kBool) : Tensor();
  check(gsdf_nan_rows(n, fp(o), fp(sc), fp(q), count.data_ptr<int32_t>(), mask ? (uint8_t *)m.data_ptr() : nullptr, cur_stream()), "nan_rows");
  if (mask) *mask = m;
  return count;
}

Tensor gs_sdf_coupling(const Tensor &samples, const Tensor &ids, const Tensor &weights, ::TCNNEncoding &enc, ::TCNNNetwork &dec,
                       const std::vector<float> &origin, double map_size_inv, double scale, double delta, double w_eik, Tensor table_grad,
                       Tensor decoder_grad, StreamGate *samples_grad_ready) {
  TORCH_CHECK(origin.size() == 3, "gs_sdf_coupling: map_origin needs 3 entries");
  TORCH_CHECK(samples.dim() == 2 && samples.size(1) == 3 && ids.scalar_type() == torch::kInt64, "gs_sdf_coupling: samples [M,3], ids int64");
  TORCH_CHECK(table_grad.defined() && table_grad.numel() == enc.params_.numel() && table_grad.is_contiguous() &&
                  decoder_grad.defined() && decoder_grad.numel() == dec.params_.numel() && decoder_grad.is_contiguous() &&
                  table_grad.scalar_type() == torch::kFloat32 && decoder_grad.scalar_type() == torch::kFloat32,
              "gs_sdf_coupling: table_grad / decoder_grad must be contiguous fp32 buffers shaped like the parameters");
  std::vector<int64_t> iv = {enc.n_levels_, enc.n_feat_, enc.log2_hashmap_, enc.base_res_};
  iv.insert(iv.end(), dec.dims_.begin(), dec.dims_.end());
  std::vector<double> dv = {enc.per_level_scale_, origin[0], origin[1], origin[2], map_size_inv, scale, delta, w_eik};
  return CouplingFn::apply(samples, ids, weights, enc.params_, dec.params_, table_grad, decoder_grad, iv, dv,
                           reinterpret_cast<int64_t>(samples_grad_ready));
}

void update_state(std::map<std::string, Tensor> &state, const Tensor &densify_grad, const Tensor &gaussian_ids, const Tensor &visibilities,
                  const Tensor &radii, int64_t n_gaussians, int n_cameras, int width, int height, bool want_radii) {
  torch::NoGradGuard ng;
  Tensor g = f32c(densify_grad, "densify gradient");
  for (const char *k : {"grad2d", "count", "vis", "radii"}) {
    if (std::string(k) == "radii" && !want_radii) continue;
    if (!state.count(k)) state[k] = torch::zeros({n_gaussians}, g.options());
  }
  Tensor ids = gaussian_ids.contiguous(), vis = f32c(visibilities.reshape({-1}), "visibilities");
  Tensor rp = want_radii ? radii.contiguous() : Tensor();
  check(gsdf_densify_stats(ids.size(0), n_gaussians, n_cameras, width, height, fp(g), ids.data_ptr<int64_t>(), fp(vis),
                           want_radii ? rp.data_ptr<int32_t>() : nullptr, fpm(state["grad2d"]), fpm(state["count"]), fpm(state["vis"]),
                           want_radii ? fpm(state["radii"]) : nullptr, cur_stream()),
        "densify_stats");
}

std::vector<Tensor> splat_activations(const Tensor &anchors, const Tensor &offsets, const Tensor &scaling, const Tensor &opacity) {
  return Activate::apply(anchors, offsets, scaling, opacity);
}

int FusedAdam::add_group(const Tensor &flat, const Tensor &flat_grad, const std::vector<int64_t> &sizes, const std::vector<double> &lrs) {
  TORCH_CHECK(flat.is_cuda() && flat.is_contiguous() && flat.scalar_type() == torch::kFloat32 && flat_grad.sizes() == flat.sizes(),
              "FusedAdam: flat fp32 device buffers expected");
  TORCH_CHECK(sizes.size() == lrs.size() && !sizes.empty(), "FusedAdam: one learning rate per segment");
  Group g{flat, flat_grad, torch::zeros_like(flat), torch::zeros_like(flat), {}, {}};
  int64_t off = 0;
  for (size_t i = 0; i < sizes.size(); ++i) { g.begins.push_back(off); off += sizes[i]; g.lrs.push_back((float)lrs[i]); }
  TORCH_CHECK(off == flat.numel(), "FusedAdam: segments cover ", off, " of ", flat.numel(), " elements");
  groups_.push_back(std::move(g));
  return (int)groups_.size() - 1;
}

void FusedAdam::step(bool zero_grad) {
  torch::NoGradGuard ng;
  ++t_;
  for (auto &g : groups_)
    check((zero_grad ? gsdf_adam_step_zero_grad(g.flat.numel(), (int)g.lrs.size(), g.begins.data(), g.lrs.data(), fpm(g.flat), fpm(g.grad), fpm(g.m),
                                                fpm(g.v), (float)b1_, (float)b2_, (float)eps_, t_, cur_stream())
                     : gsdf_adam_step(g.flat.numel(), (int)g.lrs.size(), g.begins.data(), g.lrs.data(), fpm(g.flat), fp(g.grad), fpm(g.m), fpm(g.v),
                                      (float)b1_, (float)b2_, (float)eps_, t_, cur_stream())),
          "adam_step");
}

// The step of group 0 in two launches: step_tail() — the elements behind the first `head_segments` segments (from the next multiple of 4 on: the
// kernel's vectors stay aligned) and every other group, the step count advances — and later step_head(), the elements before that point.  For a
// caller whose first segments' gradient arrives last (the joint iteration: the offsets' gradient waits for the SDF leg, the other 11 of a splat's
// 14 parameters do not).  Elementwise, so the two launches write the bits step() writes.
static void adam_range(const Tensor &flat, const Tensor &grad, const Tensor &m, const Tensor &v, const std::vector<int64_t> &begins, const std::vector<float> &lrs,
                       int64_t e0, int64_t e1, bool zero_grad, double b1, double b2, double eps, int64_t t) {
  if (e1 <= e0) return;
  std::vector<int64_t> bg;
  std::vector<float> lr;
  for (size_t k = 0; k < begins.size(); ++k) {
    const int64_t end = k + 1 < begins.size() ? begins[k + 1] : flat.numel();
    if (end <= e0 && k + 1 < begins.size()) continue;   // wholly before the range
    if (begins[k] >= e1) break;
    bg.push_back(std::max<int64_t>(begins[k] - e0, 0));
    lr.push_back(lrs[k]);
  }
  float *p = flat.data_ptr<float>() + e0, *g = grad.data_ptr<float>() + e0, *mm = m.data_ptr<float>() + e0, *vv = v.data_ptr<float>() + e0;
  check(zero_grad ? gsdf_adam_step_zero_grad(e1 - e0, (int)lr.size(), bg.data(), lr.data(), p, g, mm, vv, (float)b1, (float)b2, (float)eps, t, cur_stream())
                  : gsdf_adam_step(e1 - e0, (int)lr.size(), bg.data(), lr.data(), p, g, mm, vv, (float)b1, (float)b2, (float)eps, t, cur_stream()),
        "adam_step");
}
static int64_t head_end(const std::vector<int64_t> &begins, int head_segments, int64_t n) {
  const int64_t b = head_segments < (int)begins.size() ? begins[head_segments] : n;
  return std::min<int64_t>((b + 3) / 4 * 4, n);
}
void FusedAdam::step_tail(int head_segments, bool zero_grad) {
  torch::NoGradGuard ng;
  ++t_;
  for (size_t gi = 0; gi < groups_.size(); ++gi) {
    Group &g = groups_[gi];
    adam_range(g.flat, g.grad, g.m, g.v, g.begins, g.lrs, gi == 0 ? head_end(g.begins, head_segments, g.flat.numel()) : 0, g.flat.numel(), zero_grad, b1_, b2_,
               eps_, t_);
  }
}
void FusedAdam::step_head(int head_segments, bool zero_grad) {
  torch::NoGradGuard ng;
  Group &g = groups_.at(0);
  adam_range(g.flat, g.grad, g.m, g.v, g.begins, g.lrs, 0, head_end(g.begins, head_segments, g.flat.numel()), zero_grad, b1_, b2_, eps_, t_);
}

void FusedAdam::replace_group(int group, const Tensor &flat, const Tensor &flat_grad, const Tensor &m, const Tensor &v, const std::vector<int64_t> &sizes) {
  Group &g = groups_.at(group);
  TORCH_CHECK(sizes.size() == g.lrs.size(), "FusedAdam::replace_group: ", sizes.size(), " segments, the group has ", g.lrs.size());
  TORCH_CHECK(flat.is_cuda() && flat.is_contiguous() && flat.scalar_type() == torch::kFloat32 && flat_grad.sizes() == flat.sizes() &&
              m.sizes() == flat.sizes() && v.sizes() == flat.sizes(), "FusedAdam::replace_group: flat fp32 device buffers of one size expected");
  int64_t off = 0;
  for (size_t i = 0; i < sizes.size(); ++i) { g.begins[i] = off; off += sizes[i]; }
  TORCH_CHECK(off == flat.numel(), "FusedAdam::replace_group: segments cover ", off, " of ", flat.numel(), " elements");
  g.flat = flat; g.grad = flat_grad; g.m = m; g.v = v;
}

void FusedAdam::zero_segment_moments(int group, int segment) {
  Group &g = groups_.at(group);
  const int64_t b = g.begins.at(segment), e = (size_t)segment + 1 < g.begins.size() ? g.begins[segment + 1] : g.flat.numel();
  torch::NoGradGuard ng;
  g.m.slice(0, b, e).zero_();
  g.v.slice(0, b, e).zero_();
}

}  // namespace gsdf_extras
