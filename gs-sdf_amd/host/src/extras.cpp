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
    Tensor xs = f32c(samples.detach().index_select(0, ids), "samples");
    const int64_t n = xs.size(0), K = c.delta > 0 ? 7 : 1, nq = K * n;
    const int nf = c.L * c.F, nl = (int)c.dims.size() - 1;
    Tensor x01 = empty_like_opts(xs, {nq, 3}, torch::kFloat32);
    check(gsdf_sdf_query_points(n, K == 7, fp(xs), (float)c.delta, c.origin, (float)c.map_size_inv, fpm(x01), cur_stream()), "sdf_query_points");
    Tensor feat = empty_like_opts(xs, {nq, nf}, torch::kFloat32), jac = empty_like_opts(xs, {n, nf, 3}, torch::kFloat32);
    if (K == 7 && n > 0)
      check(gsdf_hashgrid_fwd_stencil(nq, n, n, c.L, c.F, c.H, c.R, c.S, fp(x01), fp(table), fpm(feat), fpm(jac), cur_stream()), "hashgrid_fwd_stencil");
    else
      check(gsdf_hashgrid_fwd_jac_rows(nq, n, c.L, c.F, c.H, c.R, c.S, fp(x01), fp(table), fpm(feat), fpm(jac), cur_stream()), "hashgrid_fwd_jac");
    Tensor attr = empty_like_opts(xs, {nq, (int64_t)c.dims.back()}, torch::kFloat32);
    Tensor acts = empty_like_opts(xs, {(int64_t)gsdf_mlp_acts_floats(nq, nl)}, torch::kFloat32);
    check(gsdf_mlp_fwd(nq, nl, c.dims.data(), fp(W), nullptr, fp(feat), fpm(attr), fpm(acts), cur_stream()), "mlp_fwd");
    Tensor loss = empty_like_opts(xs, {}, torch::kFloat32), v_attr = torch::empty_like(attr);
    Tensor w = f32c(weights.detach().reshape({-1}), "weights");
    check(gsdf_gs_sdf_eik_loss(n, K == 7, fp(attr), (int)attr.size(1), fp(w), ids.data_ptr<int64_t>(), (float)c.scale, (float)c.delta,
                               (float)c.w_eik, fpm(loss), fpm(v_attr), cur_stream()), "gs_sdf_eik_loss");
    ctx->save_for_backward({ids, x01, feat, jac, acts, v_attr, table, W, table_grad, decoder_grad});
    ctx->saved_data["n_rows"] = samples.size(0);
    ctx->saved_data["gate"] = gate_handle;
    ctx->saved_data["iv"] = iv;
    ctx->saved_data["dv"] = dv;
    return loss;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &ids = s[0], &x01 = s[1], &feat = s[2], &jac = s[3], &acts = s[4], &v_attr = s[5], &table = s[6], &W = s[7];
    Tensor table_grad = s[8], decoder_grad = s[9];
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
    if (nq >= 24576) {
      int merge = 0;
      for (int l = 0; l < c.L; ++l)
        if ((c.R * std::pow((double)c.S, l) - 1.0) * c.delta * c.map_size_inv < 1.0) ++merge;
      const size_t nb = gsdf_hashgrid_bwd_binned_ws_bytes(nq, c.L, c.F, c.H, c.R, c.S);
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

void FusedAdam::step() {
  torch::NoGradGuard ng;
  ++t_;
  for (auto &g : groups_)
    check(gsdf_adam_step(g.flat.numel(), (int)g.lrs.size(), g.begins.data(), g.lrs.data(), fpm(g.flat), fp(g.grad), fpm(g.m), fpm(g.v),
                         (float)b1_, (float)b2_, (float)eps_, t_, cur_stream()),
          "adam_step");
}

}  // namespace gsdf_extras
