// tcnn_ops.cpp — libtorch operator layer for the SDF network: TCNNEncoding (hash grid with first and second
// order autograd) and TCNNNetwork (fused fp32-MFMA MLP), as the reference uses them
// (/root/reference/include/neural_net/encoding_map.cpp:15-26,59; local_map.cpp:44-55,94,151-172).
#include "tcnn_binding/tcnn_binding.h"
#include "util.h"

using namespace gsdf_host;
using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

namespace {

struct GridCfg { int L, F, H, R; float S; int64_t stencil_n = 0; int merge_levels = 0; };   // stencil_n > 0: forward_stencil batches
GridCfg cfg_of(const c10::IValue &v) {
  auto t = v.toIntVector();
  float s;
  int32_t bits = (int32_t)t[4];
  std::memcpy(&s, &bits, 4);
  return {(int)t[0], (int)t[1], (int)t[2], (int)t[3], s, t.size() > 5 ? t[5] : 0, t.size() > 6 ? (int)t[6] : 0};
}
std::vector<int64_t> cfg_pack(const GridCfg &c) {
  int32_t bits;
  std::memcpy(&bits, &c.S, 4);
  return {c.L, c.F, c.H, c.R, bits, c.stencil_n, c.merge_levels};
}

// Orders of differentiation.  LocalMap::get_gradient with hessian = true and numerical_grad = 0, then a loss on the Hessian (the reference's
// curvate_weight > 0, /root/reference/include/neural_net/local_map.cpp:163-168, neural_mapping.cpp:117-121; default 0.0, config/base.yaml:32) takes a
// THIRD derivative of the encoding: GridFwd -> GridBwd -> GridBwd2 (the double backward as an operator of its own, round 6) -> gsdf_hashgrid_bwd_bwd_bwd.
// The decoder's part of that loss needs nothing new: the Hessian row sums depend on the decoder through v_feat = d sdf / d feat only, and d loss /
// d v_feat arrives at MlpBwd's double backward like the eikonal term's.  What is evaluated by kernels inside a backward() and has no operator of
// its own — the decoder's double backward, the encoding's third order — passes through this node when create_graph = true: the VALUES are exact,
// differentiating them again (a fourth derivative of the encoding, a third of the decoder's weight path) raises instead of silently missing a term.
struct ThirdOrderGuard : public torch::autograd::Function<ThirdOrderGuard> {
  static Tensor forward(AutogradContext *, const Tensor &value, const Tensor &anchor) {
    (void)anchor;   // an input that requires grad: it puts this node on the graph
    return value.view_as(value);
  }
  static tensor_list backward(AutogradContext *, tensor_list) {
    TORCH_CHECK(false, "TCNNEncoding / TCNNNetwork: derivatives of this order are not implemented (the encoding is differentiable three times — a "
                       "loss on the analytic Hessian trains —, the decoder's weight path twice)");
    return {};
  }
};
static Tensor guard3(const Tensor &value, const Tensor &anchor) {
  if (!value.defined() || !torch::GradMode::is_enabled() || !anchor.defined() || !anchor.requires_grad()) return value;
  return ThirdOrderGuard::apply(value, anchor);
}

struct GridBwd : public torch::autograd::Function<GridBwd> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &v_feat_, const Tensor &x, const Tensor &table,
                             std::vector<int64_t> cfgv, bool want_table) {
    const GridCfg c = cfg_of(c10::IValue(cfgv));
    Tensor v_feat = f32c(v_feat_, "v_feat");
    const int64_t B = x.size(0);
    Tensor v_x = torch::empty_like(x);
    Tensor v_table = want_table ? torch::zeros_like(table) : Tensor();
    // large batches: the table gradient without global atomics (gsdf_hashgrid_bwd_binned), d/dx from the plain kernel
    const size_t binned = (want_table && (B >= 24576 || gsdf_deterministic(-1))) ? gsdf_hashgrid_bwd_binned_ws_bytes(B, c.L, c.F, c.H, c.R, c.S) : 0;
    if (binned) {
      Tensor ws = empty_like_opts(x, {(int64_t)binned}, torch::kUInt8);
      check(gsdf_hashgrid_bwd(B, c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fp(v_feat), nullptr, fpm(v_x), cur_stream()),
            "TCNNEncoding backward");
      check(gsdf_hashgrid_bwd_binned_stencil(B, c.stencil_n * 7 == B ? c.stencil_n : 0, c.merge_levels, c.L, c.F, c.H, c.R, c.S, fp(x),
                                             fp(v_feat), fpm(v_table), ws.data_ptr(), binned, cur_stream()),
            "TCNNEncoding backward (binned scatter)");
    } else {
      check(gsdf_hashgrid_bwd(B, c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fp(v_feat), fpm(v_table), fpm(v_x), cur_stream()),
            "TCNNEncoding backward");
    }
    ctx->save_for_backward({v_feat, x, table});
    ctx->saved_data["cfg"] = cfgv;
    if (!want_table) {
      v_table = torch::zeros({0}, x.options());
      ctx->mark_non_differentiable({v_table});
    }
    return {v_x, v_table};
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g);
};

// the grid's double backward: (vv_x, v_feat, x, table) -> (g_vfeat, g_x, g_table), one kernel (+ the binned scatter for large batches)
static void grid_double_backward(const GridCfg &c, const Tensor &vv, const Tensor &v_feat, const Tensor &x, const Tensor &table, Tensor &g_vfeat,
                                 Tensor &g_x, Tensor &g_table) {
  const int64_t B = x.size(0);
  // large batches: the table part of the double backward through the binned scatter's second-order form (no global atomics)
  const size_t binned = (g_table.defined() && (B >= 24576 || gsdf_deterministic(-1))) ? gsdf_hashgrid_bwd_binned_ws_bytes(B, c.L, c.F, c.H, c.R, c.S) : 0;
  if (binned) {
    Tensor ws = empty_like_opts(x, {(int64_t)binned}, torch::kUInt8);
    check(gsdf_hashgrid_bwd_binned2(B, c.L, c.F, c.H, c.R, c.S, fp(x), nullptr, fp(v_feat), fp(vv), fpm(g_table), ws.data_ptr(), binned,
                                    cur_stream()), "TCNNEncoding double backward (binned scatter)");
  }
  if (g_vfeat.defined() || g_x.defined() || (g_table.defined() && !binned))
    check(gsdf_hashgrid_bwd_bwd(B, c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fp(v_feat), fp(vv), fpm(g_vfeat),
                                binned ? nullptr : fpm(g_table), fpm(g_x), cur_stream()),
          "TCNNEncoding double backward");
}

// The double backward as an operator (create_graph = true while it runs: LocalMap::get_gradient's second autograd::grad call, local_map.cpp:163-166);
// its own backward is the encoding's THIRD order, gsdf_hashgrid_bwd_bwd_bwd.
struct GridBwd2 : public torch::autograd::Function<GridBwd2> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &vv_, const Tensor &v_feat, const Tensor &x, const Tensor &table,
                             std::vector<int64_t> cfgv, bool want_vfeat, bool want_x, bool want_table) {
    const GridCfg c = cfg_of(c10::IValue(cfgv));
    Tensor vv = f32c(vv_, "vv_x");
    Tensor g_vfeat = want_vfeat ? torch::empty_like(v_feat) : Tensor();
    Tensor g_x = want_x ? torch::empty_like(x) : Tensor();
    Tensor g_table = want_table ? torch::zeros_like(table) : Tensor();
    grid_double_backward(c, vv, v_feat, x, table, g_vfeat, g_x, g_table);
    ctx->save_for_backward({vv, v_feat, x, table});
    ctx->saved_data["cfg"] = cfgv;
    tensor_list out = {g_vfeat, g_x, g_table};
    std::vector<Tensor> dead;
    for (int i = 0; i < 3; ++i) {
      if (!out[i].defined()) out[i] = torch::zeros({0}, x.options());
      // (a loss on the TABLE gradient of the double backward is nobody's path: not differentiable here)
      if (out[i].numel() == 0 || i == 2) dead.push_back(out[i]);
    }
    ctx->mark_non_differentiable(dead);
    return out;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &vv = s[0], &v_feat = s[1], &x = s[2], &table = s[3];
    const GridCfg c = cfg_of(ctx->saved_data["cfg"]);
    tensor_list none(8);
    if (!g[0].defined() && !g[1].defined()) return none;
    const int64_t B = x.size(0);
    Tensor mu = g[0].defined() ? f32c(g[0], "gradient of g_vfeat") : Tensor();
    Tensor lam = g[1].defined() ? f32c(g[1], "gradient of g_x") : torch::zeros_like(x);
    Tensor t_vv = ctx->needs_input_grad(0) ? torch::empty_like(vv) : Tensor();
    Tensor t_vfeat = ctx->needs_input_grad(1) ? torch::empty_like(v_feat) : Tensor();
    Tensor t_x = ctx->needs_input_grad(2) ? torch::empty_like(x) : Tensor();
    Tensor t_table = ctx->needs_input_grad(3) ? torch::zeros_like(table) : Tensor();
    check(gsdf_hashgrid_bwd_bwd_bwd(B, c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fp(v_feat), fp(vv), fp(lam), fp(mu), fpm(t_vfeat), fpm(t_table),
                                    fpm(t_vv), fpm(t_x), cur_stream()), "TCNNEncoding third-order backward");
    const Tensor &anchor = table.requires_grad() ? table : (x.requires_grad() ? x : v_feat);
    return {guard3(t_vv, anchor), guard3(t_vfeat, anchor), guard3(t_x, anchor), guard3(t_table, anchor), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

tensor_list GridBwd::backward(AutogradContext *ctx, tensor_list g) {
  auto s = ctx->get_saved_variables();
  const Tensor &v_feat = s[0], &x = s[1], &table = s[2];
  if (!g[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  if (torch::GradMode::is_enabled()) {   // create_graph = true: the double backward stays on the graph (third order)
    auto o = GridBwd2::apply(g[0], v_feat, x, table, ctx->saved_data["cfg"].toIntVector(), ctx->needs_input_grad(0), ctx->needs_input_grad(1),
                             ctx->needs_input_grad(2));
    return {ctx->needs_input_grad(0) ? o[0] : Tensor(), ctx->needs_input_grad(1) ? o[1] : Tensor(), ctx->needs_input_grad(2) ? o[2] : Tensor(), Tensor(),
            Tensor()};
  }
  const GridCfg c = cfg_of(ctx->saved_data["cfg"]);
  Tensor vv = f32c(g[0], "vv_x");
  Tensor g_vfeat = ctx->needs_input_grad(0) ? torch::empty_like(v_feat) : Tensor();
  Tensor g_x = ctx->needs_input_grad(1) ? torch::empty_like(x) : Tensor();
  Tensor g_table = ctx->needs_input_grad(2) ? torch::zeros_like(table) : Tensor();
  grid_double_backward(c, vv, v_feat, x, table, g_vfeat, g_x, g_table);
  return {g_vfeat, g_x, g_table, Tensor(), Tensor()};
}

struct GridFwd : public torch::autograd::Function<GridFwd> {
  static Tensor forward(AutogradContext *ctx, const Tensor &x_, const Tensor &table_, std::vector<int64_t> cfgv) {
    const GridCfg c = cfg_of(c10::IValue(cfgv));
    Tensor x = f32c(x_, "x"), table = f32c(table_, "params");
    const int64_t B = x.size(0);
    Tensor feat = empty_like_opts(x, {B, (int64_t)c.L * c.F}, torch::kFloat32);
    if (c.stencil_n > 0 && c.stencil_n * 7 == B)
      check(gsdf_hashgrid_fwd_stencil(B, c.stencil_n, 0, c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fpm(feat), nullptr, cur_stream()),
            "TCNNEncoding forward (stencil batch)");
    else
      check(gsdf_hashgrid_fwd(B, c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fpm(feat), cur_stream()), "TCNNEncoding forward");
    ctx->save_for_backward({x, table});
    ctx->saved_data["cfg"] = cfgv;
    return feat;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    auto o = GridBwd::apply(g[0], s[0], s[1], ctx->saved_data["cfg"].toIntVector(), ctx->needs_input_grad(1));
    return {ctx->needs_input_grad(0) ? o[0] : Tensor(), ctx->needs_input_grad(1) ? o[1] : Tensor(), Tensor()};
  }
};

std::vector<int> to_int(const std::vector<int64_t> &v) { return std::vector<int>(v.begin(), v.end()); }

// (v_out, x, w[, b], acts) -> (v_in, v_w, v_b): the decoder's first-order backward as a differentiable operator; its own backward
// is gsdf_mlp_bwd_bwd (masked bias-free forward for dL/d v_out + the weight-gradient GEMM on (vv_in, masked activations)).
struct MlpBwd : public torch::autograd::Function<MlpBwd> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &v_out_, const Tensor &x, const Tensor &w, const Tensor &b_, const Tensor &acts,
                             std::vector<int64_t> dims64, bool want_w, bool want_b) {
    const Tensor b = b_.numel() > 0 ? b_ : Tensor();
    const std::vector<int> dims = to_int(dims64);
    const int nl = (int)dims.size() - 1;
    const int64_t B = x.size(0);
    Tensor v_out = f32c(v_out_, "grad");
    Tensor v_in = torch::empty_like(x);
    Tensor ws = empty_like_opts(x, {(int64_t)gsdf_mlp_bwd_ws_bytes(B, nl)}, torch::kUInt8);
    // the two-kernel form: the per-layer gradients stay in `ws` for the double backward
    check(gsdf_mlp_bwd(B, nl, dims.data(), fp(w), fp(b), fp(x), fp(acts), fp(v_out), fpm(v_in), nullptr, nullptr, ws.data_ptr(), cur_stream()),
          "TCNNNetwork backward");
    Tensor v_w = want_w ? torch::zeros_like(w) : torch::zeros({0}, x.options());
    Tensor v_b = (want_b && b.defined()) ? torch::zeros_like(b) : torch::zeros({0}, x.options());
    if (want_w)
      check(gsdf_mlp_bwd_weights(B, nl, dims.data(), (want_b && b.defined()) ? 1 : 0, fp(x), fp(acts), fp(v_out), ws.data_ptr(), fpm(v_w),
                                 (want_b && b.defined()) ? fpm(v_b) : nullptr, cur_stream()), "TCNNNetwork backward (weights)");
    ctx->save_for_backward({v_out, w, acts, ws});
    ctx->saved_data["dims"] = dims64;
    ctx->mark_non_differentiable({v_w, v_b});
    return {v_in, v_w, v_b};
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &v_out = s[0], &w = s[1], &acts = s[2], &ws = s[3];
    const std::vector<int> dims = to_int(ctx->saved_data["dims"].toIntVector());
    const int nl = (int)dims.size() - 1;
    if (!g[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    Tensor vv_in = f32c(g[0], "vv_in");
    const int64_t B = vv_in.size(0);
    Tensor g_vout = torch::empty_like(v_out);
    Tensor g_w = ctx->needs_input_grad(2) ? torch::zeros_like(w) : Tensor();
    Tensor ws2 = empty_like_opts(vv_in, {(int64_t)gsdf_mlp_bwd_bwd_ws_bytes(B, nl)}, torch::kUInt8);
    check(gsdf_mlp_bwd_bwd(B, nl, dims.data(), fp(w), fp(acts), fp(v_out), ws.data_ptr(), fp(vv_in), fpm(g_vout), fpm(g_w), ws2.data_ptr(),
                           cur_stream()), "TCNNNetwork double backward");
    const Tensor &anchor = w.requires_grad() ? w : v_out;
    return {ctx->needs_input_grad(0) ? guard3(g_vout, anchor) : Tensor(), Tensor(), guard3(g_w, anchor), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

struct MlpFn : public torch::autograd::Function<MlpFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &x_, const Tensor &w_, const Tensor &b_, std::vector<int64_t> dims64) {
    Tensor x = f32c(x_, "x"), w = f32c(w_, "params");
    Tensor b = (b_.defined() && b_.numel() > 0) ? f32c(b_, "biases") : Tensor();     // an empty tensor stands for "no biases"
    const std::vector<int> dims = to_int(dims64);
    const int nl = (int)dims.size() - 1;
    const int64_t B = x.size(0);
    Tensor out = empty_like_opts(x, {B, dims.back()}, torch::kFloat32);
    const bool need = x.requires_grad() || w.requires_grad() || (b.defined() && b.requires_grad());
    Tensor acts = need ? empty_like_opts(x, {(int64_t)gsdf_mlp_acts_floats(B, nl)}, torch::kFloat32) : Tensor();
    check(gsdf_mlp_fwd(B, nl, dims.data(), fp(w), fp(b), fp(x), fpm(out), fpm(acts), cur_stream()), "TCNNNetwork forward");
    ctx->save_for_backward({x, w, acts.defined() ? acts : torch::empty({0}, x.options()), b.defined() ? b : torch::empty({0}, x.options())});
    ctx->saved_data["dims"] = dims64;
    return out;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &x = s[0], &w = s[1], &acts = s[2];
    const Tensor b = s[3].numel() > 0 ? s[3] : Tensor();
    auto dims64 = ctx->saved_data["dims"].toIntVector();
    const bool want_w = ctx->needs_input_grad(1), want_b = b.defined() && ctx->needs_input_grad(2);
    if (torch::GradMode::is_enabled()) {
      // create_graph = true (LocalMap::get_gradient's analytic branch, local_map.cpp:151-172): the backward as a differentiable op
      auto o = MlpBwd::apply(g[0], x, w, b.defined() ? b : torch::empty({0}, x.options()), acts, dims64, want_w, want_b);
      return {ctx->needs_input_grad(0) ? o[0] : Tensor(), want_w ? o[1] : Tensor(), want_b ? o[2] : Tensor(), Tensor()};
    }
    const std::vector<int> dims = to_int(dims64);
    const int nl = (int)dims.size() - 1;
    const int64_t B = x.size(0);
    Tensor v_out = f32c(g[0], "grad");
    Tensor v_in = ctx->needs_input_grad(0) ? torch::empty_like(x) : Tensor();
    Tensor v_w = want_w ? torch::zeros_like(w) : Tensor();
    Tensor v_b = want_b ? torch::zeros_like(b) : Tensor();
    // 0 bytes when both gradients are requested and the one-pass backward covers the topology
    Tensor ws = empty_like_opts(x, {(int64_t)gsdf_mlp_bwd_ws_bytes_for(B, nl, dims.data(), v_w.defined() ? 1 : 0)}, torch::kUInt8);
    check(gsdf_mlp_bwd(B, nl, dims.data(), fp(w), fp(b), fp(x), fp(acts), fp(v_out), fpm(v_in), fpm(v_w), fpm(v_b), ws.data_ptr(), cur_stream()),
          "TCNNNetwork backward");
    return {v_in, v_w, v_b, Tensor()};
  }
};
}  // namespace

// tiny-cuda-nn's SphericalHarmonics encoding (the reference's SHEncoding, include/neural_net/encodings/encodings.h:6-27: declared, never
// evaluated by the training path): x in [0,1]^3 is the direction (d + 1) / 2; outputs the degree^2 real spherical-harmonic basis values of
// d = 2 x - 1 in tiny-cuda-nn's order and sign convention (the basis of SPEC A.2 / csrc/view_colors.hip).  Parameter-free; composed of
// libtorch elementwise operations (autograd of any order for free) — not a hot path, so no kernel of its own.
static torch::Tensor sh_encoding(const torch::Tensor &x01, int degree) {
  TORCH_CHECK(degree >= 1 && degree <= 4, "SphericalHarmonics encoding: degree must be in [1, 4], got ", degree);
  torch::Tensor d = x01 * 2.0 - 1.0;
  torch::Tensor x = d.select(1, 0), y = d.select(1, 1), z = d.select(1, 2);
  std::vector<torch::Tensor> o;
  o.push_back(torch::full_like(x, 0.28209479177387814));
  if (degree > 1) {
    o.push_back(-0.48860251190291987 * y);
    o.push_back(0.48860251190291987 * z);
    o.push_back(-0.48860251190291987 * x);
  }
  if (degree > 2) {
    torch::Tensor x2 = x * x, y2 = y * y, z2 = z * z;
    o.push_back(1.0925484305920792 * (x * y));
    o.push_back(-1.0925484305920792 * (y * z));
    o.push_back(0.94617469575755997 * z2 - 0.31539156525251999);
    o.push_back(-1.0925484305920792 * (x * z));
    o.push_back(0.54627421529603959 * x2 - 0.54627421529603959 * y2);
    if (degree > 3) {
      o.push_back(0.59004358992664352 * y * (-3.0 * x2 + y2));
      o.push_back(2.8906114426405538 * (x * y) * z);
      o.push_back(0.45704579946446572 * y * (1.0 - 5.0 * z2));
      o.push_back(0.3731763325901154 * z * (5.0 * z2 - 3.0));
      o.push_back(0.45704579946446572 * x * (1.0 - 5.0 * z2));
      o.push_back(1.4453057213202769 * z * (x2 - y2));
      o.push_back(0.59004358992664352 * x * (-x2 + 3.0 * y2));
    }
  }
  return torch::stack(o, 1);
}

void TCNNEncoding::init_encoding(int n_input_dims, const nlohmann::json &config, const std::string &name) {
  name_ = name;
  n_input_dims_ = n_input_dims;
  otype_ = config.value("otype", std::string("Grid"));
  if (otype_ == "SphericalHarmonics") {  // declared by the reference (encodings.h:6-27) but never evaluated in training; parameter-free
    sh_degree_ = config.value("degree", 4);
    TORCH_CHECK(n_input_dims == 3, "TCNNEncoding: the SphericalHarmonics encoding takes 3 input dims");
    params_ = torch::empty({0});
    return;
  }
  TORCH_CHECK(otype_ == "Grid" || otype_ == "HashGrid", "TCNNEncoding: otype '", otype_, "' is not implemented");
  TORCH_CHECK(config.value("type", std::string("Hash")) == "Hash", "TCNNEncoding: only type Hash is implemented");
  TORCH_CHECK(config.value("interpolation", std::string("Linear")) == "Linear", "TCNNEncoding: only Linear interpolation");
  TORCH_CHECK(n_input_dims == 3, "TCNNEncoding: only 3 input dims are implemented");
  n_levels_ = config.value("n_levels", 16);
  n_feat_ = config.value("n_features_per_level", 2);
  log2_hashmap_ = config.value("log2_hashmap_size", 19);
  base_res_ = config.value("base_resolution", 16);
  per_level_scale_ = (float)config.value("per_level_scale", 2.0);
  offsets_.assign(n_levels_ + 1, 0);
  const int64_t total = gsdf_hashgrid_offsets(n_levels_, n_feat_, log2_hashmap_, base_res_, per_level_scale_, offsets_.data());
  TORCH_CHECK(total >= 0, "TCNNEncoding: ", gsdf_last_error());
  // tiny-cuda-nn initialises grid parameters U(-1e-4, 1e-4); created on the CPU like the reference's module and
  // moved with .to(device) by the caller (encoding_map.cpp:62-74), or directly on the current device when available
  params_ = (torch::rand({total * n_feat_}) * 2.0f - 1.0f) * 1e-4f;
  if (torch::cuda::is_available()) params_ = params_.to(torch::kCUDA);
  params_.set_requires_grad(true);
}

torch::Tensor TCNNEncoding::forward(const torch::Tensor &x) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == 3, "TCNNEncoding::forward: expected [B,3]");
  if (otype_ == "SphericalHarmonics") return sh_encoding(x, sh_degree_);
  return GridFwd::apply(x, params_.view({-1, n_feat_}), cfg_pack({n_levels_, n_feat_, log2_hashmap_, base_res_, per_level_scale_}));
}

torch::Tensor TCNNEncoding::forward_stencil(const torch::Tensor &x, int64_t n_groups, double delta_unit) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == 3 && n_groups >= 0 && x.size(0) == 7 * n_groups,
              "TCNNEncoding::forward_stencil: expected [7 * n_groups, 3]");
  GridCfg c{n_levels_, n_feat_, log2_hashmap_, base_res_, per_level_scale_};
  c.stencil_n = n_groups;
  // the coarse levels at which the +-delta points usually share the base point's cell: scale_l * delta < 1
  for (int l = 0; l < n_levels_; ++l)
    if ((base_res_ * std::pow((double)per_level_scale_, l) - 1.0) * delta_unit < 1.0) ++c.merge_levels;
  return GridFwd::apply(x, params_.view({-1, n_feat_}), cfg_pack(c));
}

TCNNNetwork::TCNNNetwork(int n_input_dims, int n_output_dims, const nlohmann::json &config, const std::string &name)
    : name_(name) {
  TORCH_CHECK(config.value("otype", std::string("FullyFusedMLP")) == "FullyFusedMLP", "TCNNNetwork: only FullyFusedMLP");
  TORCH_CHECK(config.value("activation", std::string("ReLU")) == "ReLU" &&
                  config.value("output_activation", std::string("None")) == "None",
              "TCNNNetwork: only ReLU hidden / None output activation");
  const int h = config.value("n_neurons", 64), nh = config.value("n_hidden_layers", 3);
  dims_.push_back(n_input_dims);
  for (int i = 0; i < nh; ++i) dims_.push_back(h);
  dims_.push_back(n_output_dims);
  std::vector<Tensor> ws;
  for (size_t l = 0; l + 1 < dims_.size(); ++l) {
    const float bound = 1.0f / std::sqrt((float)dims_[l]);
    ws.push_back((torch::rand({(int64_t)dims_[l + 1] * dims_[l]}) * 2.0f - 1.0f) * bound);
  }
  params_ = torch::cat(ws);
  if (torch::cuda::is_available()) params_ = params_.to(torch::kCUDA);
  params_.set_requires_grad(true);
  if (config.value("bias", false)) {   // torch::nn::Linear's bias initialisation: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    std::vector<Tensor> bs;
    for (size_t l = 0; l + 1 < dims_.size(); ++l)
      bs.push_back((torch::rand({(int64_t)dims_[l + 1]}) * 2.0f - 1.0f) * (1.0f / std::sqrt((float)dims_[l])));
    biases_ = torch::cat(bs);
    if (torch::cuda::is_available()) biases_ = biases_.to(torch::kCUDA);
    biases_.set_requires_grad(true);
  }
}

torch::Tensor TCNNNetwork::forward(const torch::Tensor &x) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == dims_[0], "TCNNNetwork::forward: expected [B,", dims_[0], "]");
  return MlpFn::apply(x, params_, biases_.defined() ? biases_ : torch::empty({0}, params_.options()), std::vector<int64_t>(dims_.begin(), dims_.end()));
}
