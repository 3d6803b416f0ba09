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

// (v_feat, x, table) -> (v_x, v_table): the encoding's backward as a differentiable op (grad of grad)
struct GridBwd : public torch::autograd::Function<GridBwd> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &v_feat_, const Tensor &x, const Tensor &table,
                             std::vector<int64_t> cfgv, bool want_table) {
    const GridCfg c = cfg_of(c10::IValue(cfgv));
    Tensor v_feat = f32c(v_feat_, "v_feat");
    const int64_t B = x.size(0);
    Tensor v_x = torch::empty_like(x);
    Tensor v_table = want_table ? torch::zeros_like(table) : Tensor();
    // large batches: the table gradient without global atomics (gsdf_hashgrid_bwd_binned), d/dx from the plain kernel
    const size_t binned = (want_table && B >= 24576) ? gsdf_hashgrid_bwd_binned_ws_bytes(B, c.L, c.F, c.H, c.R, c.S) : 0;
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
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &v_feat = s[0], &x = s[1], &table = s[2];
    const GridCfg c = cfg_of(ctx->saved_data["cfg"]);
    if (!g[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    Tensor vv = f32c(g[0], "vv_x");
    Tensor g_vfeat = ctx->needs_input_grad(0) ? torch::empty_like(v_feat) : Tensor();
    Tensor g_x = ctx->needs_input_grad(1) ? torch::empty_like(x) : Tensor();
    Tensor g_table = ctx->needs_input_grad(2) ? torch::zeros_like(table) : Tensor();
    check(gsdf_hashgrid_bwd_bwd(x.size(0), c.L, c.F, c.H, c.R, c.S, fp(x), fp(table), fp(v_feat), fp(vv), fpm(g_vfeat),
                                fpm(g_table), fpm(g_x), cur_stream()),
          "TCNNEncoding double backward");
    return {g_vfeat, g_x, g_table, Tensor(), Tensor()};
  }
};

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

struct MlpFn : public torch::autograd::Function<MlpFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &x_, const Tensor &w_, std::vector<int64_t> dims64) {
    Tensor x = f32c(x_, "x"), w = f32c(w_, "params");
    std::vector<int> dims(dims64.begin(), dims64.end());
    const int nl = (int)dims.size() - 1;
    const int64_t B = x.size(0);
    Tensor out = empty_like_opts(x, {B, dims.back()}, torch::kFloat32);
    const bool need = x.requires_grad() || w.requires_grad();
    Tensor acts = need ? empty_like_opts(x, {(int64_t)gsdf_mlp_acts_floats(B, nl)}, torch::kFloat32) : Tensor();
    check(gsdf_mlp_fwd(B, nl, dims.data(), fp(w), nullptr, fp(x), fpm(out), fpm(acts), cur_stream()), "TCNNNetwork forward");
    ctx->save_for_backward({x, w, acts});
    ctx->saved_data["dims"] = dims64;
    return out;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    // first order only, like tcnn's FullyFusedMLP: the reference forces numerical_grad with this decoder
    // (params.cpp:396-399).  Under create_graph the result would silently be treated as constant: refuse instead.
    TORCH_CHECK(!(torch::GradMode::is_enabled() && g[0].defined() && g[0].requires_grad()),
                "TCNNNetwork: double backward through the fused MLP is not implemented (use numerical_grad, as the reference "
                "does with decoder_implementation 1)");
    auto s = ctx->get_saved_variables();
    auto dims64 = ctx->saved_data["dims"].toIntVector();
    std::vector<int> dims(dims64.begin(), dims64.end());
    const int nl = (int)dims.size() - 1;
    const int64_t B = s[0].size(0);
    Tensor v_out = f32c(g[0], "grad");
    Tensor v_in = ctx->needs_input_grad(0) ? torch::empty_like(s[0]) : Tensor();
    Tensor v_w = ctx->needs_input_grad(1) ? torch::zeros_like(s[1]) : Tensor();
    // 0 bytes when both gradients are requested and the one-pass backward covers the topology
    Tensor ws = empty_like_opts(s[0], {(int64_t)gsdf_mlp_bwd_ws_bytes_for(B, nl, dims.data(), v_w.defined() ? 1 : 0)}, torch::kUInt8);
    check(gsdf_mlp_bwd(B, nl, dims.data(), fp(s[1]), nullptr, fp(s[0]), fp(s[2]), fp(v_out), fpm(v_in), fpm(v_w), nullptr,
                       ws.data_ptr(), cur_stream()),
          "TCNNNetwork backward");
    return {v_in, v_w, Tensor()};
  }
};
}  // namespace

void TCNNEncoding::init_encoding(int n_input_dims, const nlohmann::json &config, const std::string &name) {
  name_ = name;
  n_input_dims_ = n_input_dims;
  otype_ = config.value("otype", std::string("Grid"));
  if (otype_ == "SphericalHarmonics") {  // declared by the reference (encodings.h:6-27) but never evaluated in training
    sh_degree_ = config.value("degree", 4);
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
  TORCH_CHECK(otype_ != "SphericalHarmonics", "TCNNEncoding: the SphericalHarmonics encoding is declared but not implemented "
              "(unused by the reference's training path)");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == 3, "TCNNEncoding::forward: expected [B,3]");
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
}

torch::Tensor TCNNNetwork::forward(const torch::Tensor &x) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == dims_[0], "TCNNNetwork::forward: expected [B,", dims_[0], "]");
  return MlpFn::apply(x, params_, std::vector<int64_t>(dims_.begin(), dims_.end()));
}
