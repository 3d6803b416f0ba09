// neural_gs.cpp — gsdf_model::NeuralGS: the reference's splat model (/root/reference/include/neural_gaussian/neural_gaussian.{h,cpp})
// on the drop-in operator layer: parameters and activations, render(), densification statistics, the refinement strategy
// (grow = duplicate / split, prune, opacity reset, LR decay) with the Adam-state surgery of
// include/optimizer/optimizer_utils/optimizer_utils.cpp, SDF-aided initialisation and the gs.ply checkpoint.
// Python mirror: gs_sdf_amd/neural_gs.py; the two are compared in tests/test_gpu_cpp_model.py.
#include <cmath>
#include <fstream>
#include <sstream>

#include "gsdf_extras/gsdf_extras.h"
#include "gsdf_model/gsdf_model.h"
#include "gsplat_cpp/fully_fused_projection.h"
#include "gsplat_cpp/rasterize_to_pixels.h"
#include "gsplat_cpp/rendering.h"
#include "spatial.h"
#include "util.h"

using torch::Tensor;
namespace F = torch::nn::functional;

namespace gsdf_model {

namespace {
const char *const kParamNames[6] = {"offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest"};

inline int num_sh_bases(int degree) {
  TORCH_CHECK(degree >= 0 && degree <= 4, "spherical harmonic degree must be in [0, 4]");
  return (degree + 1) * (degree + 1);
}

Tensor normalize_rows(const Tensor &t, double eps = 1e-12) { return F::normalize(t, F::NormalizeFuncOptions().dim(-1).eps(eps)); }

// utils::normalized_quat_to_rotmat (include/utils/utils.cpp:538-558), (w,x,y,z) -> [n,3,3]
Tensor normalized_quat_to_rotmat(const Tensor &q) {
  auto c = q.unbind(-1);
  const Tensor &w = c[0], &x = c[1], &y = c[2], &z = c[3];
  return torch::stack({1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                       2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}, -1).reshape({-1, 3, 3});
}

// utils::rotation_6d_to_matrix (include/utils/utils.cpp:693-719): Gram-Schmidt, columns b1, b2, b1 x b2
Tensor rotation_6d_to_matrix(const Tensor &r6) {
  Tensor a1 = r6.slice(-1, 0, 3), a2 = r6.slice(-1, 3, 6);
  Tensor b1 = normalize_rows(a1);
  Tensor b2 = normalize_rows(a2 - (b1 * a2).sum(-1, true) * b1);
  return torch::stack({b1, b2, torch::cross(b1, b2, -1)}, -1);
}

// gauss_utils.hpp:32-47
Tensor random_quat_tensor(int64_t n) {
  Tensor u = torch::rand({n}), v = torch::rand({n}), w = torch::rand({n});
  return torch::stack({torch::sqrt(1 - u) * torch::sin(2 * M_PI * v), torch::sqrt(1 - u) * torch::cos(2 * M_PI * v),
                       torch::sqrt(u) * torch::sin(2 * M_PI * w), torch::sqrt(u) * torch::cos(2 * M_PI * w)}, -1);
}

Tensor nz(const Tensor &mask) { return mask.nonzero().reshape({-1}); }
}  // namespace

// ---- neural_gaussian.cpp:19-127 ----------------------------------------------------------------------------------------------
std::map<std::string, Tensor> init_gs_with_sdf(LocalMap &local_map, const Tensor &xyzs, float mesh_res, bool init_opa, int64_t batch_size) {
  torch::NoGradGuard ng;
  const int64_t n = xyzs.size(0);
  Tensor grad = torch::empty({n, 3}, xyzs.options()), curv = torch::empty({n, 3}, xyzs.options()), quat = torch::empty({n, 4}, xyzs.options());
  Tensor opa = init_opa ? torch::empty({n}, xyzs.options()) : Tensor();
  for (int64_t s = 0; s < n; s += batch_size) {
    const int64_t e = std::min(n, s + batch_size);
    Tensor x = xyzs.slice(0, s, e);
    auto gh = local_map.get_gradient(x, mesh_res, Tensor(), true, true);
    // normal = SDF gradient, in-plane axis = the diagonal Hessian's direction; the splat's third axis is the normal
    Tensor rot = rotation_6d_to_matrix(torch::cat({normalize_rows(gh[0]), normalize_rows(gh[1])}, -1));
    rot = torch::stack({rot.select(-1, 1), rot.select(-1, 2), rot.select(-1, 0)}, -1);
    using torch::indexing::Slice;
    Tensor trace = rot.index({Slice(), 0, 0}) + rot.index({Slice(), 1, 1}) + rot.index({Slice(), 2, 2});
    Tensor angle = torch::acos((trace.unsqueeze(-1) - 1.0f) * 0.5f);
    Tensor axis = torch::stack({rot.index({Slice(), 2, 1}) - rot.index({Slice(), 1, 2}), rot.index({Slice(), 0, 2}) - rot.index({Slice(), 2, 0}),
                                rot.index({Slice(), 1, 0}) - rot.index({Slice(), 0, 1})}, -1) / (2.0f * torch::sin(angle));
    axis = normalize_rows(axis);
    Tensor q = torch::cat({torch::cos(angle * 0.5f), torch::sin(angle * 0.5f) * axis}, -1).nan_to_num();   // angle = 0 -> NaN axis
    grad.slice(0, s, e).copy_(gh[0]);
    curv.slice(0, s, e).copy_(gh[1]);
    quat.slice(0, s, e).copy_(q);
    if (init_opa) {
      auto si = local_map.get_sdf(x);
      opa.slice(0, s, e).copy_(torch::exp(-si[0].square() * si[1]).squeeze(-1));
    }
  }
  std::map<std::string, Tensor> out = {{"quaternion", quat}, {"grad", grad}, {"curv_dom", curv}};
  if (init_opa) out["opacity"] = opa;
  return out;
}

// ---- neural_gaussian.cpp:129-271 ---------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, std::map<std::string, Tensor>>
rasterization_2dgs_sdf(const Tensor &means, const Tensor &quats, const Tensor &scales, const Tensor &opacities, const Tensor &colors,
                       const Tensor &viewmats, const Tensor &Ks, int width, int height, const std::string &render_mode, float near_plane,
                       float far_plane, float radius_clip, at::optional<int> sh_degree, bool absgrad, bool center_reg) {
  TORCH_CHECK(render_mode == "RGB" || render_mode == "D" || render_mode == "ED" || render_mode == "RGB+D" || render_mode == "RGB+ED",
              "Invalid render_mode");
  const int64_t N = means.size(0), C = viewmats.size(0);
  TORCH_CHECK(opacities.dim() == 1 && opacities.size(0) == N, "Invalid opacities shape");
  // stochastic SDF samples on the splat's disc unless k_center_reg (:259-265): the mode is this call's argument, no ambient state
  auto proj = gsplat_cpp::fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, width, height, near_plane, far_plane, radius_clip, true, false,
                                                      /*stochastic_samples=*/!center_reg);
  const Tensor &camera_ids = std::get<0>(proj), &gaussian_ids = std::get<1>(proj), &radii = std::get<2>(proj), &means2d = std::get<3>(proj);
  const Tensor &depths = std::get<4>(proj), &ray_transforms = std::get<5>(proj), &normals = std::get<6>(proj);
  Tensor samples = std::get<7>(proj), samples_weights = std::get<8>(proj);
  if (center_reg) {
    samples = means.index_select(0, gaussian_ids);
    samples_weights = torch::ones_like(samples_weights);
  }
  Tensor pt_opacities = opacities.index_select(0, gaussian_ids);
  Tensor pt_colors = gsplat_cpp::get_view_colors(viewmats, means, radii, colors, camera_ids, gaussian_ids, sh_degree);
  auto enc = gsplat_cpp::tile_encode(width, height, 16, means2d, radii, depths, true, C, camera_ids, gaussian_ids);
  Tensor means2d_absgrad = torch::zeros_like(means2d).requires_grad_(absgrad);
  Tensor densify = torch::zeros_like(means2d).requires_grad_(true);
  auto rast = rasterize_to_pixels_2dgs(means2d, ray_transforms, pt_colors, pt_opacities, normals, densify, width, height, 16, std::get<2>(enc),
                                       std::get<1>(enc), at::nullopt, at::nullopt, true, means2d_absgrad, false);
  // :229-240 in one pass: expected depth, cat(colours, depth), normals to world space (+ the colour / depth slices render() takes)
  const bool ed = render_mode == "ED" || render_mode == "RGB+ED";
  auto post = gsdf_extras::render_post(std::get<0>(rast), std::get<1>(rast), std::get<2>(rast), std::get<3>(rast), viewmats, ed);
  std::map<std::string, Tensor> meta;
  if (absgrad) meta["absgrad"] = means2d_absgrad;
  meta["color"] = post[2];
  meta["depth"] = post[3];
  meta["render_normal"] = post[1];
  meta["render_median"] = std::get<5>(rast);
  meta["normal"] = normals;
  meta["gaussian_ids"] = gaussian_ids;
  meta["radii"] = radii;
  meta["gradient_2dgs"] = densify;
  meta["width"] = torch::tensor({width});
  meta["height"] = torch::tensor({height});
  meta["n_cameras"] = torch::tensor({(int)C});
  meta["samples"] = samples;
  meta["samples_weights"] = samples_weights;
  meta["samples_opacities"] = pt_opacities;
  meta["visibilities"] = std::get<6>(rast);
  return std::make_tuple(post[0], std::get<2>(rast), meta);
}

// ---- construction ------------------------------------------------------------------------------------------------------------
NeuralGS::NeuralGS(const LocalMap::Ptr &local_map_ptr, const Tensor &points, int num_train_data, float spatial_scale, bool sdf_enable,
                   const GSConfig &cfg)
    : local_map_ptr_(local_map_ptr), cfg_(cfg), original_spatial_scale_(spatial_scale), sdf_enable_(sdf_enable), num_train_data_(num_train_data) {
  torch::NoGradGuard ng;
  spatial_scale_ = std::min(original_spatial_scale_, 2.f);   // a larger position LR degenerates the training (:281-282)
  pause_refine_after_reset = cfg.pause_refine_after_reset;
  auto device = points.device();
  anchors_ = points;
  scaling_ = torch::log(torch::sqrt(torch::clamp_min(distCUDA2(points), 1e-6f))).unsqueeze(-1).repeat({1, 3}).to(device);   // :312-316
  const int64_t n = anchors_.size(0);
  if (sdf_enable_ && cfg.geo_init) {   // :320-326
    TORCH_CHECK(local_map_ptr_ != nullptr, "NeuralGS: geo_init needs a LocalMap");
    auto r = init_gs_with_sdf(*local_map_ptr_, anchors_, 0.5f * local_map_ptr_->cfg_.leaf_size, true, cfg.vis_batch_pt_num);
    quaternion_ = r["quaternion"];
    opacity_ = r["opacity"];
  } else {
    quaternion_ = random_quat_tensor(n).to(device);
    opacity_ = torch::logit(0.1f * torch::ones({n}, device));
  }
  offsets_ = torch::zeros({n, 3}, device);
  features_dc_ = torch::rand({n, 1, 3}, device);
  features_rest_ = torch::zeros({n, num_sh_bases(cfg.sh_degree) - 1, 3}, device);
  Tensor is_nan = anchors_.isnan().any(-1) | scaling_.isnan().any(-1) | quaternion_.isnan().any(-1) | opacity_.isnan();   // :408-423
  if (is_nan.any().item<bool>()) {
    Tensor valid = nz(~is_nan);
    anchors_ = anchors_.index_select(0, valid).contiguous();
    offsets_ = offsets_.index_select(0, valid).contiguous();
    scaling_ = scaling_.index_select(0, valid).contiguous();
    quaternion_ = quaternion_.index_select(0, valid).contiguous();
    opacity_ = opacity_.index_select(0, valid).contiguous();
    features_dc_ = features_dc_.index_select(0, valid).contiguous();
    features_rest_ = features_rest_.index_select(0, valid).contiguous();
  }
  build_param_groups();
}

NeuralGS::NeuralGS(const LocalMap::Ptr &local_map_ptr, const Tensor &anchors, const Tensor &scaling, const Tensor &quaternion, const Tensor &opacity,
                   const Tensor &features_dc, const Tensor &features_rest, int num_train_data, float spatial_scale, const GSConfig &cfg)
    : local_map_ptr_(local_map_ptr), cfg_(cfg), original_spatial_scale_(spatial_scale), num_train_data_(num_train_data) {
  torch::NoGradGuard ng;
  spatial_scale_ = std::min(original_spatial_scale_, 2.f);
  pause_refine_after_reset = cfg.pause_refine_after_reset;
  auto own = [](const Tensor &t) { return t.detach().clone().contiguous(); };
  anchors_ = own(anchors);
  offsets_ = torch::zeros_like(anchors_);
  scaling_ = own(scaling);
  quaternion_ = own(quaternion);
  opacity_ = own(opacity.reshape({-1}));
  features_dc_ = own(features_dc);
  features_rest_ = own(features_rest);
  build_param_groups();
}

std::vector<Tensor *> NeuralGS::params() { return {&offsets_, &scaling_, &quaternion_, &opacity_, &features_dc_, &features_rest_}; }

void NeuralGS::build_param_groups() {   // :426-453
  anchors_ = register_parameter("anchors", anchors_, false);
  auto ps = params();
  for (int k = 0; k < 6; ++k) *ps[k] = register_parameter(kParamNames[k], *ps[k], true);
  const double lrs[6] = {1.6e-4 * spatial_scale_, 0.005, 0.001, 0.05, 0.0025, 0.0025 / 20.0};
  optimizer_params_groups_.clear();
  for (int k = 0; k < 6; ++k) {
    auto opt = std::make_unique<torch::optim::AdamOptions>(lrs[k]);
    opt->eps(1e-15);
    optimizer_params_groups_.emplace_back(std::vector<Tensor>{*ps[k]}, std::move(opt));
  }
}

// ---- activations (:463-478) --------------------------------------------------------------------------------------------------
Tensor NeuralGS::get_xyz() { return (anchors_ + offsets_).view({-1, 3}); }
Tensor NeuralGS::get_scale() { return torch::exp(scaling_).view({-1, 3}); }
Tensor NeuralGS::get_opacity(bool) { return torch::sigmoid(opacity_); }   // k_render_mode (opaque evaluation renders) is a viewer option

// ---- render (:495-566) ---------------------------------------------------------------------------------------------------------
std::map<std::string, Tensor> NeuralGS::render(const Tensor &pose_cam2world, const Cameras &camera, bool training, int bck_color) {
  std::lock_guard<std::mutex> guard(render_mutex_);
  (void)training;
  auto device = anchors_.device();
  Tensor intr = torch::tensor({{camera.fx, 0.0f, camera.cx}, {0.0f, camera.fy, camera.cy}, {0.0f, 0.0f, 1.0f}}).to(device).unsqueeze(0);
  Tensor rot = pose_cam2world.slice(1, 0, 3).to(device), pos = pose_cam2world.slice(1, 3, 4).to(device);
  Tensor w2c = torch::cat({torch::cat({rot.t(), -rot.t().matmul(pos)}, 1), torch::tensor({{0.0f, 0.0f, 0.0f, 1.0f}}).to(device)}, 0).unsqueeze(0);
  // generate_gaussian (:480-492): exp / sigmoid / anchors + offsets in one launch
  auto act = gsdf_extras::splat_activations(anchors_, offsets_, scaling_, opacity_);
  Tensor sh = torch::cat({features_dc_, features_rest_}, 1);
  auto [renders, alphas, info] = rasterization_2dgs_sdf(act[0], quaternion_.view({-1, 4}), act[1], act[2], sh, w2c.contiguous(), intr, camera.width,
                                                        camera.height, "RGB+ED", cfg_.near, cfg_.far, 0.0f, sh_degree_to_use_, cfg_.use_absgrad,
                                                        cfg_.center_reg);
  Tensor color = info["color"][0], depth = info["depth"][0];   // = renders.slice(-1, 0, 3)[0], renders.slice(-1, 3, 4)[0]
  info.erase("color");
  info.erase("depth");
  std::map<std::string, Tensor> out;
  if (bck_color == 2) out["color"] = color + (1.0f - alphas[0]) * torch::rand({camera.height, camera.width, 3}, color.options());
  else if (bck_color == 1) out["color"] = color + (1.0f - alphas[0]);
  else out["color"] = color;
  out["depth"] = depth;
  out["alpha"] = alphas;
  out["xyz"] = act[0];
  out.insert(info.begin(), info.end());
  if (out[key_for_gradient].requires_grad()) out[key_for_gradient].retain_grad();
  return out;
}

// ---- densification statistics (:626-688) -----------------------------------------------------------------------------------
void NeuralGS::update_state(std::map<std::string, Tensor> &info) {
  const Tensor &src = cfg_.use_absgrad ? info.at("absgrad") : info.at(key_for_gradient);
  TORCH_CHECK(src.grad().defined(), "NeuralGS::update_state: ", key_for_gradient, " has no gradient (call after backward())");
  gsdf_extras::update_state(state, src.grad(), info.at("gaussian_ids"), info.at("visibilities"), info.at("radii"), anchors_.size(0),
                            info.at("n_cameras").item<int>(), info.at("width").item<int>(), info.at("height").item<int>(),
                            cfg_.refine_scale2d_stop_iter > 0);
}

void NeuralGS::zero_state() {
  state["grad2d"].zero_();
  state["count"].zero_();
  if (cfg_.refine_scale2d_stop_iter > 0) state["radii"].zero_();
}

// ---- Adam-state surgery (optimizer_utils.cpp:5-165) ----------------------------------------------------------------------------
// One routine for prune / cat / prune+cat: rows keep_idx of the old tensor followed by the extension rows.  Where the reference
// returns early — a parameter without Adam state (no step yet) is left UNCHANGED there, which desynchronises it from the other
// five — the tensor is still replaced here.
void NeuralGS::apply_rows(const std::shared_ptr<torch::optim::Adam> &p, const Tensor &keep_idx, const std::vector<Tensor> &ext) {
  auto ps = params();
  for (int k = 0; k < 6; ++k) {
    Tensor old = ps[k]->detach();
    auto rows = [&](const Tensor &t) { return keep_idx.defined() ? t.index_select(0, keep_idx) : t; };
    const bool grow = !ext.empty() && ext[k].defined();
    Tensor fresh = (grow ? torch::cat({rows(old), ext[k]}, 0) : rows(old)).contiguous().set_requires_grad(true);
    if (p != nullptr) {
      auto &slot = p->param_groups().at(gs_param_start_idx + k).params().at(0);
      auto *old_impl = slot.unsafeGetTensorImpl();
      auto it = p->state().find(old_impl);
      if (it != p->state().end()) {
        auto st = std::make_unique<torch::optim::AdamParamState>(static_cast<torch::optim::AdamParamState &>(*it->second));
        p->state().erase(it);
        auto moments = [&](const Tensor &m) { return grow ? torch::cat({rows(m), torch::zeros_like(ext[k])}, 0) : rows(m); };
        st->exp_avg(moments(st->exp_avg()));
        st->exp_avg_sq(moments(st->exp_avg_sq()));
        p->state()[fresh.unsafeGetTensorImpl()] = std::move(st);
      }
      slot = fresh;
    }
    *ps[k] = fresh;
    parameters_[kParamNames[k]] = fresh;   // keep torch::save(module) / named_parameters() in step with the members
  }
}

// ---- grow (:690-827) -----------------------------------------------------------------------------------------------------------
std::pair<int, int> NeuralGS::grow_gs(int iter, const std::shared_ptr<torch::optim::Adam> &p) {
  Tensor grads = state.at("grad2d") / state.at("count").clamp_min(1);
  Tensor is_grad_high = grads > cfg_.grow_grad2d;
  Tensor is_small = std::get<0>(get_scale().detach().slice(-1, 0, 2).max(-1)) <= cfg_.grow_scale3d * spatial_scale_;
  Tensor is_dupli = is_grad_high & is_small, is_split = is_grad_high & ~is_small;
  if (iter < cfg_.refine_scale2d_stop_iter) is_split = is_split | (state.at("radii") > cfg_.grow_scale2d);
  const int n_dupli = duplicate(p, is_dupli);
  is_split = torch::cat({is_split, torch::zeros({n_dupli}, is_split.options())});   // the copies are not split in the same round
  const int n_split = split(p, is_split);
  return {n_dupli, n_split};
}

int NeuralGS::duplicate(const std::shared_ptr<torch::optim::Adam> &p, const Tensor &is_dupli) {
  const int n = is_dupli.sum().item<int>();
  if (n > 0) {
    Tensor idx = nz(is_dupli);
    anchors_ = torch::cat({anchors_, anchors_.index_select(0, idx)}, 0);
    std::vector<Tensor> ext;
    for (Tensor *t : params()) ext.push_back(t->detach().index_select(0, idx));
    apply_rows(p, Tensor(), ext);
    for (auto &kv : state) kv.second = torch::cat({kv.second, kv.second.index_select(0, idx)});
    parameters_["anchors"] = anchors_;
  }
  return n;
}

int NeuralGS::split(const std::shared_ptr<torch::optim::Adam> &p, const Tensor &is_split) {
  const int n = is_split.sum().item<int>();
  if (n > 0) {
    const int K = 2;
    Tensor sel = nz(is_split), rest = nz(~is_split);
    Tensor scales = get_scale().detach().index_select(0, sel);
    scales = torch::cat({scales.slice(-1, 0, 2), torch::zeros({n, 1}, scales.options())}, 1);   // a disc: no extent along the normal
    Tensor sample_scales = scales.unsqueeze(0) * torch::randn({K, n, 3}, scales.options());
    Tensor rot = normalized_quat_to_rotmat(normalize_rows(quaternion_.detach().index_select(0, sel)));
    // einsum("nij,nj,bnj->bni")
    Tensor split_offsets = (torch::matmul(rot.unsqueeze(0), (scales.unsqueeze(0) * sample_scales).unsqueeze(-1)).squeeze(-1) +
                            offsets_.detach().index_select(0, sel).unsqueeze(0)).reshape({-1, 3});
    std::vector<Tensor> ext = {split_offsets,
                               torch::log(scales / 1.6f).repeat({K, 1}),
                               quaternion_.detach().index_select(0, sel).repeat({K, 1}),
                               opacity_.detach().index_select(0, sel).repeat({K}),
                               features_dc_.detach().index_select(0, sel).repeat({K, 1, 1}),
                               features_rest_.detach().index_select(0, sel).repeat({K, 1, 1})};
    anchors_ = torch::cat({anchors_.index_select(0, rest), anchors_.index_select(0, sel).repeat({K, 1})}, 0);
    apply_rows(p, rest, ext);
    for (auto &kv : state) kv.second = torch::cat({kv.second.index_select(0, rest), kv.second.index_select(0, sel).repeat({K})});
    parameters_["anchors"] = anchors_;
  }
  return n;
}

// ---- prune (:829-916) -----------------------------------------------------------------------------------------------------------
int NeuralGS::prune_gs(const std::shared_ptr<torch::optim::Adam> &p, const Tensor &is_prune) {
  const int n = is_prune.sum().item<int>();
  if (n > 0) {
    Tensor valid = nz(~is_prune);
    anchors_ = anchors_.index_select(0, valid);
    apply_rows(p, valid, {});
    for (auto &kv : state) kv.second = kv.second.index_select(0, valid);
    parameters_["anchors"] = anchors_;
  }
  return n;
}

int NeuralGS::prune_gs(int iter, const std::shared_ptr<torch::optim::Adam> &p, bool prune_opa_only) {
  Tensor is_prune = get_opacity().detach() < cfg_.prune_opa;
  Tensor scale = get_scale().detach().slice(-1, 0, 2);
  is_prune = is_prune | (std::get<0>(scale.min(-1)) < 1e-4);
  if (!prune_opa_only && iter > cfg_.reset_every)
    is_prune = is_prune | (std::get<0>(scale.max(-1)) > cfg_.prune_scale3d * original_spatial_scale_);
  return prune_gs(p, is_prune);
}

int NeuralGS::prune_invisible_gs(int iter, const std::shared_ptr<torch::optim::Adam> &p) {
  if (iter > 0 && iter % num_train_data_ == 0 && state.count("vis")) {
    Tensor is_prune = state["vis"] < 1e-4;
    state["vis"].zero_();
    return prune_gs(p, is_prune);
  }
  return 0;
}

int NeuralGS::prune_nan_gs(int, const std::shared_ptr<torch::optim::Adam> &p) {
  // one launch (count + mask) instead of 3 x (isnan, any) + 2 x or; the mask is only gathered from when the count is non-zero
  Tensor mask;
  Tensor count = gsdf_extras::nan_rows(offsets_.detach(), scaling_.detach(), quaternion_.detach(), &mask);
  if (count.item<int>() == 0) return 0;
  return prune_gs(p, mask);
}

void NeuralGS::reset_opacity(const std::shared_ptr<torch::optim::Adam> &p) {   // :918-926 -> reset_optimizer: new values, fresh moments
  const double cap = std::log(cfg_.prune_opa * 2.0 / (1.0 - cfg_.prune_opa * 2.0));
  Tensor fresh = opacity_.detach().clamp_max(cap).set_requires_grad(true);
  if (p != nullptr) {
    auto &slot = p->param_groups().at(gs_param_start_idx + 3).params().at(0);
    auto it = p->state().find(slot.unsafeGetTensorImpl());
    if (it != p->state().end()) {
      auto st = std::make_unique<torch::optim::AdamParamState>(static_cast<torch::optim::AdamParamState &>(*it->second));
      p->state().erase(it);
      st->exp_avg(torch::zeros_like(fresh));
      st->exp_avg_sq(torch::zeros_like(fresh));
      p->state()[fresh.unsafeGetTensorImpl()] = std::move(st);
    }
    slot = fresh;
  }
  opacity_ = fresh;
  parameters_["opacity"] = fresh;
}

// ---- per-iteration callback (:568-624) -----------------------------------------------------------------------------------------
void NeuralGS::train_callback(int iter, int total_iter, const std::shared_ptr<torch::optim::Adam> &p, std::map<std::string, Tensor> &info) {
  std::lock_guard<std::mutex> guard(render_mutex_);
  torch::NoGradGuard ng;
  const int refine_stop_iter = total_iter / 2;
  if (!info.empty()) {
    if (iter >= refine_stop_iter) return;
    update_state(info);
    prune_nan_gs(iter, p);
    prune_invisible_gs(iter, p);
    sh_degree_to_use_ = std::min(cfg_.sh_degree, iter / cfg_.sh_degree_interval);   // one more SH band every sh_degree_interval
    if (iter > 0) {
      if (iter > cfg_.refine_start_iter && iter % cfg_.refine_every == 0 && (iter % cfg_.reset_every) >= pause_refine_after_reset) {
        grow_gs(iter, p);
        prune_gs(iter, p);
        zero_state();
      }
      if (iter % cfg_.reset_every == 0) reset_opacity(p);
    }
  }
  if (p != nullptr) {   // exponential decay of the position LR; the SDF groups (the ones before gs_param_start_idx) follow it, capped
    const float r = (float)iter / (float)total_iter;
    const float lr0 = 1.6e-4f * spatial_scale_, lr1 = 1.6e-6f * spatial_scale_;
    const float lr = std::exp(std::log(lr0) * (1 - r) + std::log(lr1) * r);
    p->param_groups().at(gs_param_start_idx).options().set_lr(lr);
    const float sdf_lr = cfg_.detach_sdf_grad ? 0.0f : std::min(lr, cfg_.lr_end);
    for (int i = 0; i < gs_param_start_idx; ++i) p->param_groups()[i].options().set_lr(sdf_lr);
  }
}

void NeuralGS::freeze_structure() {
  for (Tensor *t : {&offsets_, &scaling_, &quaternion_}) t->requires_grad_(false);
}
void NeuralGS::unfreeze_structure() {
  for (Tensor *t : {&offsets_, &scaling_, &quaternion_}) t->requires_grad_(true);
}

// ---- gs.ply (:928-1188): 3DGS-compatible binary PLY ---------------------------------------------------------------------------
void NeuralGS::export_gs_to_ply(const std::filesystem::path &output_path) {
  torch::NoGradGuard ng;
  auto host = [](const Tensor &t) { return t.detach().to(torch::kCPU).to(torch::kFloat32).contiguous(); };
  const int64_t n = anchors_.size(0);
  Tensor xyz = host(get_xyz());
  Tensor f_dc = host(features_dc_.detach().transpose(1, 2).flatten(1));
  Tensor f_rest = cfg_.sh_degree > 0 ? host(features_rest_.detach().transpose(1, 2).flatten(1)) : torch::zeros({n, 0});
  Tensor opa = host(opacity_).reshape({n, 1});
  Tensor scale = host(scaling_).clone();
  scale.select(1, 2).fill_(std::log(1e-6));   // a flat third axis for 3DGS viewers
  Tensor rot = host(quaternion_);
  std::vector<std::string> names = {"x", "y", "z"};
  for (int64_t i = 0; i < f_dc.size(1); ++i) names.push_back("f_dc_" + std::to_string(i));
  for (int64_t i = 0; i < f_rest.size(1); ++i) names.push_back("f_rest_" + std::to_string(i));
  for (const char *k : {"opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"}) names.emplace_back(k);
  Tensor data = torch::cat({xyz, f_dc, f_rest, opa, scale, rot}, 1).contiguous();
  std::ofstream f(output_path, std::ios::binary);
  TORCH_CHECK(f.good(), "export_gs_to_ply: cannot open ", output_path.string());
  f << "ply\nformat binary_little_endian 1.0\nelement vertex " << n << "\n";
  for (auto &k : names) f << "property float " << k << "\n";
  f << "end_header\n";
  f.write(reinterpret_cast<const char *>(data.data_ptr<float>()), (std::streamsize)(data.numel() * sizeof(float)));
}

void NeuralGS::load_ply_to_gs(const std::filesystem::path &input_path) {
  torch::NoGradGuard ng;
  std::ifstream f(input_path, std::ios::binary);
  TORCH_CHECK(f.good(), "load_ply_to_gs: cannot open ", input_path.string());
  std::vector<std::string> names;
  int64_t n = 0;
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ss(line);
    std::string a, b, c;
    ss >> a >> b >> c;
    if (a == "element" && b == "vertex") n = std::stoll(c);
    else if (a == "property") { TORCH_CHECK(b == "float", "load_ply_to_gs: only float properties are supported"); names.push_back(c); }
    else if (a == "format") { TORCH_CHECK(b == "binary_little_endian", "load_ply_to_gs: only binary_little_endian PLY is supported"); }
    else if (a == "end_header") break;
  }
  const int64_t cols = (int64_t)names.size();
  Tensor data = torch::empty({n, cols}, torch::kFloat32);
  f.read(reinterpret_cast<char *>(data.data_ptr<float>()), (std::streamsize)(n * cols * sizeof(float)));
  TORCH_CHECK(f.gcount() == (std::streamsize)(n * cols * sizeof(float)), "load_ply_to_gs: truncated file");
  std::map<std::string, int64_t> col;
  for (int64_t i = 0; i < cols; ++i) col[names[i]] = i;
  auto device = anchors_.defined() ? anchors_.device() : torch::Device(torch::kCUDA);
  auto take = [&](const std::vector<std::string> &keys) {
    std::vector<int64_t> idx;
    for (auto &k : keys) { TORCH_CHECK(col.count(k), "load_ply_to_gs: missing property ", k); idx.push_back(col[k]); }
    return data.index_select(1, torch::tensor(idx)).contiguous().to(device);
  };
  int64_t n_dc = 0, n_rest = 0;
  for (auto &k : names) { n_dc += k.rfind("f_dc_", 0) == 0; n_rest += k.rfind("f_rest_", 0) == 0; }
  auto numbered = [](const char *prefix, int64_t cnt) { std::vector<std::string> v; for (int64_t i = 0; i < cnt; ++i) v.push_back(prefix + std::to_string(i)); return v; };
  anchors_ = take({"x", "y", "z"});   // anchors = positions, offsets = 0 (:1160-1170)
  offsets_ = torch::zeros_like(anchors_);
  scaling_ = take({"scale_0", "scale_1", "scale_2"});
  quaternion_ = take({"rot_0", "rot_1", "rot_2", "rot_3"});
  opacity_ = take({"opacity"}).reshape({-1});
  features_dc_ = take(numbered("f_dc_", n_dc)).reshape({n, 3, -1}).transpose(1, 2).contiguous();
  features_rest_ = n_rest ? take(numbered("f_rest_", n_rest)).reshape({n, 3, -1}).transpose(1, 2).contiguous() : torch::zeros({n, 0, 3}, anchors_.options());
  cfg_.sh_degree = (int)std::lround(std::sqrt(1.0 + (double)(n_rest / 3))) - 1;
  sh_degree_to_use_ = cfg_.sh_degree;
  // (re)register: the loader may run on a constructed model
  for (const char *k : {"anchors", "offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest"})
    if (parameters_.contains(k)) parameters_.erase(k);
  build_param_groups();
  state.clear();
}

}  // namespace gsdf_model
