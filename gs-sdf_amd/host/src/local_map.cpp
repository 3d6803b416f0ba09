// local_map.cpp — gsdf_model::LocalMap: SubMap + EncodingMap + LocalMap of the reference
// (/root/reference/include/neural_net/{sub_map,encoding_map,local_map}.{h,cpp}) on the drop-in operator layer.
// Python mirror: gs_sdf_amd/sdf.py (class LocalMap); the two are compared in tests/test_gpu_cpp_model.py.
#include <cmath>

#include "gsdf_extras/gsdf_extras.h"
#include "gsdf_model/gsdf_model.h"
#include "kaolin_wisp_cpp/spc_ops/spc_ops.h"
#include "util.h"

using torch::Tensor;

namespace gsdf_model {

// ---- DepthSamples (ray_utils.h:10-142) ---------------------------------------------------------------------------------
namespace {
template <typename F>
DepthSamples map_fields(const DepthSamples &s, F f) {
  DepthSamples o;
  auto g = [&](const Tensor &t) { return t.defined() ? f(t) : Tensor(); };
  o.origin = g(s.origin); o.direction = g(s.direction); o.depth = g(s.depth); o.xyz = g(s.xyz); o.ray_sdf = g(s.ray_sdf); o.ridx = g(s.ridx);
  o.pred_sdf = g(s.pred_sdf); o.pred_isigma = g(s.pred_isigma);
  return o;
}
}  // namespace

int64_t DepthSamples::size(int dim) const {
  for (const Tensor *t : {&xyz, &origin, &direction, &depth, &ray_sdf, &ridx})
    if (t->defined()) return t->size(dim);
  return 0;
}
DepthSamples DepthSamples::index_select(const Tensor &idx) const {
  return map_fields(*this, [&](const Tensor &t) { return t.index_select(0, idx); });
}
DepthSamples DepthSamples::cat(const DepthSamples &other) const {
  DepthSamples o;
  auto c = [](const Tensor &a, const Tensor &b) { return (a.defined() && b.defined()) ? torch::cat({a, b}, 0) : Tensor(); };
  o.origin = c(origin, other.origin); o.direction = c(direction, other.direction); o.depth = c(depth, other.depth); o.xyz = c(xyz, other.xyz);
  o.ray_sdf = c(ray_sdf, other.ray_sdf); o.ridx = c(ridx, other.ridx); o.pred_sdf = c(pred_sdf, other.pred_sdf);
  o.pred_isigma = c(pred_isigma, other.pred_isigma);
  return o;
}

// ---- MapConfig: params.cpp:243-256 ---------------------------------------------------------------------------------------
int MapConfig::octree_level() const { return (int)std::ceil(std::log2((inner_map_size + 2.f * leaf_size) / leaf_size)); }
float MapConfig::map_size() const { return std::pow(2.f, (float)octree_level()) * leaf_size; }

// ---- construction: sub_map.cpp:7-20, encoding_map.cpp:6-30, local_map.cpp:16-56 --------------------------------------------
LocalMap::LocalMap(const Tensor &pos_W_M, const MapConfig &cfg) : cfg_(cfg) {
  TORCH_CHECK(torch::cuda::is_available(), "gsdf_model::LocalMap: no HIP device (this path has no CPU fallback)");
  pos_W_M_ = pos_W_M.view({1, 3}).to(torch::kCUDA).to(torch::kFloat32);
  const float half = 0.5f * cfg.inner_map_size;
  xyz_min_W_ = pos_W_M_ - half;
  xyz_max_W_ = pos_W_M_ + half;
  map_size_inv_ = 1.0f / cfg.map_size();
  // encoder: the tcnn "Grid"/"Hash" configuration of encoding_map.cpp:15-23
  nlohmann::json enc_cfg = {{"otype", "Grid"}, {"type", "Hash"}, {"n_levels", cfg.n_levels}, {"n_features_per_level", cfg.n_features_per_level},
                            {"log2_hashmap_size", cfg.log2_hashmap_size}, {"base_resolution", cfg.base_resolution},
                            {"per_level_scale", cfg.per_level_scale}, {"interpolation", "Linear"}};
  p_encoder_tcnn_ = std::make_shared<TCNNEncoding>(3, enc_cfg, "encoder_local_map");
  p_encoder_tcnn_->params_ = register_parameter(p_encoder_tcnn_->name_, p_encoder_tcnn_->params_, true);   // local_map.cpp:73-75
  const int feat = (int)p_encoder_tcnn_->get_out_dim();
  // decoder: implementation 0 = Linear(feat, h) + ReLU, geo_num_layer x [Linear(h, h) + ReLU], Linear(h, 2) with biases
  // (local_map.cpp:29-42), on the fused kernels; implementation 1 = tcnn FullyFusedMLP with geo_num_layer hidden layers (:44-55)
  const bool torch_topology = cfg.decoder_implementation == 0;
  nlohmann::json net_cfg = {{"otype", "FullyFusedMLP"}, {"activation", "ReLU"}, {"output_activation", "None"}, {"n_neurons", cfg.hidden_dim},
                            {"n_hidden_layers", cfg.geo_num_layer + (torch_topology ? 1 : 0)}, {"bias", torch_topology}};
  p_decoder_tcnn_ = std::make_shared<TCNNNetwork>(feat, 2, net_cfg, "decoder");
  p_decoder_tcnn_->params_ = register_parameter(p_decoder_tcnn_->name_, p_decoder_tcnn_->params_, true);
  if (p_decoder_tcnn_->biases_.defined()) p_decoder_tcnn_->biases_ = register_parameter("decoder_bias", p_decoder_tcnn_->biases_, true);
}

// ---- checkpoint layout (neural_mapping.cpp:1331-1378; reference module layout local_map.cpp:29-55,73-75) ------------------------
namespace {
// the reference's decoder_implementation 0 module, with the flat fused-kernel parameters copied in (or out)
torch::nn::Sequential sequential_of(const TCNNNetwork &net, bool copy_in) {
  torch::nn::Sequential seq;
  const auto &d = net.dims_;
  int64_t wo = 0, bo = 0;
  torch::NoGradGuard ng;
  for (size_t l = 0; l + 1 < d.size(); ++l) {
    torch::nn::Linear lin(torch::nn::LinearOptions(d[l], d[l + 1]).bias(true));
    if (copy_in) {   // on the parameters' device, as torch::save of the reference's module keeps its tensors (torch::load restores them there)
      lin->to(net.params_.device());
      lin->weight.copy_(net.params_.detach().slice(0, wo, wo + (int64_t)d[l] * d[l + 1]).view({d[l + 1], d[l]}));
      if (net.biases_.defined()) lin->bias.copy_(net.biases_.detach().slice(0, bo, bo + d[l + 1]));
      else lin->bias.zero_();
    }
    wo += (int64_t)d[l] * d[l + 1]; bo += d[l + 1];
    seq->push_back(lin);
    if (l + 2 < d.size()) seq->push_back(torch::nn::ReLU(torch::nn::ReLUOptions(true)));
  }
  return seq;
}
}  // namespace

void LocalMap::save(torch::serialize::OutputArchive &archive) const {
  archive.write(p_encoder_tcnn_->name_, p_encoder_tcnn_->params_.detach());
  if (cfg_.decoder_implementation != 0) {
    archive.write(p_decoder_tcnn_->name_, p_decoder_tcnn_->params_.detach());
    return;
  }
  torch::serialize::OutputArchive child(archive.compilation_unit());
  sequential_of(*p_decoder_tcnn_, true)->save(child);
  archive.write("decoder", child);
}

void LocalMap::load(torch::serialize::InputArchive &archive) {
  torch::NoGradGuard ng;
  // everything is read and validated BEFORE anything is copied: a refused checkpoint leaves the map untouched
  Tensor enc;
  TORCH_CHECK(archive.try_read(p_encoder_tcnn_->name_, enc), "LocalMap::load: no '", p_encoder_tcnn_->name_, "' parameter (not a LocalMap checkpoint)");
  TORCH_CHECK(enc.numel() == p_encoder_tcnn_->params_.numel(), "LocalMap::load: hash-grid table has ", enc.numel(), " entries, this map ",
              p_encoder_tcnn_->params_.numel());
  Tensor dec_w, dec_b;
  if (cfg_.decoder_implementation != 0) {
    // the flat FullyFusedMLP parameter, either unpadded (what save() writes) or in upstream tiny-cuda-nn's layout, whose last layer is padded
    // to a multiple of 16 output rows (what a checkpoint written by the reference's own tcnn build, or by checkpoint.py with
    // pad_tcnn_output, holds): the real rows come first, the padding is dropped
    TORCH_CHECK(archive.try_read(p_decoder_tcnn_->name_, dec_w), "LocalMap::load: no flat 'decoder' parameter (decoder_implementation 1)");
    const auto &d = p_decoder_tcnn_->dims_;
    const int64_t n_plain = p_decoder_tcnn_->params_.numel(), in_last = d[d.size() - 2], out_last = d.back();
    const int64_t rows_padded = (out_last + 15) / 16 * 16, n_padded = n_plain - out_last * in_last + rows_padded * in_last;
    TORCH_CHECK(dec_w.numel() == n_plain || dec_w.numel() == n_padded, "LocalMap::load: flat 'decoder' parameter has ", dec_w.numel(), " entries, expected ",
                n_plain, " (unpadded) or ", n_padded, " (tiny-cuda-nn's padded last layer)");
    if (dec_w.numel() != n_plain) {
      dec_w = dec_w.reshape({-1});
      const int64_t head = n_plain - out_last * in_last;
      dec_w = torch::cat({dec_w.slice(0, 0, head), dec_w.slice(0, head, head + rows_padded * in_last).view({rows_padded, in_last}).slice(0, 0, out_last).reshape({-1})});
    }
  } else {
    torch::serialize::InputArchive child;
    TORCH_CHECK(archive.try_read("decoder", child), "LocalMap::load: no 'decoder' submodule (decoder_implementation 0 expects the Sequential layout)");
    auto seq = sequential_of(*p_decoder_tcnn_, false);
    seq->load(child);   // throws on a missing / mis-shaped decoder.<2k>.weight / .bias
    std::vector<Tensor> ws, bs;
    for (auto &mod : seq->children())
      if (auto *lin = mod->as<torch::nn::Linear>()) { ws.push_back(lin->weight.reshape({-1})); bs.push_back(lin->bias.reshape({-1})); }
    dec_w = torch::cat(ws).detach(); dec_b = torch::cat(bs).detach();
    TORCH_CHECK(dec_w.numel() == p_decoder_tcnn_->params_.numel(), "LocalMap::load: decoder topology differs");
  }
  p_encoder_tcnn_->params_.copy_(enc.reshape(p_encoder_tcnn_->params_.sizes()).to(p_encoder_tcnn_->params_.device()));
  p_decoder_tcnn_->params_.copy_(dec_w.reshape(p_decoder_tcnn_->params_.sizes()).to(p_decoder_tcnn_->params_.device()));
  if (p_decoder_tcnn_->biases_.defined() && dec_b.defined()) p_decoder_tcnn_->biases_.copy_(dec_b.to(p_decoder_tcnn_->biases_.device()));
}

// ---- SubMap ----------------------------------------------------------------------------------------------------------------
const LocalMap::HostFrame &LocalMap::host_frame() {
  const Tensor *src[3] = {&pos_W_M_, &xyz_min_W_, &xyz_max_W_};
  for (int k = 0; k < 3; ++k)
    if (host_frame_ptr_[k] != src[k]->data_ptr() || host_frame_ver_[k] != src[k]->_version()) host_frame_valid_ = false;
  if (!host_frame_valid_) {
    for (int k = 0; k < 3; ++k) { host_frame_ptr_[k] = src[k]->data_ptr(); host_frame_ver_[k] = src[k]->_version(); }
    // the bounds get_inrange_mask compares against (padding 0), computed by the same libtorch operations, read back once
    Tensor v = torch::cat({pos_W_M_.reshape({-1}), (xyz_min_W_ + 0.f + 1e-6).reshape({-1}), (xyz_max_W_ - 0.f - 1e-6).reshape({-1})}).to(torch::kCPU);
    const float *p = v.data_ptr<float>();
    for (int k = 0; k < 3; ++k) { host_frame_.pos[k] = p[k]; host_frame_.lo[k] = p[3 + k]; host_frame_.hi[k] = p[6 + k]; }
    host_frame_valid_ = true;
  }
  return host_frame_;
}
Tensor LocalMap::xyz_to_m1p1_pts(const Tensor &xyz) const { return (xyz - pos_W_M_) * 2 * map_size_inv_; }
Tensor LocalMap::m1p1_pts_to_xyz(const Tensor &pts) const { return scale_from_m1p1(pts) + pos_W_M_; }
Tensor LocalMap::scale_from_m1p1(const Tensor &t) const { return t * 0.5 * (1.0 / map_size_inv_); }
Tensor LocalMap::xyz_to_zp1_pts(const Tensor &xyz) const { return 0.5f * xyz_to_m1p1_pts(xyz) + 0.5f; }

void LocalMap::update_octree_as(const Tensor &xyz, bool is_prior) {
  torch::NoGradGuard ng;
  const int level = cfg_.octree_level();
  Tensor qpts = spc_ops::quantize_points(xyz_to_m1p1_pts(xyz.to(pos_W_M_.device())), level);
  qpts = std::get<0>(torch::unique_dim(qpts.contiguous(), 0));
  if (!is_prior) qpts = spc_ops::points_to_neighbors(qpts).view({-1, 3}).clamp(0, (1 << level) - 1);
  p_acc_strcut_occ_ = std::shared_ptr<OctreeAS>(from_quantized_points(qpts, level));
}

Tensor LocalMap::get_inrange_mask(const Tensor &xyz, float padding) const {
  return ((xyz < (xyz_max_W_ - padding - 1e-6).to(xyz.device())) & (xyz > (xyz_min_W_ + padding + 1e-6).to(xyz.device()))).all(1);
}

void LocalMap::get_intersect_point(const Tensor &points, const Tensor &rays, Tensor &z_nears, Tensor &z_fars, Tensor &mask_intersect,
                                   float padding) const {
  // parallel rays are nudged off zero so that the slab test stays finite; they fail the near < far check below
  Tensor tmp = torch::where(rays == 0, rays + 1e-6, rays);
  Tensor a = (xyz_min_W_ + padding - points) / tmp, b = (xyz_max_W_ - padding - points) / tmp;
  z_nears = std::get<0>(torch::max(torch::minimum(a, b), 1));
  z_fars = std::get<0>(torch::min(torch::maximum(a, b), 1));
  mask_intersect = z_nears < z_fars;
}

Tensor LocalMap::get_valid_mask(const Tensor &xyz, int level) {
  TORCH_CHECK(p_acc_strcut_occ_ != nullptr, "LocalMap::get_valid_mask: update_octree_as has not been called");
  return p_acc_strcut_occ_->query(xyz_to_m1p1_pts(xyz), level).pidx > -1;
}

// ---- LocalMap --------------------------------------------------------------------------------------------------------------
void LocalMap::freeze_net() {
  p_encoder_tcnn_->params_.requires_grad_(false);
  p_decoder_tcnn_->params_.requires_grad_(false);
  if (p_decoder_tcnn_->biases_.defined()) p_decoder_tcnn_->biases_.requires_grad_(false);
}
void LocalMap::unfreeze_net() {
  p_encoder_tcnn_->params_.requires_grad_(true);
  p_decoder_tcnn_->params_.requires_grad_(true);
  if (p_decoder_tcnn_->biases_.defined()) p_decoder_tcnn_->biases_.requires_grad_(true);
}

Tensor LocalMap::get_feat(const Tensor &xyz, int encoding_type, bool normalized) {
  (void)encoding_type;   // the reference has a single encoder type
  return p_encoder_tcnn_->forward(normalized ? xyz : xyz_to_zp1_pts(xyz));
}

std::vector<Tensor> LocalMap::get_sdf(const Tensor &xyz) {
  Tensor attr = p_decoder_tcnn_->forward(get_feat(xyz, 0, false));
  auto parts = torch::split(attr, {1, 1}, -1);   // sdf, raw isigma
  return {parts[0], 1 + torch::softplus(parts[1], /*beta=*/100) * (1.0 / cfg_.bce_sigma)};
}

std::vector<Tensor> LocalMap::get_gradient(const Tensor &xyz_, float delta, Tensor sdf, bool hessian, bool numerical_grad) {
  if (numerical_grad) {
    // central differences on the 6-point stencil (+x,-x,+y,-y,+z,-z), [6,n,3] -> one get_sdf over 6n points
    Tensor offs = torch::tensor({{{delta, 0.f, 0.f}}, {{-delta, 0.f, 0.f}}, {{0.f, delta, 0.f}}, {{0.f, -delta, 0.f}}, {{0.f, 0.f, delta}}, {{0.f, 0.f, -delta}}},
                                xyz_.options().requires_grad(false));
    Tensor ps = get_sdf((xyz_.unsqueeze(0) + offs).view({-1, 3}))[0].view({6, xyz_.size(0), 1});
    const double inv = 1.0 / delta;
    Tensor grad = 0.5 * inv * torch::cat({ps[0] - ps[1], ps[2] - ps[3], ps[4] - ps[5]}, 1);
    if (!hessian) return {grad};
    if (!sdf.defined()) sdf = get_sdf(xyz_)[0];
    Tensor hess = inv * inv * (torch::cat({ps[0] + ps[1], ps[2] + ps[3], ps[4] + ps[5]}, 1) - 2 * sdf);
    return {grad, hess};
  }
  // analytic: torch::autograd::grad(create_graph = true) through the fused decoder and the hash grid (both second order)
  const bool grad_mode = torch::GradMode::is_enabled();
  torch::GradMode::set_enabled(true);
  Tensor xyz = xyz_;
  if (!xyz.requires_grad() || !sdf.defined()) {
    xyz.requires_grad_(true);
    sdf = get_sdf(xyz)[0];
  }
  Tensor g = torch::autograd::grad({sdf}, {xyz}, {torch::ones_like(sdf)}, true, true)[0];
  std::vector<Tensor> out = {g};
  if (hessian) out.push_back(torch::autograd::grad({g}, {xyz}, {torch::ones_like(g)}, true, true)[0]);
  torch::GradMode::set_enabled(grad_mode);
  return out;
}

namespace {
// utils::sample_free_pts (include/utils/utils.cpp:366-393): stratified, jittered samples in [0, depth) of every ray
DepthSamples sample_free_pts(const DepthSamples &rays, int sample_num) {
  const int64_t n = rays.origin.size(0);
  Tensor steps = torch::arange(sample_num, rays.origin.options()).unsqueeze(0).repeat({n, 1});
  steps = (steps + torch::rand({n, sample_num}, rays.origin.options())) / (double)sample_num;
  Tensor ridx = torch::arange(n, rays.origin.options().dtype(torch::kInt64)).unsqueeze(1).repeat({1, sample_num}).reshape({-1});
  DepthSamples out = rays.index_select(ridx);
  Tensor d = out.depth * steps.reshape({-1, 1});
  out.xyz = out.origin + out.direction * d;
  out.ray_sdf = out.depth - d;
  out.depth = d;
  out.ridx = rays.ridx.defined() ? rays.ridx.index_select(0, ridx) : ridx;
  return out;
}
}  // namespace

DepthSamples LocalMap::sample(const DepthSamples &samples_in, int voxel_sample_num, bool sample_free) {
  DepthSamples samples;
  if (voxel_sample_num < 1) {
    samples.xyz = samples_in.xyz;
    samples.ray_sdf = torch::zeros_like(samples_in.depth);
    samples.direction = samples_in.direction;
    samples.ridx = samples_in.ridx;
    return samples;
  }
  TORCH_CHECK(p_acc_strcut_occ_ != nullptr, "LocalMap::sample: update_octree_as has not been called");
  // one sample per occupied voxel a ray crosses (octree ray march in the [-1,1] frame), SDF target = ray depth - sample depth
  auto rm = p_acc_strcut_occ_->raymarch(xyz_to_m1p1_pts(samples_in.origin).contiguous(), samples_in.direction.contiguous(), "voxel", voxel_sample_num);
  samples = samples_in.index_select(rm.ridx);
  samples.ridx = rm.ridx;
  samples.xyz = m1p1_pts_to_xyz(rm.samples);
  Tensor d = scale_from_m1p1(rm.depth_samples);
  samples.ray_sdf = samples.depth - d;
  samples.depth = d;
  if (sample_free) samples = samples.cat(sample_free_pts(samples_in, cfg_.free_sample_num));
  // keep what lies in front of the surface
  return samples.index_select((samples.ray_sdf > 0).reshape({-1}).nonzero().reshape({-1}));
}

DepthSamples LocalMap::filter_sample(const DepthSamples &samples) {
  TORCH_CHECK(p_acc_strcut_occ_ != nullptr, "LocalMap::filter_sample: update_octree_as has not been called");
  return samples.index_select((p_acc_strcut_occ_->query(xyz_to_m1p1_pts(samples.xyz)).pidx > -1).nonzero().reshape({-1}));
}

DepthSamples sample_rays_composed(LocalMap &local_map, DepthSamples rays, float sample_std, float truncated_dis, int surface_sample_num,
                         bool sample_free) {
  const int64_t n = rays.size(0);
  auto dev = rays.origin.device();
  rays.ridx = torch::arange(n, torch::TensorOptions().dtype(torch::kInt64).device(dev));
  rays.ray_sdf = torch::zeros({n, 1}, rays.origin.options());
  DepthSamples pts = local_map.sample(rays, 1, sample_free);
  {   // utils::sample_surface_pts: k samples per ray at signed distance N(0, std) from the end point, along the ray
    DepthSamples s;
    Tensor sdf = torch::randn({n, surface_sample_num, 1}, rays.origin.options()) * sample_std;
    auto rep = [&](const Tensor &t) { return t.unsqueeze(1).repeat({1, surface_sample_num, 1}).view({-1, t.size(1)}); };
    s.origin = rep(rays.origin);
    s.xyz = (rays.xyz.unsqueeze(1) - rays.direction.unsqueeze(1) * sdf).view({-1, 3});
    s.direction = rep(rays.direction);
    s.ridx = rays.ridx.unsqueeze(1).repeat({1, surface_sample_num}).view({-1});
    s.depth = rep(rays.depth);
    s.ray_sdf = sdf.view({-1, 1});
    pts = pts.cat(s);
  }
  pts.ray_sdf = torch::where(pts.ray_sdf.abs() > truncated_dis, pts.ray_sdf.sign() * truncated_dis, pts.ray_sdf);
  pts = pts.cat(rays);
  return pts.index_select(local_map.get_inrange_mask(pts.xyz).nonzero().reshape({-1}));
}


// The same batch in three kernels (count per ray and segment, one-workgroup scan, fill: csrc/occupancy.hip ray_sampler_kernel) + the two
// random draws, which stay torch's: rand [n, F] first, randn [n, S, 1] second — the generator calls of the composition above (and of the
// reference's NeuralSLAM::sample), so both paths place every sample identically.  Rows come in the composition's order.
DepthSamples sample_rays(LocalMap &local_map, DepthSamples rays, float sample_std, float truncated_dis, int surface_sample_num,
                         bool sample_free) {
  using namespace gsdf_host;
  static const bool composed = [] { const char *e = getenv("GSDF_FUSED_SAMPLER"); return e && e[0] == '0'; }();
  const int F = sample_free ? local_map.cfg_.free_sample_num : 0, S = surface_sample_num;
  if (composed || F > 64 || S > 64 || F < 0 || S < 0) return sample_rays_composed(local_map, rays, sample_std, truncated_dis, surface_sample_num, sample_free);
  torch::NoGradGuard ng;
  TORCH_CHECK(local_map.p_acc_strcut_occ_ != nullptr, "sample_rays: update_octree_as has not been called");
  Tensor origin = f32c(rays.origin.detach(), "rays.origin"), direction = f32c(rays.direction.detach(), "rays.direction");
  Tensor depth = f32c(rays.depth.detach(), "rays.depth"), end = f32c(rays.xyz.detach(), "rays.xyz");
  const int64_t n = origin.size(0);
  TORCH_CHECK(origin.dim() == 2 && origin.size(1) == 3 && direction.sizes() == origin.sizes() && end.sizes() == origin.sizes() && depth.numel() == n,
              "sample_rays: expected origin / direction / xyz [n,3] and depth [n,1]");
  Tensor rf = F > 0 ? torch::rand({n, F}, origin.options()) : Tensor();
  Tensor rs = torch::randn({n, S, 1}, origin.options());
  gsdf_ray_sampler_args a{};
  a.level = local_map.p_acc_strcut_occ_->max_level_;
  a.n_rays = n;
  a.origin = fp(origin); a.direction = fp(direction); a.depth = fp(depth); a.end_xyz = fp(end);
  a.grid = local_map.p_acc_strcut_occ_->grid_.data_ptr();
  a.rand_free = F > 0 ? fp(rf) : nullptr; a.randn_surf = fp(rs);
  a.free_sample_num = F; a.surface_sample_num = S;
  const LocalMap::HostFrame &hf = local_map.host_frame();
  for (int k = 0; k < 3; ++k) { a.map_origin[k] = hf.pos[k]; a.range_lo[k] = hf.lo[k]; a.range_hi[k] = hf.hi[k]; }
  a.map_size_inv = local_map.map_size_inv_;
  a.map_half = (float)(1.0 / (double)local_map.map_size_inv_);      // scale_from_m1p1: t * 0.5 * (1.0 / map_size_inv_), the scalar rounded to fp32
  a.sample_std = sample_std; a.truncated_dis = truncated_dis;
  Tensor counts = empty_like_opts(origin, {4 * n}, torch::kInt32), incl = empty_like_opts(origin, {4 * n}, torch::kInt64);
  const int64_t B = count_via_host_word(origin, [&](int64_t *total) {
    check(gsdf_ray_sampler_count(&a, n ? counts.data_ptr<int32_t>() : nullptr, n ? incl.data_ptr<int64_t>() : nullptr, total, cur_stream()), "sample_rays (count)");
  }, n * (int64_t)(3 * ((int64_t)1 << a.level) + F + S + 1));      // <= 3 cells per slab of the major axis + the other three segments
  DepthSamples out;
  out.xyz = empty_like_opts(origin, {B, 3}, torch::kFloat32); out.ray_sdf = empty_like_opts(origin, {B, 1}, torch::kFloat32);
  out.ridx = empty_like_opts(origin, {B}, torch::kInt64); out.origin = empty_like_opts(origin, {B, 3}, torch::kFloat32);
  out.direction = empty_like_opts(origin, {B, 3}, torch::kFloat32); out.depth = empty_like_opts(origin, {B, 1}, torch::kFloat32);
  if (B > 0)
    check(gsdf_ray_sampler_fill(&a, counts.data_ptr<int32_t>(), incl.data_ptr<int64_t>(), fpm(out.xyz), fpm(out.ray_sdf), out.ridx.data_ptr<int64_t>(),
                                fpm(out.origin), fpm(out.direction), fpm(out.depth), cur_stream()), "sample_rays (fill)");
  return out;
}

}  // namespace gsdf_model
