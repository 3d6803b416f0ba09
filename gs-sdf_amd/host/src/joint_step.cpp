// joint_step.cpp — gsdf_extras::JointIteration: the loop body of NeuralSLAM::gs_train
// (/root/reference/include/neural_mapping/neural_mapping.cpp:400-486) on the drop-in operators and the fused pieces of
// gsdf_extras, entirely in C++/libtorch: what the reference's node runs after the edits of INTEGRATION.md section 5.
// Mirrors the step bench.py times (single-stream form): per-ray SDF batch, render + 0.8 L1 + 0.2 D-SSIM, GS<->SDF coupling with
// the eikonal regulariser, backward, update_state, fused Adam.  The parameters live in two flat buffers (splats; SDF table +
// decoder), every trainable tensor is a view whose .grad is a view of the matching flat gradient buffer.
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cmath>
#include <limits>

#include "gsdf_extras/gsdf_extras.h"
#include "gsplat_cpp/fully_fused_projection.h"
#include "gsplat_cpp/rasterize_to_pixels.h"
#include "gsplat_cpp/rendering.h"
#include "tcnn_binding/tcnn_binding.h"
#include "stream_gate.h"
#include "util.h"

using namespace gsdf_host;
using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

namespace {
// the stencil hash-grid forward as a resident grid while the other leg shares the chip (JointConfig::hashgrid_resident; the hint is per host thread)
struct ResidentGridHint {
  int before;
  explicit ResidentGridHint(int wgs_per_cu) : before(gsdf_hashgrid_fwd_stencil_resident(wgs_per_cu)) {}
  ~ResidentGridHint() { gsdf_hashgrid_fwd_stencil_resident(before); }
  static int of(const gsdf_extras::JointConfig &cfg) {
    // (the analytic configuration only: in the numerical one the stencil forward runs beside other kernels of its own leg and the resident
    //  grid costs 6 % of the step, 120.7 -> 113.8 it/s)
    return cfg.two_streams && cfg.analytic ? cfg.hashgrid_resident : -1;
  }
};
// the gradient buffers are zeroed by the Adam launch that consumes them (gsdf_adam_step_zero_grad); measured equal to separate fills within noise
// in round 4, kept because it removes the fill launches from both legs
constexpr bool fused_zero() { return true; }


// neural_gaussian.cpp:229-240 in one pass: expected depth, cat(colours, depth), normals to world space (+ the colour / depth slices)
struct RenderPost : public torch::autograd::Function<RenderPost> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &rc_, const Tensor &rd_, const Tensor &ra_, const Tensor &rn_,
                             const Tensor &viewmats_, bool expected_depth) {
    Tensor rc = f32c(rc_, "render_colors"), rd = f32c(rd_, "render_depths"), ra = f32c(ra_, "render_alphas"), rn = f32c(rn_, "render_normals");
    Tensor vm = f32c(viewmats_, "viewmats");
    auto shp = ra.sizes().vec();
    shp.back() = 4;
    Tensor renders = empty_like_opts(ra, shp, torch::kFloat32), nw = torch::empty_like(rn), c3 = torch::empty_like(rc), d1 = torch::empty_like(rd);
    check(gsdf_render_post_fwd(ra.numel(), expected_depth ? 1 : 0, fp(vm), fp(rc), fp(rd), fp(ra), fp(rn), fpm(renders), fpm(nw), fpm(c3), fpm(d1),
                               cur_stream()), "render_post_fwd");
    ctx->save_for_backward({rd, ra, vm});
    ctx->saved_data["ed"] = expected_depth;
    return {renders, nw, c3, d1};
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &rd = s[0], &ra = s[1], &vm = s[2];
    auto shp = ra.sizes().vec();
    shp.back() = 3;
    Tensor v_rc = empty_like_opts(ra, shp, torch::kFloat32), v_rn = empty_like_opts(ra, shp, torch::kFloat32);
    Tensor v_rd = torch::empty_like(ra), v_ra = torch::empty_like(ra);
    auto opt = [](const Tensor &t) { return t.defined() ? t.contiguous() : Tensor(); };
    Tensor g0 = opt(g[0]), g1 = opt(g[1]), g2 = opt(g[2]), g3 = opt(g[3]);
    check(gsdf_render_post_bwd(ra.numel(), ctx->saved_data["ed"].toBool() ? 1 : 0, fp(vm), fp(rd), fp(ra), fp(g0), fp(g1), fp(g2), fp(g3),
                               fpm(v_rc), fpm(v_rd), fpm(v_ra), fpm(v_rn), cur_stream()), "render_post_bwd");
    return {v_rc, v_rd, v_ra, v_rn, Tensor(), Tensor()};
  }
};

// scalar 0 whose backward delivers fixed upstream gradients to the given tensors (op-level gradients of the bench step)
struct InjectGrads : public torch::autograd::Function<InjectGrads> {
  static Tensor forward(AutogradContext *ctx, const Tensor &a, const Tensor &ga, const Tensor &b, const Tensor &gb, const Tensor &c, const Tensor &gc,
                        const Tensor &d, const Tensor &gd) {
    ctx->save_for_backward({ga, gb, gc, gd});
    return torch::zeros({}, a.options());
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    return {s[0] * g[0], Tensor(), s[1] * g[0], Tensor(), s[2] * g[0], Tensor(), s[3] * g[0], Tensor()};
  }
};

// identity whose backward first makes the current stream wait for an event: the gradient that enters here was produced on
// another stream (Python mirror: trainer.join_grad / GradGate)
struct JoinGrad : public torch::autograd::Function<JoinGrad> {
  static Tensor forward(AutogradContext *ctx, const Tensor &x, int64_t gate_handle) {
    ctx->saved_data["gate"] = gate_handle;
    return x.view_as(x);
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto *gate = reinterpret_cast<gsdf_extras::StreamGate *>(ctx->saved_data["gate"].toInt());
    if (gate != nullptr && gate->armed) {
      gate->event.block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
      gate->armed = false;
    }
    return {g[0], Tensor()};
  }
};

}  // namespace

namespace gsdf_extras {

std::vector<Tensor> render_post(const Tensor &render_colors, const Tensor &render_depths, const Tensor &render_alphas, const Tensor &render_normals,
                                const Tensor &viewmats, bool expected_depth) {
  return RenderPost::apply(render_colors, render_depths, render_alphas, render_normals, viewmats, expected_depth);
}

struct JointStreams {
  // (a high-priority queue for the SDF leg: rounds 3-5 it cost 1-5 % of the step; round 6, with the SDF leg's chain of kernels being the step, it was worth
  //  0.5 % of the headline step — and cost 12 % of the 1000-step run with refinement in the same process (217 against 247 it/s, reproducibly): not used)
  c10::hip::HIPStreamMasqueradingAsCUDA side = c10::hip::getStreamFromPoolMasqueradingAsCUDA(false);
  at::cuda::CUDAEvent fwd_done, entry, side_done;
  StreamGate gate;
  bool side_pending = false;
  std::unique_ptr<gsdf_host::HostWords> counts;   // step_direct: M, I, n_gs_sdf as host-visible words (no copy, no synchronisation)
};

JointIteration::~JointIteration() = default;

JointIteration::JointIteration(const Tensor &anchors, const std::vector<Tensor> &fields, std::shared_ptr<::TCNNEncoding> enc,
                               std::shared_ptr<::TCNNNetwork> dec, const std::vector<float> &map_origin, double map_size, double bce_sigma,
                               int occ_level, const JointConfig &cfg)
    : cfg_(cfg), enc_(std::move(enc)), dec_(std::move(dec)), origin_(map_origin), map_size_inv_(1.0 / map_size),
      bce_isigma_(1.0 / bce_sigma), occ_level_(occ_level), adam_(0.9, 0.999, 1e-15), adam_sdf_(0.9, 0.999, 1e-15) {
  TORCH_CHECK(fields.size() == 6, "JointIteration: fields = {offsets, scaling, quaternion, opacity, features_dc, features_rest}");
  TORCH_CHECK(origin_.size() == 3, "JointIteration: map_origin needs 3 entries");
  anchors_ = f32c(anchors.detach(), "anchors");
  const int64_t N = anchors_.size(0);
  // splat family: one flat buffer, the fields in the reference's group order (neural_gaussian.cpp:434-453)
  int64_t total = 0;
  for (const auto &f : fields) total += f.numel();
  flat_ = torch::empty({total}, anchors_.options());
  flat_grad_ = torch::zeros({total}, anchors_.options());
  const double lrs[6] = {cfg.lr_offsets, cfg.lr_scaling, cfg.lr_quaternion, cfg.lr_opacity, cfg.lr_features_dc, cfg.lr_features_rest};
  std::vector<int64_t> sizes;
  std::vector<double> seg_lrs;
  int64_t off = 0;
  for (size_t i = 0; i < 6; ++i) {
    const int64_t n = fields[i].numel();
    { torch::NoGradGuard ng; flat_.slice(0, off, off + n).copy_(fields[i].reshape({-1})); }
    field_cols_.push_back(N > 0 ? n / N : 0);
    if (n > 0) { sizes.push_back(n); seg_lrs.push_back(lrs[i]); }
    off += n;
  }
  field_shapes_.clear();
  for (const auto &f : fields) { auto sh = f.sizes().vec(); field_shapes_.push_back(std::vector<int64_t>(sh.begin() + 1, sh.end())); }
  bind_views(flat_, flat_grad_, N);
  n_rest_ = N > 0 ? fields[5].numel() / (3 * N) : 0;
  adam_.add_group(flat_, flat_grad_, sizes, seg_lrs);
  // SDF family: table, decoder weights (, decoder biases) in one flat buffer; the operators' params_ become views of it
  n_table_ = enc_->params_.numel(); n_dec_ = dec_->params_.numel(); n_bias_ = dec_->biases_.defined() ? dec_->biases_.numel() : 0;
  const int64_t nt = n_table_, nd = n_dec_, nb = n_bias_;
  sdf_flat_ = torch::empty({nt + nd + nb}, anchors_.options());
  sdf_flat_grad_ = torch::zeros({nt + nd + nb}, anchors_.options());
  {
    torch::NoGradGuard ng;
    sdf_flat_.slice(0, 0, nt).copy_(enc_->params_.detach().reshape({-1}));
    sdf_flat_.slice(0, nt, nt + nd).copy_(dec_->params_.detach().reshape({-1}));
    if (nb) sdf_flat_.slice(0, nt + nd, nt + nd + nb).copy_(dec_->biases_.detach().reshape({-1}));
  }
  auto bind = [&](Tensor &p, int64_t a, int64_t b) {
    p = sdf_flat_.slice(0, a, b);
    p.requires_grad_(true);
    p.mutable_grad() = sdf_flat_grad_.slice(0, a, b);
  };
  bind(enc_->params_, 0, nt);
  bind(dec_->params_, nt, nt + nd);
  if (nb) bind(dec_->biases_, nt + nd, nt + nd + nb);
  adam_sdf_.add_group(sdf_flat_, sdf_flat_grad_, {nt + nd + nb}, {cfg.lr_sdf});
  nan_total_ = torch::zeros({1}, anchors_.options().dtype(torch::kInt32));
  // occupancy structure of the map from the splat centres (SubMap::update_octree_as, sub_map.cpp:22-35)
  const size_t nbytes = gsdf_occ_bytes(occ_level_);
  TORCH_CHECK(nbytes > 0, "JointIteration: occupancy level outside [1,12]");
  occ_grid_ = torch::empty({(int64_t)(nbytes / 4)}, anchors_.options().dtype(torch::kInt32));
  Tensor m1p1 = ((anchors_ - torch::tensor(origin_, anchors_.options())) * (2.0 * map_size_inv_)).contiguous();
  check(gsdf_occ_build(occ_level_, N, fp(m1p1), 1, occ_grid_.data_ptr(), cur_stream()), "occ_build");
}

// the six trainable tensors as views of the flat buffer, their .grad views of the flat gradient buffer (constructor, and after every refinement)
void JointIteration::bind_views(const Tensor &flat, const Tensor &flat_grad, int64_t n) {
  flat_ = flat; flat_grad_ = flat_grad;
  views_.clear();
  int64_t off = 0;
  for (size_t i = 0; i < 6; ++i) {
    const int64_t cnt = n * field_cols_[i];
    std::vector<int64_t> shape = {n};
    shape.insert(shape.end(), field_shapes_[i].begin(), field_shapes_[i].end());
    Tensor v = flat_.slice(0, off, off + cnt).view(shape);
    v.requires_grad_(true);
    v.mutable_grad() = flat_grad_.slice(0, off, off + cnt).view(shape);
    views_.push_back(v);
    off += cnt;
  }
}

void JointIteration::sync() {
  if (streams_ && streams_->side_pending) {
    streams_->side_done.block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
    streams_->side_pending = false;
  }
}

std::map<std::string, int64_t> JointIteration::step(const Tensor &viewmat, const Tensor &K, const Tensor &target, const Tensor &ray_pts,
                                                    const Tensor &ray_sdf, const std::vector<Tensor> &upstream, bool update,
                                                    const std::vector<float> &cam_host) {
  if (direct_ok(viewmat)) return step_direct(viewmat, K, target, ray_pts, ray_sdf, update, cam_host);
  const int W = cfg_.width, H = cfg_.height;
  std::map<std::string, int64_t> sizes;
  // Two legs on two HIP streams (cfg.two_streams; the schedule of bench.py's overlapped step): the SDF network's work — ray batch,
  // coupling node, its Adam — on `side`, the splat leg on the caller's stream; they touch at the samples (forward) and at the
  // samples' gradient (backward, JoinGrad).
  using StreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
  const bool two = cfg_.two_streams;
  auto main_stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA();
  if (two && !streams_) streams_ = std::make_unique<JointStreams>();
  if (two) {
    // every step: whatever the caller produced on its stream (the ray batch, the constructor's copies, a parameter edit) is
    // ordered before the second stream's work
    streams_->entry.record(main_stream);
    streams_->entry.block(streams_->side);
    streams_->gate.armed = false;   // (a step whose coupling saw no samples leaves its gate unconsumed)
  }
  const int64_t N = anchors_.size(0);
  const int64_t nt = n_table_, nd = n_dec_, nb = n_bias_;
  Tensor tg = sdf_flat_grad_.slice(0, 0, nt), dg = sdf_flat_grad_.slice(0, nt, nt + nd);
  Tensor bg = nb ? sdf_flat_grad_.slice(0, nt + nd, nt + nd + nb) : Tensor();
  if (!cfg_.analytic) {
    // ---- per-ray SDF batch (:138-188), numerical configuration: BCE sdf_loss + eikonal on the numerical gradient, 7 n rows in one
    //      encoder / decoder / loss launch (independent of the render: issued first)
    c10::optional<StreamGuard> sg;
    if (two) {
      sg.emplace(streams_->side);
      ray_pts.record_stream(streams_->side);
      ray_sdf.record_stream(streams_->side);
    }
    const int64_t n = ray_pts.size(0);
    Tensor q = query_points(ray_pts, origin_, map_size_inv_, true, cfg_.sdf_delta);
    Tensor attr = dec_->forward(enc_->forward_stencil(q, n, cfg_.sdf_delta * map_size_inv_));
    sdf_ray_loss(attr, ray_sdf, n, bce_isigma_, cfg_.sdf_delta, cfg_.eik_w).backward();
  }
  // ---- render (:195-300 -> neural_gaussian.cpp:129-271) + photometric loss
  auto act = splat_activations(anchors_, views_[0], views_[1], views_[3].reshape({N}));
  Tensor dc = views_[4].reshape({N, 1, 3});
  Tensor sh = n_rest_ == 0 ? dc : torch::cat({dc, views_[5].reshape({N, n_rest_, 3})}, 1);
  auto proj = gsplat_cpp::fully_fused_projection_2dgs(act[0], views_[2], act[1], viewmat, K, W, H, cfg_.near_plane, cfg_.far_plane, 0.f, true, false,
                                                      /*stochastic_samples=*/!cfg_.center_reg);
  const Tensor &camera_ids = std::get<0>(proj), &gaussian_ids = std::get<1>(proj), &radii = std::get<2>(proj), &means2d = std::get<3>(proj);
  const Tensor &depths = std::get<4>(proj), &ray_transforms = std::get<5>(proj), &normals = std::get<6>(proj);
  // k_center_reg: the splat centres are the SDF samples (weight 1); otherwise the projection's stochastic points on the discs
  Tensor samples = cfg_.center_reg ? act[0].index_select(0, gaussian_ids) : std::get<7>(proj);
  if (two) samples = JoinGrad::apply(samples, reinterpret_cast<int64_t>(&streams_->gate));   // created HERE: consumed late in the backward
  Tensor samples_weights = cfg_.center_reg ? torch::ones_like(std::get<8>(proj)) : std::get<8>(proj);
  Tensor pt_opac = act[2].index_select(0, gaussian_ids);
  Tensor colors = gsplat_cpp::get_view_colors(viewmat, act[0], radii, sh, camera_ids, gaussian_ids, cfg_.sh_degree);
  auto enc = gsplat_cpp::tile_encode(W, H, 16, means2d, radii, depths, true, viewmat.size(0), camera_ids, gaussian_ids);
  Tensor densify = torch::zeros_like(means2d).requires_grad_(true), absgrad = torch::zeros_like(means2d);
  auto rast = rasterize_to_pixels_2dgs(means2d, ray_transforms, colors, pt_opac, normals, densify, W, H, 16, std::get<2>(enc), std::get<1>(enc),
                                       at::nullopt, at::nullopt, true, absgrad, false);
  auto post = RenderPost::apply(std::get<0>(rast), std::get<1>(rast), std::get<2>(rast), std::get<3>(rast), viewmat, true);   // render_mode RGB+ED (neural_gaussian.cpp:229-234)
  const Tensor &render_normal = post[1], &color3 = post[2], &depth1 = post[3];
  const Tensor &alphas = std::get<2>(rast), &median = std::get<5>(rast), &vis = std::get<6>(rast);
  // the splat leg's own loss terms (everything but the GS <-> SDF coupling)
  auto splat_loss = [&]() -> Tensor {
    Tensor loss = l1_dssim_loss(color3[0], target, cfg_.rgb_w, cfg_.dssim_w);
    if (cfg_.reference_terms) {
      // render_normal_weight x depth->normal consistency (:243-266; depth_type 0: the expected depth) + isotropic_weight x isotropic
      // regulariser of the visible splats (:268-276), one launch each
      std::vector<float> intr, pose;
      if (cam_host.size() >= 16) {
        intr.assign(cam_host.begin(), cam_host.begin() + 4);
        pose.assign(cam_host.begin() + 4, cam_host.begin() + 16);
      } else {   // read the camera back (one device->host copy)
        Tensor Kc = K[0].detach().to(torch::kCPU), c2w = torch::linalg_inv(viewmat[0].detach().to(torch::kCPU).to(torch::kFloat64)).to(torch::kFloat32).contiguous();
        intr = {Kc[0][0].item<float>(), Kc[1][1].item<float>(), Kc[0][2].item<float>(), Kc[1][2].item<float>()};
        pose.assign(c2w.data_ptr<float>(), c2w.data_ptr<float>() + 12);
      }
      loss = loss + cfg_.normal_w * normal_consistency_loss(depth1[0], alphas[0], render_normal[0], intr, pose) +
             cfg_.isotropic_w * isotropic_loss(act[1], gaussian_ids);
    } else if (upstream.size() == 4) {
      loss = loss + InjectGrads::apply(depth1, upstream[0], alphas, upstream[1], render_normal, upstream[2], median, upstream[3]);
    }
    return loss;
  };
  // ---- GS <-> SDF coupling (:420-462) at the visible, occupancy-valid splats' samples
  // visibility > k_visible_thr, get_valid_mask(samples), samples_weights * visibilities, nonzero: three launches + one size read-back
  Tensor visd = f32c(vis.detach(), "visibilities");
  Tensor w_all = torch::empty({samples.size(0), 1}, samples.options().requires_grad(false));
  Tensor ids_all = torch::empty({samples.size(0)}, samples.options().dtype(torch::kInt64).requires_grad(false));
  Tensor n_ids = torch::empty({1}, ids_all.options());
  {
    Tensor sd = f32c(samples.detach(), "samples"), swd = f32c(samples_weights.detach(), "samples_weights");
    Tensor vws = torch::empty({(int64_t)gsdf_visible_set_ws_bytes(sd.size(0))}, samples.options().dtype(torch::kUInt8).requires_grad(false));
    check(gsdf_visible_set(occ_level_, -1, sd.size(0), fp(sd), origin_.data(), (float)map_size_inv_, occ_grid_.data_ptr(), fp(visd), fp(swd),
                           (float)cfg_.vis_thresh, fpm(w_all), ids_all.data_ptr<int64_t>(), n_ids.data_ptr<int64_t>(), vws.data_ptr(), cur_stream()),
          "visible_set");
  }
  Tensor ids = ids_all.narrow(0, 0, n_ids.item<int64_t>());
  const bool has = ids.numel() > 0;
  // the SDF work that depends on the render: numerical configuration = the coupling node; analytic (default) configuration = the
  // WHOLE SDF batch of the iteration (per-ray points + splat samples) in one node
  auto sdf_node = [&](const Tensor &smp, StreamGate *gate) -> Tensor {
    ResidentGridHint hint(ResidentGridHint::of(cfg_));
    if (cfg_.analytic)
      return joint_sdf_loss_analytic(ray_pts, ray_sdf, has ? smp : Tensor(), has ? ids : Tensor(), has ? w_all : Tensor(), *enc_, *dec_, origin_,
                                     map_size_inv_, bce_isigma_, cfg_.sdf_w, cfg_.gs_sdf_w, cfg_.sdf_delta, cfg_.eik_w, cfg_.align_w, tg, dg, bg, gate);
    return gs_sdf_coupling(smp, ids, w_all, *enc_, *dec_, origin_, map_size_inv_, cfg_.gs_sdf_w, cfg_.sdf_delta, cfg_.eik_w, tg, dg, gate);
  };
  const bool sdf_work = cfg_.analytic || has;
  // two streams, schedule A (default): SDF forward on `side` FIRST, then the splat leg's losses and its whole backward on the caller's
  // stream (beside the hash-grid forward: VALU-bound compositing backward next to a gather-bound kernel), then the SDF backward on
  // `side`, and last the samples' gradient through the few nodes between the samples and the splat parameters.  The splat leg's
  // stream no longer idles while the host issues the SDF leg, and the compositing backward is off the tail of the step.
  // (Schedule B, the round-2 order — SDF forward + backward, then one backward over {loss, samples} — is what the code below falls back to
  //  when there is no SDF work.)
  constexpr bool schedule_a = true;
  if (!two) {
    Tensor loss = splat_loss();
    if (sdf_work) loss = loss + sdf_node(samples, nullptr);
    loss.backward();
  } else if (sdf_work && schedule_a) {
    streams_->fwd_done.record(main_stream);
    streams_->fwd_done.block(streams_->side);
    Tensor samples_cut = samples.detach().requires_grad_(true);
    for (const Tensor &t : {samples_cut, w_all, ids}) t.record_stream(streams_->side);
    if (cfg_.analytic) { ray_pts.record_stream(streams_->side); ray_sdf.record_stream(streams_->side); }
    Tensor sdf_loss;
    {
      StreamGuard sg(streams_->side);
      sdf_loss = sdf_node(samples_cut, &streams_->gate);
    }
    splat_loss().backward({}, /*retain_graph=*/true);   // the nodes between the samples and the parameters are walked again below
    {
      StreamGuard sg(streams_->side);
      sdf_loss.backward();
      if (!streams_->gate.armed) streams_->gate.record_here();
    }
    Tensor gs = samples_cut.grad();
    if (gs.defined()) {
      gs.record_stream(main_stream);
      if (cfg_.center_reg && views_[0].grad().defined()) {
        // samples = (anchors + offsets)[gaussian_ids]: their gradient is a row scatter into the offsets' gradient — one launch instead of
        // the ~12 of walking JoinGrad -> index_select -> the activation node a second time
        torch::NoGradGuard ng;
        if (streams_->gate.armed) { streams_->gate.event.block(main_stream); streams_->gate.armed = false; }
        views_[0].mutable_grad().index_add_(0, gaussian_ids, gs);
      } else {
        samples.backward(gs);   // JoinGrad (created with the samples) makes the caller's stream wait for the gate first
      }
    }
  } else if (sdf_work) {
    Tensor loss = splat_loss();
    // graph cut at the samples: the SDF leg (forward + backward) on `side`, its d loss / d samples joins the splat leg's
    // backward where the samples' gradient is consumed
    streams_->fwd_done.record(main_stream);
    streams_->fwd_done.block(streams_->side);
    Tensor samples_cut = samples.detach().requires_grad_(true);
    for (const Tensor &t : {samples_cut, w_all, ids}) t.record_stream(streams_->side);
    if (cfg_.analytic) { ray_pts.record_stream(streams_->side); ray_sdf.record_stream(streams_->side); }
    {
      StreamGuard sg(streams_->side);
      // the node records the gate as soon as d loss / d samples has been issued, BEFORE its table scatter: the splat leg's backward
      // tail and the next step's render do not wait for the scatter
      sdf_node(samples_cut, &streams_->gate).backward();
      if (!streams_->gate.armed) streams_->gate.record_here();
    }
    Tensor gs = samples_cut.grad();
    if (gs.defined()) {
      gs.record_stream(main_stream);
      torch::autograd::backward({loss, samples}, {Tensor(), gs});
    } else {
      loss.backward();
    }
  } else {
    splat_loss().backward();
  }
  // ---- train_callback -> update_state (:486, neural_gaussian.cpp:626-680), optimizers (each family on its leg's stream)
  update_state(state_, densify.grad(), gaussian_ids, vis, radii, N, (int)viewmat.size(0), W, H, false);
  if (splat_hook_) splat_hook_(flat_grad_);            // view-parallel: the splat family's gradients are final here
  if (update) {
    adam_.step(/*zero_grad=*/fused_zero());
    if (!fused_zero()) flat_grad_.zero_();
    if (cfg_.reference_terms) count_nan_rows();   // prune_nan_gs's test, no host sync
  }
  {
    c10::optional<StreamGuard> sg;
    if (two) sg.emplace(streams_->side);
    if (sdf_hook_) sdf_hook_(sdf_flat_grad_);          // ... and the SDF family's, on its own stream (after its scatter)
    if (update) {
      adam_sdf_.step(/*zero_grad=*/fused_zero());
      if (!fused_zero()) sdf_flat_grad_.zero_();
    }
  }
  if (two) {   // what the caller's stream has to wait for before it touches the SDF family (sync())
    streams_->side_done.record(streams_->side);
    streams_->side_pending = true;
  }
  sizes["M"] = gaussian_ids.size(0);
  sizes["I"] = std::get<1>(enc).size(0);
  sizes["n_gs_sdf"] = ids.numel();
  return sizes;
}

// ---- the splat leg without the autograd engine ----------------------------------------------------------------------------------
// Same operators, same order of arithmetic as the autograd composition above (tests/test_gpu_bench_step.py compares the flat
// gradients with the Python step's); what goes away is libtorch's glue around them.  In the autograd step the ~0.3 ms of work between
// the compositing backward and the next render took 1.2 ms on the splat leg's stream: 13 AccumulateGrad adds, ~30 zero fills, the
// gather / scatter nodes of index_select and cat, each a 5 us kernel that waits 30-100 us for a CU beside the SDF leg's kernels.
// Here every backward kernel accumulates straight into its segment of the flat gradient buffer (they all += into dense outputs):
//   xyz   <- projection backward + view-colour backward + the samples' gradient          -> offsets' segment (d xyz / d offsets = 1)
//   quats <- projection backward                                                         -> quaternion's segment
//   scales (activated) <- projection backward + isotropic backward -> scratch [N,3]     -> x scale (exp') -> scaling's segment
//   opacities (per visible splat) -> scratch [N] -> x o (1 - o)                          -> opacity's segment
//   SH coefficients <- view-colour backward                                              -> features_dc's segment (degree 0)
bool JointIteration::direct_ok(const Tensor &viewmat) const {
  static const bool off = [] { const char *e = getenv("GSDF_JOINT_DIRECT"); return e && e[0] == '0'; }();
  return !off && cfg_.two_streams && cfg_.analytic && cfg_.reference_terms && viewmat.size(0) == 1;
}

namespace {
const float *ssim_window11() {   // the reference's gaussian(): NOT the symmetric Gaussian (loss_utils.cpp:6-14), as extras.cpp
  static float w[11];
  static const bool init = [] {
    double s = 0, g[11];
    for (int x = 0; x < 11; ++x) { g[x] = std::exp(-std::pow(std::floor((x - 11) / 2.0), 2) / (2.0 * 1.5 * 1.5)); s += g[x]; }
    for (int x = 0; x < 11; ++x) w[x] = (float)(g[x] / s);
    return true;
  }();
  (void)init;
  return w;
}
}  // namespace

std::map<std::string, int64_t> JointIteration::step_direct(const Tensor &viewmat_, const Tensor &K_, const Tensor &target_, const Tensor &ray_pts,
                                                           const Tensor &ray_sdf, bool update, const std::vector<float> &cam_host) {
  using StreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
  torch::NoGradGuard no_grad;
  const int W = cfg_.width, H = cfg_.height;
  std::map<std::string, int64_t> sizes;
  auto main_stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA();
  if (!streams_) streams_ = std::make_unique<JointStreams>();
  streams_->entry.record(main_stream);
  streams_->entry.block(streams_->side);
  streams_->gate.armed = false;
  const int64_t N = anchors_.size(0);
  const auto fopt = anchors_.options().requires_grad(false);
  if (!w_one_.defined()) {
    w_one_ = torch::ones({1}, fopt);
    w_normal_ = torch::full({1}, cfg_.normal_w, fopt);
    w_iso_ = torch::full({1}, cfg_.isotropic_w, fopt);
    zero_image_ = torch::zeros({1, H, W, 1}, fopt);
  }
  if (!scratch_.defined() || scratch_.numel() != 4 * N) scratch_ = torch::empty({4 * N}, fopt);
  Tensor viewmat = f32c(viewmat_.detach(), "viewmat"), K = f32c(K_.detach(), "K"), target = f32c(target_.detach(), "target");
  const int64_t nt = n_table_, nd = n_dec_, nb = n_bias_;
  Tensor tg = sdf_flat_grad_.slice(0, 0, nt), dg = sdf_flat_grad_.slice(0, nt, nt + nd);
  Tensor bg = nb ? sdf_flat_grad_.slice(0, nt + nd, nt + nd + nb) : Tensor();
  auto seg = [&](int k) { return views_[k].grad(); };   // the field's segment of the flat gradient buffer
  // ---- forward: activations, projection (one size read-back), colours, binning (one read-back), compositing, epilogue
  Tensor xyz = torch::empty({N, 3}, fopt), scales = torch::empty({N, 3}, fopt), opac = torch::empty({N}, fopt);
  check(gsdf_splat_activations_fwd(N, fp(anchors_), fp(views_[0]), fp(views_[1]), fp(views_[3]), fpm(xyz), fpm(scales), fpm(opac), cur_stream()),
        "splat_activations_fwd");
  const int64_t Ksh = 1 + n_rest_;
  Tensor sh = n_rest_ == 0 ? views_[4].detach().reshape({N, 1, 3}) : torch::cat({views_[4].detach().reshape({N, 1, 3}), views_[5].detach().reshape({N, n_rest_, 3})}, 1);
  Tensor quats = views_[2].detach();
  Tensor radii_dense = torch::empty({std::max<int64_t>(N, 1)}, fopt.dtype(torch::kInt32));
  Tensor pws = torch::empty({(int64_t)gsdf_projection_2dgs_ws_bytes(N, 1)}, fopt.dtype(torch::kUInt8));
  // The three sizes the host needs (M, I, n_gs_sdf) arrive in host-visible words (gsdf_host_words_alloc): no copy kernel, no stream
  // synchronisation, and what is queued between the launch and the wait runs while the host polls.  GSDF_HOST_COUNTS=0: device scalars.
  if (gsdf_host::HostWords::enabled() && !streams_->counts) streams_->counts = std::make_unique<gsdf_host::HostWords>(4);
  gsdf_host::HostWords *cw = gsdf_host::HostWords::enabled() ? streams_->counts.get() : nullptr;
  Tensor dev_counts = cw ? Tensor() : torch::empty({3}, fopt.dtype(torch::kInt64));
  auto count_ptr = [&](int i) { if (cw) { cw->arm(i); return cw->dev(i); } return dev_counts.data_ptr<int64_t>() + i; };
  // a count is about to size an allocation: anything outside [0, upper] is refused (a word somebody else wrote, or none at all)
  auto count_wait = [&](int i, int64_t upper) {
    const int64_t v = cw ? cw->wait(i) : read_i64(dev_counts[i]);
    TORCH_CHECK(v >= 0 && v <= upper, "JointIteration: implausible count ", v, " in word ", i, " (expected 0..", upper, ")");
    return v;
  };
  check(gsdf_projection_2dgs_cull(N, 1, fp(xyz), fp(quats), fp(scales), fp(viewmat), fp(K), W, H, cfg_.near_plane, cfg_.far_plane, 0.f,
                                  radii_dense.data_ptr<int32_t>(), pws.data_ptr(), count_ptr(0), cur_stream()), "projection(cull)");
  // (while the host waits for M: the zero fill of the backward's scratch and of the two loss values the backward kernels accumulate, the
  // images' allocations)
  scratch_.zero_();
  Tensor loss_values = torch::zeros({2}, fopt);
  auto img = [&](int64_t ch) { return torch::empty({1, H, W, ch}, fopt); };
  Tensor rc = img(3), rd = img(1), ra = img(1), rn = img(3), rm = img(1), fT = torch::empty({1, H, W}, fopt);
  Tensor last = torch::empty({1, H, W}, fopt.dtype(torch::kInt32)), med = torch::empty({1, H, W}, fopt.dtype(torch::kInt32));
  Tensor renders = img(4), nw = img(3), c3 = img(3), d1 = img(1);
  const int64_t M = count_wait(0, N);
  Tensor camera_ids = torch::empty({M}, fopt.dtype(torch::kInt64)), gaussian_ids = torch::empty({M}, fopt.dtype(torch::kInt64));
  Tensor radii = torch::empty({M}, fopt.dtype(torch::kInt32)), means2d = torch::empty({M, 2}, fopt), depths = torch::empty({M}, fopt);
  Tensor rt = torch::empty({M, 3, 3}, fopt), normals = torch::empty({M, 3}, fopt), smp = torch::empty({M, 3}, fopt), sw = torch::empty({M, 1}, fopt);
  // k_center_reg: the SDF samples are the splat centres, weight 1 (neural_gaussian.cpp:259-262); else one stochastic point on every visible
  // splat's disc (SPEC S-3), the seed drawn from torch's default CPU generator as the drop-in operator does
  const bool center = cfg_.center_reg;
  const uint64_t seed = gsplat_cpp::next_sample_seed(/*stochastic=*/!center);
  check(gsdf_projection_2dgs_fill(N, 1, fp(xyz), fp(quats), fp(scales), fp(viewmat), fp(K), W, H, seed, radii_dense.data_ptr<int32_t>(),
                                  pws.data_ptr(), M, M ? camera_ids.data_ptr<int64_t>() : nullptr, M ? gaussian_ids.data_ptr<int64_t>() : nullptr,
                                  M ? radii.data_ptr<int32_t>() : nullptr, fpm(means2d), fpm(depths), fpm(rt), fpm(normals), fpm(smp), fpm(sw),
                                  cur_stream()), "projection(fill)");
  // tile binning, first half (gsplat_cpp::tile_encode's two launches, split at its size read-back): tiles per splat and their running sum
  Tensor tpg = torch::empty({M}, fopt.dtype(torch::kInt32)), cum = torch::empty({std::max<int64_t>(M, 1)}, fopt.dtype(torch::kInt64));
  Tensor tws = torch::empty({(int64_t)gsdf_tile_count_ws_bytes(M)}, fopt.dtype(torch::kUInt8));
  check(gsdf_tile_count(M, W, H, 16, fp(means2d), M ? radii.data_ptr<int32_t>() : nullptr, M ? tpg.data_ptr<int32_t>() : nullptr, cum.data_ptr<int64_t>(),
                        tws.data_ptr(), count_ptr(1), cur_stream()), "tile_encode(count)");
  // (while the host waits for I: the gathers and the colours, which the binning does not need)
  // the visible rows of the activated parameters in one launch (xyz.index_select, opacities.index_select, torch::ones of the centre samples)
  Tensor samples = center ? torch::empty({M, 3}, fopt) : smp, samples_weights = center ? torch::empty({M, 1}, fopt) : sw;
  Tensor pt_opac = torch::empty({M}, fopt);
  check(gsdf_visible_gather(M, M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fp(xyz), fp(opac), center ? fpm(samples) : nullptr, fpm(pt_opac),
                            center ? fpm(samples_weights) : nullptr, cur_stream()), "visible_gather");
  Tensor colors = torch::empty({M, 3}, fopt);
  check(gsdf_view_colors_fwd(M, Ksh, cfg_.sh_degree, fp(viewmat), fp(xyz), fp(sh), M ? camera_ids.data_ptr<int64_t>() : nullptr,
                             M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fpm(colors), cur_stream()), "view_colors_fwd");
  Tensor vis = torch::empty({M, 1}, fopt);
  Tensor offs = torch::empty({1, (H + 15) / 16, (W + 15) / 16}, fopt.dtype(torch::kInt32));
  const int64_t I = count_wait(1, M * (int64_t)(((W + 15) / 16) * ((H + 15) / 16)));
  Tensor isect_ids = torch::empty({I}, fopt.dtype(torch::kInt64)), flat = torch::empty({I}, fopt.dtype(torch::kInt32));
  {
    Tensor ws2 = torch::empty({(int64_t)gsdf_tile_encode_ws_bytes(M, I)}, fopt.dtype(torch::kUInt8));
    check(gsdf_tile_encode(M, 1, I, W, H, 16, fp(means2d), M ? radii.data_ptr<int32_t>() : nullptr, fp(depths), M ? camera_ids.data_ptr<int64_t>() : nullptr,
                           cum.data_ptr<int64_t>(), ws2.data_ptr(), I ? isect_ids.data_ptr<int64_t>() : nullptr, I ? flat.data_ptr<int32_t>() : nullptr,
                           offs.data_ptr<int32_t>(), cur_stream()), "tile_encode");
  }
  // packed splat records + reach masks of the (tile, splat) pairs: written by the compositing forward, reused by its backward
  Tensor raster_fws = torch::empty({(int64_t)gsdf_rasterize_2dgs_fwd_ws_bytes(M, I)}, fopt.dtype(torch::kUInt8));
  check(gsdf_rasterize_2dgs_fwd(1, M, I, W, H, 16, fp(means2d), fp(rt), fp(colors), fp(pt_opac), fp(normals), nullptr, nullptr, offs.data_ptr<int32_t>(),
                                I ? flat.data_ptr<int32_t>() : nullptr, fpm(rc), fpm(rd), fpm(ra), fpm(rn), fpm(rm), last.data_ptr<int32_t>(),
                                med.data_ptr<int32_t>(), fpm(vis), fpm(fT), raster_fws.data_ptr(), cur_stream()), "rasterize_fwd");
  const int64_t P = (int64_t)H * W;
  check(gsdf_render_post_fwd(P, 1, fp(viewmat), fp(rc), fp(rd), fp(ra), fp(rn), fpm(renders), fpm(nw), fpm(c3), fpm(d1), cur_stream()), "render_post_fwd");
  // ---- the visible, occupancy-valid samples (one size), then the SDF leg's forward on the second stream
  Tensor w_all = torch::empty({M, 1}, fopt), ids_all = torch::empty({M}, fopt.dtype(torch::kInt64));
  {
    Tensor vws = torch::empty({(int64_t)gsdf_visible_set_ws_bytes(M)}, fopt.dtype(torch::kUInt8));
    check(gsdf_visible_set(occ_level_, -1, M, fp(samples), origin_.data(), (float)map_size_inv_, occ_grid_.data_ptr(), fp(vis), fp(samples_weights),
                           (float)cfg_.vis_thresh, fpm(w_all), ids_all.data_ptr<int64_t>(), count_ptr(2), vws.data_ptr(), cur_stream()),
          "visible_set");
  }
  streams_->fwd_done.record(main_stream);   // the second stream waits for the visible set, not for what follows it on this one
  // (while the host waits for n_gs_sdf: the photometric loss, which no size depends on)
  Tensor sums = torch::empty({2}, fopt), maps = torch::empty({3, H, W, 3}, fopt), l_normal = loss_values.narrow(0, 0, 1), l_iso = loss_values[1];
  check(gsdf_l1_dssim_fwd(H, W, fp(c3), fp(target), ssim_window11(), fpm(sums), fpm(maps), cur_stream()), "l1_dssim_fwd");
  Tensor ids = ids_all.narrow(0, 0, count_wait(2, M));
  const bool has = ids.numel() > 0;
  streams_->fwd_done.block(streams_->side);
  Tensor samples_cut = samples.detach().requires_grad_(true);
  for (const Tensor &t : {samples_cut, w_all, ids, ray_pts, ray_sdf}) t.record_stream(streams_->side);
  Tensor sdf_loss;
  {
    StreamGuard sg(streams_->side);
    torch::AutoGradMode grad_on(true);
    ResidentGridHint hint(ResidentGridHint::of(cfg_));
    sdf_loss = joint_sdf_loss_analytic(ray_pts, ray_sdf, has ? samples_cut : Tensor(), has ? ids : Tensor(), has ? w_all : Tensor(), *enc_, *dec_, origin_,
                                       map_size_inv_, bce_isigma_, cfg_.sdf_w, cfg_.gs_sdf_w, cfg_.sdf_delta, cfg_.eik_w, cfg_.align_w, tg, dg, bg,
                                       &streams_->gate, /*unit_upstream=*/true, /*first_order_in_forward=*/cfg_.samples_grad_first);
  }
  // ---- the splat leg's losses (values kept for the caller) and their gradients
  std::vector<float> intr, pose;
  if (cam_host.size() >= 16) {
    intr.assign(cam_host.begin(), cam_host.begin() + 4);
    pose.assign(cam_host.begin() + 4, cam_host.begin() + 16);
  } else {   // read the camera back (one device->host copy)
    Tensor Kc = K[0].to(torch::kCPU), c2w = torch::linalg_inv(viewmat[0].to(torch::kCPU).to(torch::kFloat64)).to(torch::kFloat32).contiguous();
    intr = {Kc[0][0].item<float>(), Kc[1][1].item<float>(), Kc[0][2].item<float>(), Kc[1][2].item<float>()};
    pose.assign(c2w.data_ptr<float>(), c2w.data_ptr<float>() + 12);
  }
  last_losses_ = {sums, l_normal, l_iso};   // (the normal-consistency and isotropic VALUES are accumulated by their gradient kernels: no launch of their own)
  Tensor v_c3 = img(3), v_d1 = img(1), v_nw = img(3);
  check(gsdf_l1_dssim_bwd(H, W, fp(c3), fp(target), ssim_window11(), fp(maps), fp(w_one_), (float)cfg_.rgb_w, (float)cfg_.dssim_w, fpm(v_c3), cur_stream()),
        "l1_dssim_bwd");
  constexpr bool fused_values = true;   // the two loss values come out of the fused forward + backward launches
  if (fused_values)
    check(gsdf_normal_consistency_fwd_bwd(H, W, intr.data(), pose.data(), fp(d1), fp(ra), fp(nw), fp(w_normal_), fpm(l_normal), fpm(v_d1), fpm(v_nw),
                                          cur_stream()), "normal_consistency_fwd_bwd");
  else
    check(gsdf_normal_consistency_bwd(H, W, intr.data(), pose.data(), fp(d1), fp(ra), fp(nw), fp(w_normal_), fpm(v_d1), fpm(v_nw), cur_stream()),
          "normal_consistency_bwd");
  Tensor v_scales_act = scratch_.narrow(0, 0, 3 * N).view({N, 3}), v_opac_dense = scratch_.narrow(0, 3 * N, N);
  // ---- backward through the epilogue, the compositing, the colours and the projection
  Tensor v_rc = img(3), v_rd = img(1), v_ra = img(1), v_rn = img(3);
  check(gsdf_render_post_bwd(P, 1, fp(viewmat), fp(rd), fp(ra), nullptr, fp(v_nw), fp(v_c3), fp(v_d1), fpm(v_rc), fpm(v_rd), fpm(v_ra), fpm(v_rn), cur_stream()),
        "render_post_bwd");
  Tensor v_means2d = torch::empty({M, 2}, fopt), v_rt = torch::empty({M, 3, 3}, fopt), v_colors = torch::empty({M, 3}, fopt), v_opac = torch::empty({M}, fopt);
  Tensor v_normals = torch::empty({M, 3}, fopt), v_dens = torch::empty({M, 2}, fopt);
  Tensor rws = torch::empty({(int64_t)gsdf_rasterize_2dgs_bwd_ws_bytes(M, I)}, fopt.dtype(torch::kUInt8));
  check(gsdf_rasterize_2dgs_bwd(1, M, I, W, H, 16, fp(means2d), fp(rt), fp(colors), fp(pt_opac), fp(normals), nullptr, nullptr, offs.data_ptr<int32_t>(),
                                I ? flat.data_ptr<int32_t>() : nullptr, fp(ra), last.data_ptr<int32_t>(), med.data_ptr<int32_t>(), fp(v_rc), fp(v_rd), fp(v_ra),
                                fp(v_rn), fp(zero_image_), fpm(v_means2d), fpm(v_rt), fpm(v_colors), fpm(v_opac), fpm(v_normals), fpm(v_dens), nullptr,
                                rws.data_ptr(), fp(fT), raster_fws.data_ptr(), cur_stream()), "rasterize_bwd");
  // the isotropic term feeds the projection's backward alone (the scales' gradient): behind the compositing backward, which the decoder's forward on the
  // second stream ends up waiting for (a decoder workgroup needs a whole CU) — nothing that is not an input of it runs in front of it
  if (fused_values)
    check(gsdf_isotropic_loss_fwd_bwd(M, fp(scales), M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fp(w_iso_), fpm(l_iso), fpm(v_scales_act), cur_stream()),
          "isotropic_loss_fwd_bwd");
  else
    check(gsdf_isotropic_loss_bwd(M, fp(scales), M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fp(w_iso_), fpm(v_scales_act), cur_stream()),
          "isotropic_loss_bwd");
  // train_callback -> update_state (neural_gaussian.cpp:626-680): needs the densify gradient only
  update_state(state_, v_dens, gaussian_ids, vis, radii, N, 1, W, H, false);
  Tensor v_sh_tmp;
  float *v_sh_ptr;
  if (n_rest_ == 0) {
    Tensor s4 = seg(4);
    v_sh_ptr = s4.data_ptr<float>();
  } else {
    v_sh_tmp = torch::zeros({N, Ksh, 3}, fopt);
    v_sh_ptr = v_sh_tmp.data_ptr<float>();
  }
  Tensor g_off = seg(0), g_sc = seg(1), g_q = seg(2), g_op = seg(3);
  check(gsdf_view_colors_bwd(M, Ksh, cfg_.sh_degree, fp(viewmat), fp(xyz), fp(sh), M ? camera_ids.data_ptr<int64_t>() : nullptr,
                             M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fp(v_colors), v_sh_ptr, fpm(g_off), 1, cur_stream()), "view_colors_bwd");
  // the projection's backward (and the activations' behind it): centre mode right away — the samples' gradient is a row scatter into the
  // offsets' gradient later; stochastic mode once that gradient has arrived (the samples are an output of the projection)
  auto projection_and_activations_bwd = [&](const Tensor &v_samples) {
    check(gsdf_projection_2dgs_bwd(N, 1, M, fp(xyz), fp(quats), fp(scales), fp(viewmat), fp(K), W, H, seed, M ? camera_ids.data_ptr<int64_t>() : nullptr,
                                   M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fp(v_means2d), nullptr /* v_depths = 0 */, fp(v_rt), fp(v_normals), fp(v_samples),
                                   fpm(g_off), fpm(g_q), fpm(v_scales_act), cur_stream()), "projection_bwd");
    check(gsdf_splat_activations_bwd(N, fp(scales), fp(opac), nullptr /* xyz: accumulated in place */, fp(v_scales_act), fp(v_opac_dense), fpm(g_off),
                                     fpm(g_sc), fpm(g_op), cur_stream()), "splat_activations_bwd");
  };
  check(gsdf_rows_scatter_add(M, 1, M ? gaussian_ids.data_ptr<int64_t>() : nullptr, 1, fp(v_opac), fpm(v_opac_dense), cur_stream()), "rows_scatter_add(opacity)");
  if (center) projection_and_activations_bwd(Tensor());
  if (n_rest_ != 0) {
    seg(4).view({N, 1, 3}).add_(v_sh_tmp.narrow(1, 0, 1));
    seg(5).view({N, n_rest_, 3}).add_(v_sh_tmp.narrow(1, 1, n_rest_));
  }
  if (!fused_values) {   // the two values in launches of their own, where this stream is about to wait for the samples' gradient
    check(gsdf_normal_consistency_fwd(H, W, intr.data(), pose.data(), fp(d1), fp(ra), fp(nw), fpm(l_normal), cur_stream()), "normal_consistency_fwd");
    check(gsdf_isotropic_loss_fwd(M, fp(scales), M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fpm(l_iso), cur_stream()), "isotropic_loss_fwd");
  }
  // centre mode, no collective: the samples' gradient reaches the offsets alone, so the optimizer's step over the other 11 of a splat's 14 parameters
  // (scaling, quaternion, opacity, colours) does not wait for the SDF leg — only the offsets' step stays behind the gate (FusedAdam::step_tail / _head)
  const bool split_adam = update && center && !splat_hook_ && cfg_.samples_grad_first;
  if (split_adam) adam_.step_tail(/*head_segments=*/1, /*zero_grad=*/fused_zero());
  // ---- the SDF leg's backward on the second stream, then its d loss / d samples on this one
  {
    StreamGuard sg(streams_->side);
    torch::AutoGradMode grad_on(true);
    if (!one0_.defined()) one0_ = torch::ones({}, fopt);
    torch::autograd::backward({sdf_loss}, {one0_});   // (a kept scalar 1: no fill launch for the root gradient on the loop)
    if (!streams_->gate.armed) streams_->gate.record_here();
  }
  Tensor gs = samples_cut.grad();
  if (gs.defined()) {
    gs = f32c(gs, "samples gradient");
    gs.record_stream(main_stream);
    if (streams_->gate.payload != nullptr && streams_->gate.payload != gs.data_ptr()) {   // the event was recorded at forward time for another buffer: the engine
      StreamGuard sg(streams_->side);                                                     // copied it on the second stream during backward — wait for all of that
      streams_->gate.record_here();
      sizes["samples_grad_copied"] = 1;
    }
    streams_->gate.payload = nullptr;
    if (streams_->gate.armed) { streams_->gate.event.block(main_stream); streams_->gate.armed = false; }
    if (center)   // one camera: every splat appears once among the visible rows -> plain read-modify-write
      check(gsdf_rows_scatter_add(M, 3, M ? gaussian_ids.data_ptr<int64_t>() : nullptr, 1, fp(gs), fpm(g_off), cur_stream()), "rows_scatter_add(samples)");
  }
  if (!center) projection_and_activations_bwd(gs);
  // ---- optimizers, each family on its leg's stream (view-parallel: the family's collective first, on the same stream)
  if (splat_hook_) splat_hook_(flat_grad_);
  if (update) {
    if (split_adam) adam_.step_head(/*head_segments=*/1, /*zero_grad=*/fused_zero());
    else adam_.step(/*zero_grad=*/fused_zero());
    if (!fused_zero()) flat_grad_.zero_();
    count_nan_rows();   // prune_nan_gs's test, no host sync
  }
  {
    StreamGuard sg(streams_->side);
    if (sdf_hook_) sdf_hook_(sdf_flat_grad_);
    if (update) {
      adam_sdf_.step(/*zero_grad=*/fused_zero());
      if (!fused_zero()) sdf_flat_grad_.zero_();
    }
  }
  streams_->side_done.record(streams_->side);
  streams_->side_pending = true;
  sizes["M"] = M;
  sizes["I"] = I;
  sizes["n_gs_sdf"] = ids.numel();
  return sizes;
}

// ---- refinement on the flat buffers (a18) ------------------------------------------------------------------------------------------------
struct RefinePlanArgs {
  gsdf_refine_args a{};
  bool zero_stats = false;   // zero_state() after growing / pruning: grad2d, count (, radii) restart from zero, vis is carried
};

std::map<std::string, int64_t> JointIteration::apply_row_map(RefinePlanArgs &pa) {
  torch::NoGradGuard ng;
  sync();      // the SDF leg of the last step may still be running: it reads this step's samples, not the flat buffers, but its Adam shares no state with us either way
  const int64_t N = anchors_.size(0);
  const auto fopt = anchors_.options().requires_grad(false);
  gsdf_refine_args &a = pa.a;
  a.n = N;
  a.n_rest_cols = (int)field_cols_[5];
  a.flat = fp(flat_);
  Tensor m_old = adam_.exp_avg(0), v_old = adam_.exp_avg_sq(0);
  a.adam_m = fp(m_old); a.adam_v = fp(v_old);
  a.anchors = fp(anchors_);
  auto st = [&](const char *k) -> const float * { auto it = state_.find(k); return it == state_.end() ? nullptr : fp(it->second); };
  a.grad2d = st("grad2d"); a.count = st("count"); a.vis = st("vis"); a.radii = st("radii");
  Tensor counts = torch::empty({4 * std::max<int64_t>(N, 1)}, fopt.dtype(torch::kInt32)), incl = torch::empty({4 * std::max<int64_t>(N, 1)}, fopt.dtype(torch::kInt64));
  Tensor ws = torch::empty({(int64_t)gsdf_refine_ws_bytes(N)}, fopt.dtype(torch::kUInt8));
  // the four totals in host-visible words (the reference reads three .sum().item<int>() and runs nonzero() five times per refinement step)
  int64_t tot[4];
  if (gsdf_host::HostWords::enabled()) {
    if (!refine_words_) refine_words_ = std::make_unique<gsdf_host::HostWords>(4);
    for (int k = 0; k < 4; ++k) refine_words_->arm(k);
    check(gsdf_refine_plan(&a, counts.data_ptr<int32_t>(), incl.data_ptr<int64_t>(), refine_words_->dev(0), ws.data_ptr(), cur_stream()), "refine_plan");
    for (int k = 3; k >= 0; --k) tot[k] = refine_words_->wait(k);
  } else {
    Tensor t4 = torch::empty({4}, fopt.dtype(torch::kInt64));
    check(gsdf_refine_plan(&a, counts.data_ptr<int32_t>(), incl.data_ptr<int64_t>(), t4.data_ptr<int64_t>(), ws.data_ptr(), cur_stream()), "refine_plan");
    Tensor h = t4.cpu();
    for (int k = 0; k < 4; ++k) tot[k] = h.data_ptr<int64_t>()[k];
  }
  for (int k = 0; k < 4; ++k) TORCH_CHECK(tot[k] >= 0 && tot[k] <= N, "JointIteration::refine: implausible total ", tot[k]);
  const int64_t nA = tot[0], nB = tot[1], nC = tot[2], nS = tot[3], Nn = nA + nB + 2 * nC;
  std::map<std::string, int64_t> out = {{"N_before", N}, {"N", Nn}, {"n_kept", nA}, {"n_dupli_kept", nB}, {"n_split_children_kept", 2 * nC}, {"n_split", nS}};
  if (Nn == N && nA == N) {       // nothing moves: keep every buffer (the reference's `if (n > 0)` guards)
    if (pa.zero_stats) for (const char *k : {"grad2d", "count", "radii"}) if (state_.count(k)) state_[k].zero_();
    return out;
  }
  int64_t width = 0;
  for (int64_t c : field_cols_) width += c;
  Tensor randn = nS > 0 ? torch::randn({2, nS, 3}, fopt) : Tensor();      // NeuralGS::split's draw (:781), for ALL split rows
  Tensor flat_new = torch::empty({Nn * width}, fopt), m_new = torch::empty({Nn * width}, fopt), v_new = torch::empty({Nn * width}, fopt);
  Tensor anchors_new = torch::empty({Nn, 3}, fopt);
  std::map<std::string, Tensor> state_new;
  float *sp[4] = {nullptr, nullptr, nullptr, nullptr};
  const char *keys[4] = {"grad2d", "count", "vis", "radii"};
  for (int k = 0; k < 4; ++k) {
    if (!state_.count(keys[k])) continue;
    const bool zero = pa.zero_stats && k != 2;
    state_new[keys[k]] = zero ? torch::zeros({Nn}, fopt) : torch::empty({Nn}, fopt);
    if (!zero) sp[k] = fpm(state_new[keys[k]]);
  }
  check(gsdf_refine_apply(&a, counts.data_ptr<int32_t>(), incl.data_ptr<int64_t>(), tot, nS > 0 ? fp(randn) : nullptr, fpm(flat_new), fpm(m_new), fpm(v_new),
                          fpm(anchors_new), sp, cur_stream()), "refine_apply");
  // rebind: parameter views, gradient buffer (zero: the step that follows accumulates into it), optimizer group, anchors, statistics
  Tensor grad_new = torch::zeros({Nn * width}, fopt);
  anchors_ = anchors_new;
  bind_views(flat_new, grad_new, Nn);
  std::vector<int64_t> sizes;
  for (int64_t c : field_cols_) if (c > 0) sizes.push_back(c * Nn);
  adam_.replace_group(0, flat_, flat_grad_, m_new, v_new, sizes);
  state_ = std::move(state_new);
  return out;
}

// prune_nan_gs's per-iteration test (neural_gaussian.cpp:907-916) added to the running total in ONE launch (no memset, no add kernel)
void JointIteration::count_nan_rows() {
  torch::NoGradGuard ng;
  const int64_t n = anchors_.size(0);
  check(gsdf_nan_rows_accumulate(n, fp(views_[0]), fp(views_[1]), fp(views_[2]), nan_total_.data_ptr<int32_t>(), cur_stream()), "nan_rows_accumulate");
}

// The view-parallel hook reduces statistics IN PLACE (sum of grad2d / count, max of vis / radii over the ranks), which is not idempotent: every
// consumer hands it only the keys it is about to consume and zero afterwards, so no accumulator is ever reduced twice (the invisible
// prune and the refinement of one iteration used to reduce the same map twice, and a prune alone left rank-summed grad2d / count behind).
void JointIteration::reduce_stats(std::initializer_list<const char *> keys) {
  if (!refine_hook_) return;
  std::map<std::string, Tensor> sub;
  for (const char *k : keys) {
    auto it = state_.find(k);
    if (it != state_.end()) sub.emplace(it->first, it->second);      // the same storage: the hook's in-place reduction lands in state_
  }
  if (!sub.empty()) refine_hook_(sub);
}

std::map<std::string, int64_t> JointIteration::refine(int iter, const RefineConfig &rc) {
  reduce_stats({"grad2d", "count", "radii"});
  TORCH_CHECK(state_.count("grad2d") && state_.count("count"), "JointIteration::refine: no densification statistics yet (run step() first)");
  RefinePlanArgs pa;
  pa.zero_stats = true;
  pa.a.mode = 0;
  pa.a.grow_grad2d = rc.grow_grad2d;
  pa.a.grow_scale3d = rc.grow_scale3d * rc.spatial_scale;
  pa.a.grow_scale2d = rc.grow_scale2d;
  pa.a.prune_opa = rc.prune_opa;
  pa.a.prune_scale_min = 1e-4f;
  pa.a.prune_scale3d = rc.prune_scale3d * rc.original_spatial_scale;
  pa.a.use_radii = iter < rc.refine_scale2d_stop_iter && state_.count("radii") ? 1 : 0;
  pa.a.use_prune_scale3d = iter > rc.reset_every ? 1 : 0;
  return apply_row_map(pa);
}

int64_t JointIteration::prune_rows(const Tensor &mask) {
  TORCH_CHECK(mask.defined() && mask.numel() == anchors_.size(0), "JointIteration::prune_rows: one mask entry per splat");
  Tensor m8 = mask.to(torch::kUInt8).contiguous();
  RefinePlanArgs pa;
  pa.a.mode = 1;
  pa.a.mask = m8.data_ptr<uint8_t>();
  auto r = apply_row_map(pa);
  return r["N_before"] - r["N"];
}

void JointIteration::reset_opacity(const RefineConfig &rc) {
  torch::NoGradGuard ng;
  const double cap = std::log(rc.prune_opa * 2.0 / (1.0 - rc.prune_opa * 2.0));
  views_[3].clamp_max_(cap);
  // segment index of the opacity among the non-empty segments (features_rest may be empty)
  int seg = 0;
  for (int i = 0; i < 3; ++i) seg += field_cols_[i] > 0 ? 1 : 0;
  adam_.zero_segment_moments(0, seg);
}

std::map<std::string, int64_t> JointIteration::train_callback(int iter, int total_iter, const RefineConfig &rc) {
  std::map<std::string, int64_t> out = {{"N", anchors_.size(0)}};
  TORCH_CHECK(rc.num_train_data > 0 && rc.refine_every > 0 && rc.reset_every > 0,
              "JointIteration::train_callback: num_train_data, refine_every and reset_every are periods (> 0), got ", rc.num_train_data, ", ",
              rc.refine_every, ", ", rc.reset_every);
  if (iter >= total_iter / 2) return out;       // refine_stop_iter (:573-578)
  // prune_nan_gs (:907-916): step() already accumulates the count without a host round trip (nan_splats_seen()); rows are removed by the caller
  // with prune_rows() when that count moves.  prune_invisible_gs (:892-905):
  if (iter > 0 && iter % rc.num_train_data == 0 && state_.count("vis")) {
    reduce_stats({"vis"});
    Tensor invisible = state_["vis"] < 1e-4;
    state_["vis"].zero_();
    out["n_invisible"] = prune_rows(invisible);
  }
  if (iter > 0) {
    if (iter > rc.refine_start_iter && iter % rc.refine_every == 0 && (iter % rc.reset_every) >= rc.pause_refine_after_reset) {
      auto r = refine(iter, rc);
      out.insert(r.begin(), r.end());
    }
    if (iter % rc.reset_every == 0) reset_opacity(rc);
  }
  out["N"] = anchors_.size(0);
  return out;
}

}  // namespace gsdf_extras
