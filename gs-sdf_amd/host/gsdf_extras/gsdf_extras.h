// gsdf_extras/gsdf_extras.h — the fused pieces of the training step that are NOT symbols of the replaced submodules, as
// C++/libtorch operators in libgsdf_torch.so, so that the speed bench.py measures is reachable from the reference's own
// C++ (each replaces a handful of eager libtorch kernels at the cited place; INTEGRATION.md section 5 shows the edits).
//   l1_dssim_loss        k_rgb_weight * loss::rgb_loss + k_dssim_weight * loss::dssim_loss   neural_mapping.cpp:237-240
//   query_points         SubMap::xyz_to_zp1_pts (+ the 6-point stencil of LocalMap::get_gradient)  sub_map.cpp:82-97, local_map.cpp:110-124
//   sdf_ray_loss         loss::sdf_loss + w * loss::eikonal_loss(numerical gradient)          neural_mapping.cpp:138-188, loss.cpp:49-83
//   gs_sdf_eik_loss      k_gs_sdf_weight * loss::gs_sdf_loss + k_eikonal_weight * eikonal on the splat samples   :436-457
//   gs_sdf_coupling      the whole GS<->SDF block as ONE autograd node: samples -> get_sdf -> gs_sdf_loss (+ eikonal on the
//                        numerical gradient at samples.detach()), encoder / decoder gradients accumulated in place   :420-462
//   update_state         NeuralGS::update_state                                               neural_gaussian.cpp:626-680
//   splat_activations    NeuralGS::generate_gaussian's exp / sigmoid / anchors + offsets      neural_gaussian.cpp:463-492
//   render_post          expected depth, cat(colours, depth), normals to world space          neural_gaussian.cpp:229-240
//   FusedAdam            torch::optim::Adam::step over flat (parameter, gradient) buffers     neural_mapping.cpp:466-469
//   JointIteration       the whole loop body on the above (what bench.py --cpp-step times)     neural_mapping.cpp:400-486
#pragma once
#include <torch/torch.h>

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

struct TCNNEncoding;
struct TCNNNetwork;

namespace gsdf_extras {

struct StreamGate;   // internal (src/stream_gate.h): "the tensor produced on another stream is complete"

// [H,W,3] x [H,W,3] -> scalar; differentiable w.r.t. render
torch::Tensor l1_dssim_loss(const torch::Tensor &render, const torch::Tensor &gt, double rgb_weight = 0.8, double dssim_weight = 0.2);

// world points [n,3] -> unit-cube encoder inputs; with_stencil: 7n rows (base, then +x,-x,+y,-y,+z,-z blocks of n rows)
torch::Tensor query_points(const torch::Tensor &xyz, const std::vector<float> &map_origin, double map_size_inv, bool with_stencil,
                           double delta);

// attr [(7 or 1) n, >=2] = decoder output on query_points' rows -> scalar (BCE sdf loss [+ w_eik * eikonal]); d/d attr
torch::Tensor sdf_ray_loss(const torch::Tensor &attr, const torch::Tensor &gt_sdf, int64_t n, double bce_isigma, double delta, double w_eik);

// attr [(7 or 1) n, >=1]; weights [M] (or [M,1]); ids int64 [n] (undefined: weights is [n]) -> scalar; d/d attr
torch::Tensor gs_sdf_eik_loss(const torch::Tensor &attr, const torch::Tensor &weights, const torch::Tensor &ids, int64_t n, double scale,
                              double delta, double w_eik);

// The GS<->SDF coupling of the joint iteration (neural_mapping.cpp:420-462) as one node: rows `ids` of `samples` [M,3] ->
// query_points (+ 6-point stencil when delta > 0) -> encoder (group-walking forward) -> decoder -> scale * gs_sdf_loss
// (+ w_eik * eikonal_loss(numerical gradient at the detached samples)).  backward(): d loss / d samples is returned to autograd;
// the encoder's table gradient and the decoder's weight gradient are ACCUMULATED IN PLACE into `table_grad` / `decoder_grad`
// (fp32, same shapes as enc.params_ / dec.params_, e.g. views of the flat gradient buffer FusedAdam reads; zero them per
// step) — one one-pass decoder backward, one Jacobian contraction, one binned stencil-merging scatter, no autograd adds.
// `weights` = samples_weights * visibilities, [M] or [M,1].  Bias-free decoder (TCNNNetwork).
// `samples_grad_ready` (optional, internal type): recorded on the backward's stream once d loss / d samples has been issued, before
// the table scatter — lets a caller on another stream consume that gradient without waiting for the scatter (JointIteration).
torch::Tensor gs_sdf_coupling(const torch::Tensor &samples, const torch::Tensor &ids, const torch::Tensor &weights, ::TCNNEncoding &enc,
                              ::TCNNNetwork &dec, const std::vector<float> &map_origin, double map_size_inv, double scale, double delta,
                              double w_eik, torch::Tensor table_grad, torch::Tensor decoder_grad, StreamGate *samples_grad_ready = nullptr);

// ---- the reference's DEFAULT SDF configuration (config/base.yaml:12-13: decoder_implementation 0, numerical_grad 0) -----------
// The SDF work of one joint iteration as ONE autograd node (Python mirror: gs_sdf_amd/sdf.py _SdfBatchAnalytic): one batch holds
//   rows [0, n_ray)  the per-ray batch:  w_sdf * loss::sdf_loss(get_sdf(ray_xyz), gt_sdf)                       neural_mapping.cpp:165-170
//   rows [n_ray, n)  samples[ids]:       w_gs * loss::gs_sdf_loss(get_sdf(samples[ids]), weights[ids])          :436-457
//   each set: + w_eik * eikonal_loss(ANALYTIC gradient, autograd::grad(create_graph = true), local_map.cpp:151-172; the splat samples
//             detached, :448-451) + w_align * mean |analytic - numerical.detach()| (6 forward-only stencil rows, :126-134).
// Either part may be empty (undefined tensor).  d loss / d samples is returned to autograd; the encoder's table gradient and the decoder's
// unit_upstream: the caller promises to call backward() on the returned loss itself (upstream gradient exactly 1): three scaling launches go.
// first_order_in_forward (needs unit_upstream): the FIRST-ORDER chain (data-term gradient -> one-pass decoder backward -> Jacobian contraction =
// d loss / d samples, `samples_grad_ready` recorded behind it) runs inside the forward call, ahead of the stencil rows' decoder pass, the e0
// backward and the loss kernel: the splat leg gets its samples' gradient earlier by those three launches; the first order's parameter gradients
// are then accumulated at forward time.  Same bits as the backward-time order in deterministic mode (tests/test_gpu_host_layer.py).
// weight / bias gradients are ACCUMULATED IN PLACE into table_grad / decoder_grad / bias_grad (views of the flat gradient buffer):
// one-pass decoder backward, decoder double backward (gsdf_mlp_bwd_bwd), ONE binned scatter carrying the first- and second-order
// table gradient (gsdf_hashgrid_bwd_binned2).  dec may carry biases (config "bias": true = the torch decoder's topology).
torch::Tensor joint_sdf_loss_analytic(const torch::Tensor &ray_xyz, const torch::Tensor &gt_sdf, const torch::Tensor &samples,
                                      const torch::Tensor &ids, const torch::Tensor &weights, ::TCNNEncoding &enc, ::TCNNNetwork &dec,
                                      const std::vector<float> &map_origin, double map_size_inv, double bce_isigma, double w_sdf, double w_gs,
                                      double delta, double w_eik, double w_align, torch::Tensor table_grad, torch::Tensor decoder_grad,
                                      torch::Tensor bias_grad, StreamGate *samples_grad_ready = nullptr, bool unit_upstream = false,
                                      bool first_order_in_forward = false);

// render_normal_weight's term (neural_mapping.cpp:243-266): mean(alpha^2 - nan_to_num((depth_to_normal(depth) * alpha) . render_normal)),
// alpha detached.  depth [H,W,1], alpha [H,W,1], render_normal [H,W,3] (world); intrinsics {fx, fy, cx, cy} and the camera->world pose
// (row-major [3,4]) as HOST values (the trainer knows its poses: no device->host copy per iteration).
torch::Tensor normal_consistency_loss(const torch::Tensor &depth, const torch::Tensor &alpha, const torch::Tensor &render_normal,
                                      const std::vector<float> &intrinsics4, const std::vector<float> &pose_c2w);
// isotropic_weight's term (neural_mapping.cpp:268-276) on the activated scales [N,3] of the visible splats gaussian_ids [M]
torch::Tensor isotropic_loss(const torch::Tensor &scales, const torch::Tensor &gaussian_ids);
// NeuralGS::prune_nan_gs's test (neural_gaussian.cpp:907-916): int32 device scalar [1] = splats with a NaN; mask (optional out) bool [N]
torch::Tensor nan_rows(const torch::Tensor &offsets, const torch::Tensor &scaling, const torch::Tensor &quaternion, torch::Tensor *mask = nullptr);

// state: "grad2d","count","vis"[,"radii"] created on first use; info as NeuralGS::render returns it
void update_state(std::map<std::string, torch::Tensor> &state, const torch::Tensor &densify_grad, const torch::Tensor &gaussian_ids,
                  const torch::Tensor &visibilities, const torch::Tensor &radii, int64_t n_gaussians, int n_cameras, int width,
                  int height, bool want_radii);

// rasterization_2dgs_sdf's tail (neural_gaussian.cpp:229-240) in one pass, differentiable: expected depth (render_depths / alphas when
// expected_depth), cat(colours, depth), normals rotated to world space -> {renders [C,H,W,4], render_normals_world [C,H,W,3],
// colours [C,H,W,3], depth [C,H,W,1]} (the last two = the slices NeuralGS::render takes, :533-543)
std::vector<torch::Tensor> render_post(const torch::Tensor &render_colors, const torch::Tensor &render_depths, const torch::Tensor &render_alphas,
                                       const torch::Tensor &render_normals, const torch::Tensor &viewmats, bool expected_depth);

// (anchors, offsets, log-scales [N,3], logit-opacities [N]) -> (xyz, scales, opacities); differentiable
std::vector<torch::Tensor> splat_activations(const torch::Tensor &anchors, const torch::Tensor &offsets, const torch::Tensor &scaling,
                                             const torch::Tensor &opacity);

// torch::optim::Adam semantics (no amsgrad / weight decay) over flat buffers, one launch per group
class FusedAdam {
 public:
  FusedAdam(double beta1 = 0.9, double beta2 = 0.999, double eps = 1e-15) : b1_(beta1), b2_(beta2), eps_(eps) {}
  // flat / flat_grad: contiguous fp32 device buffers; segments = consecutive (n_elements, lr) pieces covering the buffer
  int add_group(const torch::Tensor &flat, const torch::Tensor &flat_grad, const std::vector<int64_t> &sizes, const std::vector<double> &lrs);
  void set_lr(int group, int segment, double lr) { groups_.at(group).lrs.at(segment) = (float)lr; }
  void step(bool zero_grad = false);   // zero_grad: the gradient buffers are zeroed as they are consumed (no fill launches)
  // step() of group 0 in two launches (same bits): step_tail — everything behind its first `head_segments` segments, and the other groups; the step
  // count advances —, then step_head — those first segments, whose gradient arrives last
  void step_tail(int head_segments, bool zero_grad = false);
  void step_head(int head_segments, bool zero_grad = false);
  int64_t step_count() const { return t_; }
  // refinement (optimizer_utils.cpp:5-165 on flat buffers): the group's buffers are replaced by re-materialised ones of another row count;
  // `sizes` = the new segment sizes (learning rates are kept); the step count is kept, as torch::optim::Adam's per-parameter state keeps it
  void replace_group(int group, const torch::Tensor &flat, const torch::Tensor &flat_grad, const torch::Tensor &m, const torch::Tensor &v,
                     const std::vector<int64_t> &sizes);
  torch::Tensor exp_avg(int group) const { return groups_.at(group).m; }
  torch::Tensor exp_avg_sq(int group) const { return groups_.at(group).v; }
  void zero_segment_moments(int group, int segment);   // reset_optimizer of one parameter (reset_opacity, neural_gaussian.cpp:918-926)

 private:
  struct Group {
    torch::Tensor flat, grad, m, v;
    std::vector<int64_t> begins;
    std::vector<float> lrs;
  };
  std::vector<Group> groups_;
  double b1_, b2_, eps_;
  int64_t t_ = 0;
};

// The loop body of NeuralSLAM::gs_train (neural_mapping.cpp:400-486) in C++ on the drop-in operators + the pieces above
// (src/joint_step.cpp): per-ray SDF batch, render + 0.8 L1 + 0.2 D-SSIM, GS<->SDF coupling with the eikonal regulariser, backward,
// update_state, fused Adam.  What `bench.py --cpp-step` times; one HIP stream, or two (JointConfig::two_streams).
struct JointConfig {
  int width = 0, height = 0, sh_degree = 0;
  float near_plane = 0.05f, far_plane = 300.0f;
  double rgb_w = 0.8, dssim_w = 0.2;                                   // neural_mapping.cpp:237-240
  double sdf_delta = 0.02, eik_w = 0.1, gs_sdf_w = 1e-3, vis_thresh = 0.1;   // k_sample_std, k_eikonal_weight, k_gs_sdf_weight, :430-432
  double sdf_w = 1.0, align_w = 0.1;                                   // k_sdf_weight, k_align_weight (config/base.yaml:27-30)
  // analytic = the reference's DEFAULT SDF configuration (numerical_grad: 0): eikonal on the analytic gradient + align term, the per-ray
  // batch and the splat samples in ONE fused batch (joint_sdf_loss_analytic); false = the numerical-gradient configuration the
  // reference forces with the tcnn decoder (params.cpp:396-399)
  bool analytic = true;
  // reference_terms: render_normal_weight x depth->normal consistency + isotropic_weight x isotropic regulariser + prune_nan test
  // (neural_mapping.cpp:243-276, neural_gaussian.cpp:907-916); false: `upstream` op-level gradients instead (the round-2 step)
  bool reference_terms = true;
  double normal_w = 0.01, isotropic_w = 0.05;                          // config/base.yaml:43-44
  double lr_offsets = 1.6e-4, lr_scaling = 5e-3, lr_quaternion = 1e-3, lr_opacity = 5e-2, lr_features_dc = 2.5e-3,
         lr_features_rest = 2.5e-3 / 20, lr_sdf = 1e-4;               // neural_gaussian.cpp:434-453, :619-623
  // center_reg (k_center_reg): the SDF samples are the splat centres with weight 1; false = the reference's DEFAULT (the key is absent from
  // config/base.yaml -> 0): one stochastic point on every visible splat's disc, weight exp(-|eps|^2 / 2) (neural_gaussian.cpp:259-265)
  bool center_reg = true;
  bool two_streams = false;   // the SDF network's work on a second HIP stream beside the splat leg (bench.py's overlapped schedule)
  bool samples_grad_first = true;   // two_streams + analytic, direct splat leg: joint_sdf_loss_analytic(first_order_in_forward) — the samples' gradient (what the splats'
                              // optimizer and the next render wait for) leaves the SDF leg before the stencil rows' decoder pass, the e0 backward and the loss kernel
  int hashgrid_resident = 3;  // two_streams + analytic only: workgroups per CU of the stencil hash-grid forward's RESIDENT grid (gsdf_hashgrid_fwd_stencil_resident):
                              // the rest of every CU stays free for the splat leg's kernels; 0 = the full grid.  Round 6 (tools/ab_lib.sh, same box, two runs
                              // each): 2 -> 4.51 ms per step (hash-grid forward 2.11 ms beside the compositing backward), 3 -> 4.26 (1.62 ms = its time alone:
                              // three 128-register waves + one 96-register wave of the quad-list backward fill a SIMD's file), 4 -> 4.57, 5 -> 4.48, 6 -> 4.45, 1 -> 5.8;
                              // end of round 6 (the kernel fed with world points: 83 registers): 3 -> 3.86-3.88, 4 -> 3.92-3.93, 5 -> 3.97, 0 -> 3.98, 2 -> 4.04
};

// the refinement policy's configuration: GSConfig's fields (config/base.yaml:60-74) + the scene scale NeuralGS keeps (neural_gaussian.cpp:286-290)
struct RefineConfig {
  float prune_opa = 0.05f, grow_grad2d = 0.0002f, grow_scale3d = 0.01f, grow_scale2d = 0.05f, prune_scale3d = 0.1f;
  int refine_scale2d_stop_iter = 0, refine_start_iter = 500, refine_every = 100, reset_every = 3000, pause_refine_after_reset = 0;
  float spatial_scale = 1.f, original_spatial_scale = 1.f;
  int num_train_data = 1;
};

struct JointStreams;
struct RefinePlanArgs;
}  // namespace gsdf_extras
namespace gsdf_host { class HostWords; }
namespace gsdf_extras {
class JointIteration {
 public:
  ~JointIteration();
  // fields = {offsets [N,3], scaling [N,3] (log), quaternion [N,4], opacity [N] (logit), features_dc [N,3], features_rest [N,3 r]};
  // enc / dec: their params_ are MOVED into this object's flat SDF buffer (they become views of it).
  JointIteration(const torch::Tensor &anchors, const std::vector<torch::Tensor> &fields, std::shared_ptr<::TCNNEncoding> enc,
                 std::shared_ptr<::TCNNNetwork> dec, const std::vector<float> &map_origin, double map_size, double bce_sigma, int occ_level,
                 const JointConfig &cfg);
  // one iteration on view (viewmat [1,4,4], K [1,3,3]) against target [H,W,3] with the ray batch (ray_pts [n,3], ray_sdf [n,1]);
  // upstream: {} or the op-level gradients {v_depth [1,H,W,1], v_alpha [1,H,W,1], v_normal [1,H,W,3], v_median [1,H,W,1]} (used when
  // !cfg.reference_terms); cam_host: {} or {fx, fy, cx, cy, camera->world pose row-major [3,4]} as HOST values for the normal-consistency
  // term (a trainer knows its poses; when empty they are read back from viewmat / K: one device->host copy);
  // update = false leaves the gradients in the flat buffers and skips the optimizers.  Returns {"M","I","n_gs_sdf"}.
  // STREAM CONTRACT: inputs must have been produced on the caller's current stream (or be complete); with two_streams the second
  // stream waits for the caller's stream at entry, and the SDF family's update is complete on the caller's stream only after
  // sync() — the accessors sdf_flat() / sdf_flat_grad() call it.
  std::map<std::string, int64_t> step(const torch::Tensor &viewmat, const torch::Tensor &K, const torch::Tensor &target,
                                      const torch::Tensor &ray_pts, const torch::Tensor &ray_sdf, const std::vector<torch::Tensor> &upstream,
                                      bool update = true, const std::vector<float> &cam_host = {});
  void sync();   // the caller's current stream waits for everything this object has issued on its second stream
  // View-parallel training (the reference is single-GPU): called with the family's flat gradient buffer after the backward
  // pass and before that family's optimizer step, on the stream the family's optimizer runs on — e.g. an RCCL all-reduce + 1/G scale.
  void set_grad_hooks(std::function<void(torch::Tensor)> splat_hook, std::function<void(torch::Tensor)> sdf_hook) {
    splat_hook_ = std::move(splat_hook); sdf_hook_ = std::move(sdf_hook);
  }
  // ---- refinement on the flat buffers (csrc/refine.hip) -------------------------------------------------------------
  // NeuralGS::train_callback's structural part (neural_gaussian.cpp:568-618) for iteration `iter` (call it after step(): update_state has run):
  // prune_invisible_gs every num_train_data iterations; grow_gs + prune_gs + zero_state when iter > refine_start_iter, iter % refine_every == 0
  // and (iter % reset_every) >= pause_refine_after_reset; reset_opacity every reset_every.  -> {"n_dupli","n_split","n_prune","n_invisible","N"}
  std::map<std::string, int64_t> train_callback(int iter, int total_iter, const RefineConfig &rc);
  // grow_gs (duplicate + split) + prune_gs + zero_state as ONE row map: two kernels over the rows, the four counts in host-visible words
  std::map<std::string, int64_t> refine(int iter, const RefineConfig &rc);
  int64_t prune_rows(const torch::Tensor &mask);     // rows with mask != 0 leave (prune_invisible_gs / prune_nan_gs); -> rows removed
  void reset_opacity(const RefineConfig &rc);        // :918-926: opacity logits clamped to logit(2 prune_opa), fresh moments
  // View-parallel: called with the densification statistics a decision is about to consume — {"vis"} before the invisible prune, {"grad2d",
  // "count", "radii"} before a refinement — and only those (sum grad2d / count, max vis / radii over the ranks, in place); each is zeroed by
  // its consumer, so no accumulator is reduced twice
  void set_refine_hook(std::function<void(std::map<std::string, torch::Tensor> &)> hook) { refine_hook_ = std::move(hook); }
  torch::Tensor anchors() const { return anchors_; }
  std::map<std::string, torch::Tensor> &state() { return state_; }
  std::vector<torch::Tensor> splat_adam_moments() const { return {adam_.exp_avg(0), adam_.exp_avg_sq(0)}; }
  int64_t n_splats() const { return anchors_.size(0); }
  torch::Tensor splat_flat() const { return flat_; }
  torch::Tensor splat_flat_grad() const { return flat_grad_; }
  torch::Tensor sdf_flat() { sync(); return sdf_flat_; }
  torch::Tensor sdf_flat_grad() { sync(); return sdf_flat_grad_; }
  torch::Tensor nan_splats_seen() const { return nan_total_; }   // int32 device scalar: sum of the per-iteration prune_nan counts
  // {photometric 0.8 L1 + 0.2 D-SSIM sums (2 floats: sum |I - G|, sum SSIM), normal-consistency loss, isotropic loss} of the last step of
  // the direct splat leg (device tensors; empty before the first such step)
  std::vector<torch::Tensor> last_losses() const { return last_losses_; }

 private:
  JointConfig cfg_;
  std::shared_ptr<::TCNNEncoding> enc_;
  std::shared_ptr<::TCNNNetwork> dec_;
  std::vector<float> origin_;
  double map_size_inv_, bce_isigma_;
  int occ_level_;
  int64_t n_rest_ = 0;
  torch::Tensor anchors_, flat_, flat_grad_, sdf_flat_, sdf_flat_grad_, occ_grid_, nan_total_;
  int64_t n_table_ = 0, n_dec_ = 0, n_bias_ = 0;
  std::vector<int64_t> field_cols_;                    // columns per row of the six fields (3, 3, 4, 1, 3, 3 r)
  std::vector<std::vector<int64_t>> field_shapes_;     // their trailing shapes
  std::unique_ptr<gsdf_host::HostWords> refine_words_;
  std::vector<torch::Tensor> views_;
  std::map<std::string, torch::Tensor> state_;
  FusedAdam adam_, adam_sdf_;
  std::unique_ptr<JointStreams> streams_;
  std::function<void(torch::Tensor)> splat_hook_, sdf_hook_;
  std::function<void(std::map<std::string, torch::Tensor> &)> refine_hook_;
  void count_nan_rows();
  void reduce_stats(std::initializer_list<const char *> keys);   // the hook on the statistics a consumer is about to use up, and only those
  std::map<std::string, int64_t> apply_row_map(RefinePlanArgs &pa);   // plan -> totals -> new buffers -> apply -> rebind
  void bind_views(const torch::Tensor &flat, const torch::Tensor &flat_grad, int64_t n);
  // the splat leg without the autograd engine (step_direct): loss weights as device scalars, a never-written zero image, scratch
  torch::Tensor w_one_, one0_, w_normal_, w_iso_, zero_image_, scratch_;
  std::vector<torch::Tensor> last_losses_;
  bool direct_ok(const torch::Tensor &viewmat) const;
  std::map<std::string, int64_t> step_direct(const torch::Tensor &viewmat, const torch::Tensor &K, const torch::Tensor &target,
                                             const torch::Tensor &ray_pts, const torch::Tensor &ray_sdf, bool update, const std::vector<float> &cam_host);
};

}  // namespace gsdf_extras
