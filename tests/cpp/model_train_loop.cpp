// model_train_loop.cpp — a C++ PROGRAM (no Python) that links libgsdf_torch.so + libgsdf_hip.so and trains a small scene with the
// reference's own control flow on gsdf_model::NeuralGS / gsdf_model::LocalMap: what NeuralSLAM::gs_train (neural_mapping.cpp:400-486)
// does per iteration — per-ray SDF batch (sample_rays -> get_sdf -> sdf_loss + eikonal on the numerical gradient), render,
// 0.8 L1 + 0.2 D-SSIM, GS<->SDF coupling at the visible samples, backward, torch::optim::Adam, train_callback with refinement.
// Built by tests/test_gpu_cpp_program.py (g++ against libtorch and the two shared objects) and run on the GPU box: the model classes
// are linked and executed from C++, not only type-checked.
#include <torch/torch.h>

#include <cstdio>
#include <vector>

#include "gsdf_extras/gsdf_extras.h"
#include "gsdf_model/gsdf_model.h"

using torch::Tensor;

int main() {
  if (!torch::cuda::is_available()) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  torch::manual_seed(0);
  const auto dev = torch::Device(torch::kCUDA, 0);
  const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  const int W = 256, H = 192;
  const float fx = 0.8f * W, fy = 0.8f * W, cx = 0.5f * (W - 1), cy = 0.5f * (H - 1);
  // a bumpy wall 4 m in front of the camera: surface points + the rays that see them
  const int64_t n_pts = 6000;
  Tensor uv = torch::rand({n_pts, 2}, f32);
  Tensor px = uv.select(1, 0) * W, py = uv.select(1, 1) * H;
  Tensor z = 4.0f + 0.3f * torch::sin(px * 0.05f) * torch::cos(py * 0.07f);
  Tensor pts = torch::stack({(px - cx) * z / fx, (py - cy) * z / fy, z}, 1).contiguous();
  // ---- the SDF map, its occupancy structure from the surface points
  gsdf_model::MapConfig mcfg;
  mcfg.leaf_size = 0.25f;
  mcfg.inner_map_size = 15.0f;
  mcfg.decoder_implementation = 0;   // the reference's default: torch decoder topology, analytic gradient available
  auto lm = std::make_shared<gsdf_model::LocalMap>(torch::tensor({0.0f, 0.0f, 4.0f}), mcfg);
  lm->update_octree_as(pts);
  // ---- splats from the points, rotations / opacities from the (untrained) SDF as NeuralGS(points) does
  gsdf_model::GSConfig gcfg;
  gcfg.sh_degree = 1;
  gcfg.refine_start_iter = 10;
  gcfg.refine_every = 5;
  gcfg.reset_every = 1000;
  gcfg.grow_grad2d = 1e-7f;
  gcfg.center_reg = false;           // the reference's default: stochastic SDF samples on the discs
  gcfg.geo_init = false;             // (an untrained SDF has no useful normals yet)
  auto gs = std::make_shared<gsdf_model::NeuralGS>(lm, pts, 4, 1.0f, true, gcfg);
  // ---- one Adam over (SDF groups, splat groups), as neural_mapping.cpp:846-858
  std::vector<torch::optim::OptimizerParamGroup> groups;
  for (auto &p : lm->parameters()) {
    auto o = std::make_unique<torch::optim::AdamOptions>(5e-3);
    o->eps(1e-15);
    groups.emplace_back(std::vector<Tensor>{p}, std::move(o));
  }
  gs->gs_param_start_idx = (int)groups.size();
  for (auto &g : gs->optimizer_params_groups_) groups.push_back(g);
  auto adam = std::make_shared<torch::optim::Adam>(groups, torch::optim::AdamOptions(1e-3).eps(1e-15));
  // ---- views and targets (a different image to fit: the initial render, dimmed)
  gsdf_model::Cameras cam;
  cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.width = W; cam.height = H;
  std::vector<Tensor> poses, targets;
  for (int v = 0; v < 4; ++v) {
    Tensor pose = torch::eye(4).slice(0, 0, 3).clone();   // camera -> world [3,4]
    pose[0][3] = 0.1f * (v - 1.5f);
    poses.push_back(pose);
    torch::NoGradGuard ng;
    targets.push_back(gs->render(pose, cam, false)["color"].detach() * 0.5f + 0.25f);
  }
  // the sensor's rays for the SDF batch: origin 0, direction to the surface points, depth = range
  gsdf_model::DepthSamples rays;
  Tensor range = pts.norm(2, 1, true);
  rays.origin = torch::zeros_like(pts);
  rays.direction = pts / range;
  rays.depth = range;
  rays.xyz = pts;
  const float delta = 0.02f, trunc = 3 * mcfg.leaf_size;
  const int total_iter = 80;
  float first = 0.f, last = 0.f, before_refine = 0.f;
  int64_t n_first = gs->anchors_.size(0), n_last = n_first;
  for (int it = 1; it <= 30; ++it) {
    adam->zero_grad();
    // per-ray SDF batch (neural_mapping.cpp:73-104, 138-188)
    auto batch = gsdf_model::sample_rays(*lm, rays.index_select(torch::randint(n_pts, {2048}, torch::TensorOptions().dtype(torch::kInt64).device(dev))),
                                         delta, trunc, 3, true);
    auto si = lm->get_sdf(batch.xyz);
    Tensor target_occ = torch::sigmoid(-batch.ray_sdf * (1.0 / mcfg.bce_sigma)).clamp(1e-7, 1 - 1e-7);
    Tensor sdf_loss = torch::binary_cross_entropy_with_logits(-si[0] * si[1], target_occ);                 // loss.cpp:49-79
    Tensor grad = lm->get_gradient(batch.xyz.detach(), delta, Tensor(), false, true)[0];
    Tensor eik = (grad.norm(2, 1) - 1.0f).square().mean();                                               // loss.cpp:81-83
    // render + photometric loss (neural_mapping.cpp:195-240)
    auto r = gs->render(poses[it % 4], cam, true);
    Tensor loss = gsdf_extras::l1_dssim_loss(r["color"], targets[it % 4], 0.8, 0.2) + sdf_loss + 0.1 * eik;
    // GS <-> SDF coupling at the visible, occupancy-valid samples (neural_mapping.cpp:420-462)
    Tensor vis = r["visibilities"].detach();
    Tensor valid = lm->get_valid_mask(r["samples"].detach()) & (vis > 0.1).squeeze(-1);
    Tensor ids = valid.nonzero().squeeze(-1);
    if (ids.numel() > 0) {
      Tensor s = lm->get_sdf(r["samples"].index_select(0, ids))[0];
      Tensor w = (r["samples_weights"] * vis).detach().index_select(0, ids);
      loss = loss + 1e-3 * 0.5 * (w * s.square()).sum() / (double)ids.numel();                           // loss.cpp:7-11
    }
    loss.backward();
    adam->step();
    gs->train_callback(it, total_iter, adam, r);
    const float lv = loss.item<float>();
    if (it == 1) first = lv;
    if (it == gcfg.refine_start_iter) before_refine = lv;   // until here only Adam has moved the parameters
    last = lv;
    n_last = gs->anchors_.size(0);
    if (it % 5 == 0 || it == 1) std::printf("iter %2d  loss %.5f  splats %lld  visible samples %lld\n", it, lv, (long long)n_last, (long long)ids.numel());
    if (!std::isfinite(lv)) { std::printf("non-finite loss\n"); return 1; }
  }
  bool finite = true;
  for (auto &p : gs->parameters()) finite = finite && torch::isfinite(torch::nan_to_num(p, 0.0, 0.0, -1e4)).all().item<bool>();
  for (auto &p : lm->parameters()) finite = finite && torch::isfinite(p).all().item<bool>();
  // the fit must have improved while only Adam moved the parameters, the refinement schedule must then have changed the splat set (every
  // fifth iteration doubles the high-gradient splats here: the loss of the freshly split set is not a criterion), the PLY must round-trip
  gs->export_gs_to_ply("/tmp/gsdf_model_demo.ply");
  const int64_t n_saved = gs->anchors_.size(0);
  gs->load_ply_to_gs("/tmp/gsdf_model_demo.ply");
  const bool ok = finite && before_refine < 0.9f * first && n_last != n_first && gs->anchors_.size(0) == n_saved;
  std::printf("first %.5f before refinement %.5f last %.5f  splats %lld -> %lld  finite %d  => %s\n", first, before_refine, last, (long long)n_first,
              (long long)n_last, (int)finite, ok ? "CPP PROGRAM OK" : "FAILED");
  return ok ? 0 : 1;
}
