// pytest_binding.cpp — pybind11 test harness: exposes the C++ operator layer (the functions the reference's
// neural_gaussian.cpp / local_map.cpp call) to the parity tests, so that the libtorch path is exercised exactly
// as the reference would call it.  Not part of the product surface.
#include <torch/extension.h>
#include <pybind11/functional.h>

#include "cumcubes/cumcubes_wrapper.h"
#include "gsdf_extras/gsdf_extras.h"
#include "gsdf_model/gsdf_model.h"
#include "gsplat_cpp/fully_fused_projection.h"
#include "gsplat_cpp/rasterize_to_pixels.h"
#include "gsplat_cpp/rendering.h"
#include "kaolin/csrc/render/spc/raytrace.h"
#include "kaolin_wisp_cpp/octree_as/octree_as.h"
#include "kaolin_wisp_cpp/spc_ops/spc_ops.h"
#include "spatial.h"
#include "tcnn_binding/tcnn_binding.h"

namespace py = pybind11;

PYBIND11_MODULE(_gsdf_host, m) {
  m.def("fully_fused_projection_2dgs", &fully_fused_projection_2dgs);
  m.def("set_sample_mode", &gsplat_cpp::set_sample_mode);
  m.def("get_view_colors", [](const torch::Tensor &vm, const torch::Tensor &means, const torch::Tensor &radii,
                              const torch::Tensor &colors, const torch::Tensor &cam, const torch::Tensor &gid,
                              py::object deg) {
    return gsplat_cpp::get_view_colors(vm, means, radii, colors, cam, gid,
                                       deg.is_none() ? at::optional<int>() : at::optional<int>(deg.cast<int>()));
  });
  m.def("tile_encode", &gsplat_cpp::tile_encode);
  m.def("rasterize_to_pixels_2dgs",
        [](const torch::Tensor &a, const torch::Tensor &b, const torch::Tensor &c, const torch::Tensor &d,
           const torch::Tensor &e, const torch::Tensor &f, int w, int h, int t, const torch::Tensor &offs,
           const torch::Tensor &flat, py::object bg, py::object mk, bool packed, const torch::Tensor &absg, bool distloss) {
          return rasterize_to_pixels_2dgs(a, b, c, d, e, f, w, h, t, offs, flat,
                                          bg.is_none() ? at::optional<torch::Tensor>() : at::optional<torch::Tensor>(bg.cast<torch::Tensor>()),
                                          mk.is_none() ? at::optional<torch::Tensor>() : at::optional<torch::Tensor>(mk.cast<torch::Tensor>()),
                                          packed, absg, distloss);
        });
  m.def("distCUDA2", &distCUDA2);
  // fused extras (gsdf_extras/gsdf_extras.h)
  m.def("l1_dssim_loss", &gsdf_extras::l1_dssim_loss);
  m.def("query_points", &gsdf_extras::query_points);
  m.def("sdf_ray_loss", &gsdf_extras::sdf_ray_loss);
  m.def("gs_sdf_eik_loss", [](const torch::Tensor &attr, const torch::Tensor &w, py::object ids, int64_t n, double scale, double delta,
                              double w_eik) {
    return gsdf_extras::gs_sdf_eik_loss(attr, w, ids.is_none() ? torch::Tensor() : ids.cast<torch::Tensor>(), n, scale, delta, w_eik);
  });
  m.def("update_state", [](std::map<std::string, torch::Tensor> state, const torch::Tensor &g, const torch::Tensor &ids,
                           const torch::Tensor &vis, const torch::Tensor &radii, int64_t n, int c, int w, int h, bool want_radii) {
    gsdf_extras::update_state(state, g, ids, vis, radii, n, c, w, h, want_radii);
    return state;
  });
  m.def("splat_activations", &gsdf_extras::splat_activations);
  py::class_<gsdf_extras::FusedAdam>(m, "FusedAdam")
      .def(py::init<double, double, double>(), py::arg("beta1") = 0.9, py::arg("beta2") = 0.999, py::arg("eps") = 1e-15)
      .def("add_group", &gsdf_extras::FusedAdam::add_group)
      .def("set_lr", &gsdf_extras::FusedAdam::set_lr)
      .def("step", &gsdf_extras::FusedAdam::step, py::arg("zero_grad") = false)
      .def("step_tail", &gsdf_extras::FusedAdam::step_tail, py::arg("head_segments"), py::arg("zero_grad") = false)
      .def("step_head", &gsdf_extras::FusedAdam::step_head, py::arg("head_segments"), py::arg("zero_grad") = false);
  m.def("marching_cubes", [](const torch::Tensor &grid, float thresh, std::vector<float> lower, std::vector<float> upper) {
    TORCH_CHECK(lower.size() == 3 && upper.size() == 3, "marching_cubes: lower / upper need 3 entries");
    return mc::marching_cubes_wrapper(grid, thresh, lower.data(), upper.data());       // as cumcubes.cpp:9-27 calls it
  });
  m.def("quantize_points", &spc_ops::quantize_points);
  m.def("points_to_neighbors", &spc_ops::points_to_neighbors);
  m.def("points_to_corners", &spc_ops::points_to_corners);
  m.def("quantized_points_to_fpoints", &spc_ops::quantized_points_to_fpoints);
  m.def("mark_pack_boundaries", &kaolin::mark_pack_boundaries_cuda);
  py::class_<OctreeAS, std::shared_ptr<OctreeAS>>(m, "OctreeAS")
      .def_static("from_quantized_points", [](const torch::Tensor &q, int level) {
        return std::shared_ptr<OctreeAS>(from_quantized_points(q, level));          // as sub_map.cpp:33-34
      })
      .def("query", [](const OctreeAS &a, const torch::Tensor &x, int level) { return a.query(x, level).pidx; },
           py::arg("xyz"), py::arg("level") = -1)
      .def("raymarch", [](const OctreeAS &a, const torch::Tensor &o, const torch::Tensor &d, const std::string &t, int n) {
        auto r = a.raymarch(o, d, t, n);
        return std::make_tuple(r.ridx, r.samples, r.depth_samples);
      })
      .def("get_quantized_points", &OctreeAS::get_quantized_points)
      .def_readonly("grid_", &OctreeAS::grid_);
  m.def("gs_sdf_coupling", [](const torch::Tensor &samples, const torch::Tensor &ids, const torch::Tensor &weights,
                              std::shared_ptr<TCNNEncoding> enc, std::shared_ptr<TCNNNetwork> dec, std::vector<float> origin,
                              double map_size_inv, double scale, double delta, double w_eik, torch::Tensor table_grad,
                              torch::Tensor decoder_grad) {
    return gsdf_extras::gs_sdf_coupling(samples, ids, weights, *enc, *dec, origin, map_size_inv, scale, delta, w_eik, table_grad, decoder_grad);
  });
  // the defaults of gsdf_extras::JointConfig (tests/test_reference_config_defaults.py holds them to the reference's config/base.yaml)
  m.def("joint_config_defaults", [] {
    gsdf_extras::JointConfig c;
    py::dict d;
    d["near"] = c.near_plane; d["far"] = c.far_plane; d["rgb_weight"] = c.rgb_w; d["dssim_weight"] = c.dssim_w; d["eikonal_weight"] = c.eik_w;
    d["gs_sdf_weight"] = c.gs_sdf_w; d["visible_thr"] = c.vis_thresh; d["sdf_weight"] = c.sdf_w; d["align_weight"] = c.align_w;
    d["render_normal_weight"] = c.normal_w; d["isotropic_weight"] = c.isotropic_w; d["analytic"] = c.analytic; d["reference_terms"] = c.reference_terms;
    d["lr_end"] = c.lr_sdf;
    d["lrs"] = std::vector<double>{c.lr_offsets, c.lr_scaling, c.lr_quaternion, c.lr_opacity, c.lr_features_dc, c.lr_features_rest};
    return d;
  });
  py::class_<gsdf_extras::JointIteration, std::shared_ptr<gsdf_extras::JointIteration>>(m, "JointIteration")
      .def(py::init([](const torch::Tensor &anchors, const std::vector<torch::Tensor> &fields, std::shared_ptr<TCNNEncoding> enc,
                       std::shared_ptr<TCNNNetwork> dec, std::vector<float> origin, double map_size, double bce_sigma, int occ_level, int width,
                       int height, int sh_degree, bool two_streams, bool analytic, bool reference_terms, bool center_reg, int hashgrid_resident, int samples_grad_first) {
        gsdf_extras::JointConfig cfg;
        cfg.width = width; cfg.height = height; cfg.sh_degree = sh_degree; cfg.two_streams = two_streams;
        cfg.analytic = analytic; cfg.reference_terms = reference_terms; cfg.center_reg = center_reg;
        if (hashgrid_resident >= 0) cfg.hashgrid_resident = hashgrid_resident;
        if (samples_grad_first >= 0) cfg.samples_grad_first = samples_grad_first != 0;
        return std::make_shared<gsdf_extras::JointIteration>(anchors, fields, enc, dec, origin, map_size, bce_sigma, occ_level, cfg);
      }), py::arg("anchors"), py::arg("fields"), py::arg("enc"), py::arg("dec"), py::arg("origin"), py::arg("map_size"), py::arg("bce_sigma"),
           py::arg("occ_level"), py::arg("width"), py::arg("height"), py::arg("sh_degree"), py::arg("two_streams"), py::arg("analytic") = true,
           py::arg("reference_terms") = true, py::arg("center_reg") = true, py::arg("hashgrid_resident") = -1, py::arg("samples_grad_first") = -1)
      .def("step", &gsdf_extras::JointIteration::step, py::arg("viewmat"), py::arg("K"), py::arg("target"), py::arg("ray_pts"),
           py::arg("ray_sdf"), py::arg("upstream"), py::arg("update") = true, py::arg("cam_host") = std::vector<float>(),
           py::call_guard<py::gil_scoped_release>())       // step() runs the autograd engine
      .def("sync", &gsdf_extras::JointIteration::sync)
      .def("last_losses", &gsdf_extras::JointIteration::last_losses)
      .def("set_grad_hooks", &gsdf_extras::JointIteration::set_grad_hooks)
      .def("splat_flat", &gsdf_extras::JointIteration::splat_flat)
      .def("splat_flat_grad", &gsdf_extras::JointIteration::splat_flat_grad)
      .def("sdf_flat", &gsdf_extras::JointIteration::sdf_flat)
      .def("sdf_flat_grad", &gsdf_extras::JointIteration::sdf_flat_grad)
      .def("nan_splats_seen", &gsdf_extras::JointIteration::nan_splats_seen)
      .def("refine", &gsdf_extras::JointIteration::refine)
      .def("train_callback", &gsdf_extras::JointIteration::train_callback)
      .def("prune_rows", &gsdf_extras::JointIteration::prune_rows)
      .def("reset_opacity", &gsdf_extras::JointIteration::reset_opacity)
      .def("set_refine_hook", &gsdf_extras::JointIteration::set_refine_hook)
      .def("anchors", &gsdf_extras::JointIteration::anchors)
      .def("n_splats", &gsdf_extras::JointIteration::n_splats)
      .def("splat_adam_moments", &gsdf_extras::JointIteration::splat_adam_moments)
      .def("get_state", [](gsdf_extras::JointIteration &j) { return j.state(); })
      .def("set_state", [](gsdf_extras::JointIteration &j, std::map<std::string, torch::Tensor> s) { j.state() = std::move(s); })
      .def("set_splat_adam_moments", [](gsdf_extras::JointIteration &j, const torch::Tensor &m, const torch::Tensor &v) {
        auto mv = j.splat_adam_moments();
        torch::NoGradGuard ng;
        mv[0].copy_(m); mv[1].copy_(v);
      });
  py::class_<gsdf_extras::RefineConfig>(m, "RefineConfig")
      .def(py::init<>())
      .def_readwrite("prune_opa", &gsdf_extras::RefineConfig::prune_opa)
      .def_readwrite("grow_grad2d", &gsdf_extras::RefineConfig::grow_grad2d)
      .def_readwrite("grow_scale3d", &gsdf_extras::RefineConfig::grow_scale3d)
      .def_readwrite("grow_scale2d", &gsdf_extras::RefineConfig::grow_scale2d)
      .def_readwrite("prune_scale3d", &gsdf_extras::RefineConfig::prune_scale3d)
      .def_readwrite("refine_scale2d_stop_iter", &gsdf_extras::RefineConfig::refine_scale2d_stop_iter)
      .def_readwrite("refine_start_iter", &gsdf_extras::RefineConfig::refine_start_iter)
      .def_readwrite("refine_every", &gsdf_extras::RefineConfig::refine_every)
      .def_readwrite("reset_every", &gsdf_extras::RefineConfig::reset_every)
      .def_readwrite("pause_refine_after_reset", &gsdf_extras::RefineConfig::pause_refine_after_reset)
      .def_readwrite("spatial_scale", &gsdf_extras::RefineConfig::spatial_scale)
      .def_readwrite("original_spatial_scale", &gsdf_extras::RefineConfig::original_spatial_scale)
      .def_readwrite("num_train_data", &gsdf_extras::RefineConfig::num_train_data);
  py::class_<TCNNEncoding, std::shared_ptr<TCNNEncoding>>(m, "TCNNEncoding")
      .def(py::init([](int n_levels, int n_feat, int log2_hashmap, int base_res, double pls) {
        nlohmann::json cfg = {{"otype", "Grid"}, {"type", "Hash"}, {"n_levels", n_levels}, {"n_features_per_level", n_feat},
                              {"log2_hashmap_size", log2_hashmap}, {"base_resolution", base_res}, {"per_level_scale", pls},
                              {"interpolation", "Linear"}};
        return std::make_shared<TCNNEncoding>(3, cfg, "encoder_test");
      }))
      .def_static("spherical_harmonics", [](int degree) {      // what the reference's SHEncoding constructs (encodings.h:15-22)
        nlohmann::json cfg = {{"otype", "SphericalHarmonics"}, {"degree", degree}};
        return std::make_shared<TCNNEncoding>(3, cfg, "sh_encoding");
      })
      .def("forward", &TCNNEncoding::forward)
      .def("forward_stencil", &TCNNEncoding::forward_stencil)
      .def("get_out_dim", &TCNNEncoding::get_out_dim)
      .def_readwrite("params_", &TCNNEncoding::params_);
  py::class_<TCNNNetwork, std::shared_ptr<TCNNNetwork>>(m, "TCNNNetwork")
      .def(py::init([](int n_in, int n_out, int n_neurons, int n_hidden, bool bias) {
        nlohmann::json cfg = {{"otype", "FullyFusedMLP"}, {"activation", "ReLU"}, {"output_activation", "None"},
                              {"n_neurons", n_neurons}, {"n_hidden_layers", n_hidden}, {"bias", bias}};
        return std::make_shared<TCNNNetwork>(n_in, n_out, cfg, "decoder_test");
      }), py::arg("n_in"), py::arg("n_out"), py::arg("n_neurons"), py::arg("n_hidden"), py::arg("bias") = false)
      .def("forward", &TCNNNetwork::forward)
      .def_readwrite("params_", &TCNNNetwork::params_)
      .def_readwrite("biases_", &TCNNNetwork::biases_);
  m.def("joint_sdf_loss_analytic", [](py::object ray_xyz, py::object gt, py::object samples, py::object ids, py::object weights,
                                      std::shared_ptr<TCNNEncoding> enc, std::shared_ptr<TCNNNetwork> dec, std::vector<float> origin,
                                      double map_size_inv, double bce_isigma, double w_sdf, double w_gs, double delta, double w_eik, double w_align,
                                      torch::Tensor table_grad, torch::Tensor decoder_grad, py::object bias_grad, bool unit_upstream, bool first_order_in_forward) {
    auto t = [](const py::object &o) { return o.is_none() ? torch::Tensor() : o.cast<torch::Tensor>(); };
    return gsdf_extras::joint_sdf_loss_analytic(t(ray_xyz), t(gt), t(samples), t(ids), t(weights), *enc, *dec, origin, map_size_inv, bce_isigma,
                                                w_sdf, w_gs, delta, w_eik, w_align, table_grad, decoder_grad, t(bias_grad), nullptr, unit_upstream, first_order_in_forward);
  }, py::arg("ray_xyz"), py::arg("gt_sdf"), py::arg("samples"), py::arg("ids"), py::arg("weights"), py::arg("enc"), py::arg("dec"), py::arg("origin"),
     py::arg("map_size_inv"), py::arg("bce_isigma"), py::arg("w_sdf"), py::arg("w_gs"), py::arg("delta"), py::arg("w_eik"), py::arg("w_align"),
     py::arg("table_grad"), py::arg("decoder_grad"), py::arg("bias_grad"), py::arg("unit_upstream") = false, py::arg("first_order_in_forward") = false);
  m.def("normal_consistency_loss", &gsdf_extras::normal_consistency_loss);
  m.def("isotropic_loss", &gsdf_extras::isotropic_loss);
  m.def("render_post", &gsdf_extras::render_post);

  // ---- gsdf_model: the reference's model classes (tests/test_gpu_cpp_model.py compares them with the Python mirror) ----------------
  namespace gm = gsdf_model;
  auto opt_t = [](const py::object &o) { return o.is_none() ? torch::Tensor() : o.cast<torch::Tensor>(); };
  auto to_samples = [opt_t](const py::dict &d) {
    gm::DepthSamples s;
    auto g = [&](const char *k) { return d.contains(k) ? opt_t(d[k]) : torch::Tensor(); };
    s.origin = g("origin"); s.direction = g("direction"); s.depth = g("depth"); s.xyz = g("xyz"); s.ray_sdf = g("ray_sdf"); s.ridx = g("ridx");
    return s;
  };
  auto from_samples = [](const gm::DepthSamples &s) {
    py::dict d;
    auto p = [&](const char *k, const torch::Tensor &t) { if (t.defined()) d[k] = t; };
    p("origin", s.origin); p("direction", s.direction); p("depth", s.depth); p("xyz", s.xyz); p("ray_sdf", s.ray_sdf); p("ridx", s.ridx);
    return d;
  };
  py::class_<gm::MapConfig>(m, "MapConfig")
      .def(py::init<>())
      .def_readwrite("leaf_size", &gm::MapConfig::leaf_size)
      .def_readwrite("inner_map_size", &gm::MapConfig::inner_map_size)
      .def_readwrite("bce_sigma", &gm::MapConfig::bce_sigma)
      .def_readwrite("decoder_implementation", &gm::MapConfig::decoder_implementation)
      .def_readwrite("hidden_dim", &gm::MapConfig::hidden_dim)
      .def_readwrite("geo_num_layer", &gm::MapConfig::geo_num_layer)
      .def_readwrite("n_levels", &gm::MapConfig::n_levels)
      .def_readwrite("n_features_per_level", &gm::MapConfig::n_features_per_level)
      .def_readwrite("log2_hashmap_size", &gm::MapConfig::log2_hashmap_size)
      .def_readwrite("base_resolution", &gm::MapConfig::base_resolution)
      .def_readwrite("per_level_scale", &gm::MapConfig::per_level_scale)
      .def_readwrite("free_sample_num", &gm::MapConfig::free_sample_num)
      .def("octree_level", &gm::MapConfig::octree_level)
      .def("map_size", &gm::MapConfig::map_size);
  py::class_<gm::LocalMap, std::shared_ptr<gm::LocalMap>>(m, "LocalMap")
      .def(py::init([](const torch::Tensor &pos, const gm::MapConfig &cfg) { return std::make_shared<gm::LocalMap>(pos, cfg); }))
      .def_readonly("map_size_inv_", &gm::LocalMap::map_size_inv_)
      .def_readonly("pos_W_M_", &gm::LocalMap::pos_W_M_)
      .def_readonly("xyz_min_W_", &gm::LocalMap::xyz_min_W_)
      .def_readonly("xyz_max_W_", &gm::LocalMap::xyz_max_W_)
      .def_property_readonly("encoder", [](gm::LocalMap &l) { return l.p_encoder_tcnn_; })
      .def_property_readonly("decoder", [](gm::LocalMap &l) { return l.p_decoder_tcnn_; })
      .def("named_parameters", [](gm::LocalMap &l) {
        std::map<std::string, torch::Tensor> out;
        for (auto &kv : l.named_parameters()) out[kv.key()] = kv.value();
        return out;
      })
      .def("save_checkpoint", [](const std::shared_ptr<gm::LocalMap> &l, const std::string &path) { torch::save(l, path); })    // neural_mapping.cpp:1334
      .def("load_checkpoint", [](std::shared_ptr<gm::LocalMap> &l, const std::string &path) { torch::load(l, path); })          // :1351
      .def("update_octree_as", &gm::LocalMap::update_octree_as, py::arg("xyz"), py::arg("is_prior") = false)
      .def("get_inrange_mask", &gm::LocalMap::get_inrange_mask, py::arg("xyz"), py::arg("padding") = 0.f)
      .def("get_intersect_point", [](gm::LocalMap &l, const torch::Tensor &pts, const torch::Tensor &rays, float padding) {
        torch::Tensor a, b, c;
        l.get_intersect_point(pts, rays, a, b, c, padding);
        return std::make_tuple(a, b, c);
      }, py::arg("points"), py::arg("rays"), py::arg("padding") = 0.f)
      .def("get_valid_mask", &gm::LocalMap::get_valid_mask, py::arg("xyz"), py::arg("level") = -1)
      .def("xyz_to_m1p1_pts", &gm::LocalMap::xyz_to_m1p1_pts)
      .def("m1p1_pts_to_xyz", &gm::LocalMap::m1p1_pts_to_xyz)
      .def("xyz_to_zp1_pts", &gm::LocalMap::xyz_to_zp1_pts)
      .def("freeze_net", &gm::LocalMap::freeze_net)
      .def("unfreeze_net", &gm::LocalMap::unfreeze_net)
      .def("get_feat", &gm::LocalMap::get_feat, py::arg("xyz"), py::arg("encoding_type") = 0, py::arg("normalized") = false)
      .def("get_sdf", &gm::LocalMap::get_sdf)
      .def("get_gradient", [opt_t](gm::LocalMap &l, const torch::Tensor &xyz, float delta, py::object sdf, bool hessian, bool numerical) {
        torch::Tensor s = opt_t(sdf);
        py::gil_scoped_release no_gil;   // the analytic branch runs the autograd engine
        return l.get_gradient(xyz, delta, s, hessian, numerical);
      }, py::arg("xyz"), py::arg("delta") = 0.01f, py::arg("sdf") = py::none(), py::arg("hessian") = false, py::arg("numerical_grad") = true)
      .def("sample", [to_samples, from_samples](gm::LocalMap &l, const py::dict &s, int n, bool sample_free) {
        return from_samples(l.sample(to_samples(s), n, sample_free));
      }, py::arg("samples"), py::arg("voxel_sample_num") = 1, py::arg("sample_free") = true)
      .def("filter_sample", [to_samples, from_samples](gm::LocalMap &l, const py::dict &s) { return from_samples(l.filter_sample(to_samples(s))); });
  py::class_<gm::GSConfig>(m, "GSConfig")
      .def(py::init<>())
      .def_readwrite("sh_degree", &gm::GSConfig::sh_degree)
      .def_readwrite("near", &gm::GSConfig::near)
      .def_readwrite("far", &gm::GSConfig::far)
      .def_readwrite("use_absgrad", &gm::GSConfig::use_absgrad)
      .def_readwrite("center_reg", &gm::GSConfig::center_reg)
      .def_readwrite("geo_init", &gm::GSConfig::geo_init)
      .def_readwrite("detach_sdf_grad", &gm::GSConfig::detach_sdf_grad)
      .def_readwrite("prune_opa", &gm::GSConfig::prune_opa)
      .def_readwrite("grow_grad2d", &gm::GSConfig::grow_grad2d)
      .def_readwrite("grow_scale3d", &gm::GSConfig::grow_scale3d)
      .def_readwrite("grow_scale2d", &gm::GSConfig::grow_scale2d)
      .def_readwrite("prune_scale3d", &gm::GSConfig::prune_scale3d)
      .def_readwrite("refine_scale2d_stop_iter", &gm::GSConfig::refine_scale2d_stop_iter)
      .def_readwrite("refine_start_iter", &gm::GSConfig::refine_start_iter)
      .def_readwrite("refine_every", &gm::GSConfig::refine_every)
      .def_readwrite("reset_every", &gm::GSConfig::reset_every)
      .def_readwrite("sh_degree_interval", &gm::GSConfig::sh_degree_interval)
      .def_readwrite("pause_refine_after_reset", &gm::GSConfig::pause_refine_after_reset)
      .def_readwrite("lr_end", &gm::GSConfig::lr_end)
      .def_readwrite("vis_batch_pt_num", &gm::GSConfig::vis_batch_pt_num);
  // torch::optim::Adam over (SDF groups, splat groups) as NeuralSLAM builds it (neural_mapping.cpp:846-858)
  struct AdamBox { std::shared_ptr<torch::optim::Adam> p; };
  py::class_<AdamBox>(m, "Adam")
      .def("step", [](AdamBox &a) { a.p->step(); })
      .def("zero_grad", [](AdamBox &a) { a.p->zero_grad(); })
      .def("n_groups", [](AdamBox &a) { return a.p->param_groups().size(); })
      .def("lr", [](AdamBox &a, int g) { return a.p->param_groups().at(g).options().get_lr(); })
      .def("param", [](AdamBox &a, int g) { return a.p->param_groups().at(g).params().at(0); })
      .def("moments", [](AdamBox &a, int g) {
        auto &t = a.p->param_groups().at(g).params().at(0);
        auto it = a.p->state().find(t.unsafeGetTensorImpl());
        if (it == a.p->state().end()) return std::vector<torch::Tensor>{};
        auto &st = static_cast<torch::optim::AdamParamState &>(*it->second);
        return std::vector<torch::Tensor>{st.exp_avg(), st.exp_avg_sq()};
      });
  py::class_<gm::NeuralGS, std::shared_ptr<gm::NeuralGS>>(m, "NeuralGS")
      .def(py::init([](py::object lm, const torch::Tensor &anchors, const torch::Tensor &scaling, const torch::Tensor &quat, const torch::Tensor &opa,
                       const torch::Tensor &dc, const torch::Tensor &rest, int num_train_data, float spatial_scale, const gm::GSConfig &cfg) {
        auto l = lm.is_none() ? gm::LocalMap::Ptr() : lm.cast<gm::LocalMap::Ptr>();
        return std::make_shared<gm::NeuralGS>(l, anchors, scaling, quat, opa, dc, rest, num_train_data, spatial_scale, cfg);
      }))
      .def_static("from_points", [](py::object lm, const torch::Tensor &points, int num_train_data, float spatial_scale, bool sdf_enable, const gm::GSConfig &cfg) {
        auto l = lm.is_none() ? gm::LocalMap::Ptr() : lm.cast<gm::LocalMap::Ptr>();
        return std::make_shared<gm::NeuralGS>(l, points, num_train_data, spatial_scale, sdf_enable, cfg);
      })
      .def_readonly("anchors_", &gm::NeuralGS::anchors_)
      .def_readonly("offsets_", &gm::NeuralGS::offsets_)
      .def_readonly("scaling_", &gm::NeuralGS::scaling_)
      .def_readonly("quaternion_", &gm::NeuralGS::quaternion_)
      .def_readonly("opacity_", &gm::NeuralGS::opacity_)
      .def_readonly("features_dc_", &gm::NeuralGS::features_dc_)
      .def_readonly("features_rest_", &gm::NeuralGS::features_rest_)
      .def_readwrite("sh_degree_to_use_", &gm::NeuralGS::sh_degree_to_use_)
      .def_readwrite("gs_param_start_idx", &gm::NeuralGS::gs_param_start_idx)
      .def_readonly("spatial_scale_", &gm::NeuralGS::spatial_scale_)
      .def_readwrite("state", &gm::NeuralGS::state)
      .def("named_parameters", [](gm::NeuralGS &g) {
        std::map<std::string, torch::Tensor> out;
        for (auto &kv : g.named_parameters()) out[kv.key()] = kv.value();
        return out;
      })
      .def("get_xyz", &gm::NeuralGS::get_xyz)
      .def("get_scale", &gm::NeuralGS::get_scale)
      .def("get_opacity", &gm::NeuralGS::get_opacity, py::arg("training") = false)
      .def("make_optimizer", [](gm::NeuralGS &g, py::object lm, double sdf_lr) {
        std::vector<torch::optim::OptimizerParamGroup> groups;
        if (!lm.is_none()) {
          auto l = lm.cast<gm::LocalMap::Ptr>();
          for (auto &t : l->parameters()) {
            auto o = std::make_unique<torch::optim::AdamOptions>(sdf_lr);
            o->eps(1e-15);
            groups.emplace_back(std::vector<torch::Tensor>{t}, std::move(o));
          }
        }
        g.gs_param_start_idx = (int)groups.size();
        for (auto &grp : g.optimizer_params_groups_) groups.push_back(grp);
        AdamBox box;
        box.p = std::make_shared<torch::optim::Adam>(groups, torch::optim::AdamOptions(1e-3).eps(1e-15));
        return box;
      }, py::arg("local_map") = py::none(), py::arg("sdf_lr") = 1e-3)
      .def("render", [](gm::NeuralGS &g, const torch::Tensor &pose, float fx, float fy, float cx, float cy, int w, int h, bool training, int bck) {
        gm::Cameras cam;
        cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.width = w; cam.height = h;
        return g.render(pose, cam, training, bck);
      }, py::arg("pose_cam2world"), py::arg("fx"), py::arg("fy"), py::arg("cx"), py::arg("cy"), py::arg("width"), py::arg("height"),
           py::arg("training") = false, py::arg("bck_color") = 0)
      .def("train_callback", [](gm::NeuralGS &g, int iter, int total, AdamBox &a, std::map<std::string, torch::Tensor> info) {
        g.train_callback(iter, total, a.p, info);
      })
      .def("update_state", [](gm::NeuralGS &g, std::map<std::string, torch::Tensor> info) { g.update_state(info); })
      .def("grow_gs", [](gm::NeuralGS &g, int iter, AdamBox &a) { return g.grow_gs(iter, a.p); })
      .def("prune_gs", [](gm::NeuralGS &g, int iter, AdamBox &a, bool opa_only) { return g.prune_gs(iter, a.p, opa_only); },
           py::arg("iter"), py::arg("optimizer"), py::arg("prune_opa_only") = false)
      .def("prune_nan_gs", [](gm::NeuralGS &g, int iter, AdamBox &a) { return g.prune_nan_gs(iter, a.p); })
      .def("prune_invisible_gs", [](gm::NeuralGS &g, int iter, AdamBox &a) { return g.prune_invisible_gs(iter, a.p); })
      .def("reset_opacity", [](gm::NeuralGS &g, AdamBox &a) { g.reset_opacity(a.p); })
      .def("zero_state", &gm::NeuralGS::zero_state)
      .def("export_gs_to_ply", [](gm::NeuralGS &g, const std::string &p) { g.export_gs_to_ply(p); })
      .def("load_ply_to_gs", [](gm::NeuralGS &g, const std::string &p) { g.load_ply_to_gs(p); });
  m.def("sample_rays", [to_samples, from_samples](gm::LocalMap::Ptr lm, const py::dict &rays, float sample_std, float truncated_dis, int surface_sample_num,
                                                  bool sample_free) {
    return from_samples(gm::sample_rays(*lm, to_samples(rays), sample_std, truncated_dis, surface_sample_num, sample_free));
  });
  m.def("sample_rays_composed", [to_samples, from_samples](gm::LocalMap::Ptr lm, const py::dict &rays, float sample_std, float truncated_dis,
                                                           int surface_sample_num, bool sample_free) {
    return from_samples(gm::sample_rays_composed(*lm, to_samples(rays), sample_std, truncated_dis, surface_sample_num, sample_free));
  });
  m.def("init_gs_with_sdf", [](gm::LocalMap::Ptr lm, const torch::Tensor &xyz, float mesh_res, bool init_opa, int64_t batch) {
    return gm::init_gs_with_sdf(*lm, xyz, mesh_res, init_opa, batch);
  });
}
