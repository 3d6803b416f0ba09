// pytest_binding.cpp — pybind11 test harness: exposes the C++ operator layer (the functions the reference's
// neural_gaussian.cpp / local_map.cpp call) to the parity tests, so that the libtorch path is exercised exactly
// as the reference would call it.  Not part of the product surface.
#include <torch/extension.h>
#include <pybind11/functional.h>

#include "cumcubes/cumcubes_wrapper.h"
#include "gsdf_extras/gsdf_extras.h"
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
      .def("step", &gsdf_extras::FusedAdam::step);
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
  py::class_<gsdf_extras::JointIteration, std::shared_ptr<gsdf_extras::JointIteration>>(m, "JointIteration")
      .def(py::init([](const torch::Tensor &anchors, const std::vector<torch::Tensor> &fields, std::shared_ptr<TCNNEncoding> enc,
                       std::shared_ptr<TCNNNetwork> dec, std::vector<float> origin, double map_size, double bce_sigma, int occ_level, int width,
                       int height, int sh_degree, bool two_streams, bool analytic, bool reference_terms) {
        gsdf_extras::JointConfig cfg;
        cfg.width = width; cfg.height = height; cfg.sh_degree = sh_degree; cfg.two_streams = two_streams;
        cfg.analytic = analytic; cfg.reference_terms = reference_terms;
        return std::make_shared<gsdf_extras::JointIteration>(anchors, fields, enc, dec, origin, map_size, bce_sigma, occ_level, cfg);
      }), py::arg("anchors"), py::arg("fields"), py::arg("enc"), py::arg("dec"), py::arg("origin"), py::arg("map_size"), py::arg("bce_sigma"),
           py::arg("occ_level"), py::arg("width"), py::arg("height"), py::arg("sh_degree"), py::arg("two_streams"), py::arg("analytic") = true,
           py::arg("reference_terms") = true)
      .def("step", &gsdf_extras::JointIteration::step, py::arg("viewmat"), py::arg("K"), py::arg("target"), py::arg("ray_pts"),
           py::arg("ray_sdf"), py::arg("upstream"), py::arg("update") = true, py::arg("cam_host") = std::vector<float>(),
           py::call_guard<py::gil_scoped_release>())       // step() runs the autograd engine
      .def("sync", &gsdf_extras::JointIteration::sync)
      .def("set_grad_hooks", &gsdf_extras::JointIteration::set_grad_hooks)
      .def("splat_flat", &gsdf_extras::JointIteration::splat_flat)
      .def("splat_flat_grad", &gsdf_extras::JointIteration::splat_flat_grad)
      .def("sdf_flat", &gsdf_extras::JointIteration::sdf_flat)
      .def("sdf_flat_grad", &gsdf_extras::JointIteration::sdf_flat_grad)
      .def("nan_splats_seen", &gsdf_extras::JointIteration::nan_splats_seen);
  py::class_<TCNNEncoding, std::shared_ptr<TCNNEncoding>>(m, "TCNNEncoding")
      .def(py::init([](int n_levels, int n_feat, int log2_hashmap, int base_res, double pls) {
        nlohmann::json cfg = {{"otype", "Grid"}, {"type", "Hash"}, {"n_levels", n_levels}, {"n_features_per_level", n_feat},
                              {"log2_hashmap_size", log2_hashmap}, {"base_resolution", base_res}, {"per_level_scale", pls},
                              {"interpolation", "Linear"}};
        return std::make_shared<TCNNEncoding>(3, cfg, "encoder_test");
      }))
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
                                      torch::Tensor table_grad, torch::Tensor decoder_grad, py::object bias_grad) {
    auto t = [](const py::object &o) { return o.is_none() ? torch::Tensor() : o.cast<torch::Tensor>(); };
    return gsdf_extras::joint_sdf_loss_analytic(t(ray_xyz), t(gt), t(samples), t(ids), t(weights), *enc, *dec, origin, map_size_inv, bce_isigma,
                                                w_sdf, w_gs, delta, w_eik, w_align, table_grad, decoder_grad, t(bias_grad));
  });
  m.def("normal_consistency_loss", &gsdf_extras::normal_consistency_loss);
  m.def("isotropic_loss", &gsdf_extras::isotropic_loss);
}
