// binding.cpp — TEST INFRASTRUCTURE (oracle/): a pybind11 module over the REFERENCE'S OWN host classes and functions, compiled from
// /root/reference/include where they lie (oracle/ref_link/build.py) and linked with this repository's drop-in operator libraries.
// Nothing here restates reference logic: every binding forwards to the reference's symbol.  What the tests do with it:
//   * CPU (tests/test_reference_intree_pins.py): the oracle's restatements of the losses, the SSIM window, the Adam state surgery and the
//     small geometry helpers are checked against the reference's compiled functions, and golden vectors are written from them
//     (tools/gen_reference_intree_golden.py -> tests/golden/reference_intree.npz) for machines without /root/reference;
//   * GPU (tests/test_gpu_reference_classes.py): the reference's LocalMap and NeuralGS run on this repository's kernels and are compared
//     with gsdf_model:: (the C++ classes this repository ships) — queries, sampling, render, gradients, the refinement schedule.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <torch/extension.h>

#include "neural_gaussian/neural_gaussian.h"
#include "neural_net/local_map.h"
#include "optimizer/loss.h"
#include "optimizer/loss_utils/loss_utils.h"
#include "optimizer/optimizer_utils/optimizer_utils.h"
#include "params/params.h"
#include "utils/utils.h"
#include "mesher/cumcubes/include/cumcubes.hpp"

namespace py = pybind11;
using torch::Tensor;

// neural_gaussian.cpp:19-21 (external linkage, no header declares it)
std::map<std::string, Tensor> init_gs_with_sdf(const LocalMap::Ptr &_local_map_ptr, const Tensor &xyzs, const float mesh_res, const bool &init_opa);

namespace {
struct AdamBox { std::shared_ptr<torch::optim::Adam> p; };

Tensor opt_t(const py::object &o) { return o.is_none() ? Tensor() : o.cast<Tensor>(); }

DepthSamples to_samples(const py::dict &d) {
  DepthSamples s;
  auto g = [&](const char *k) { return d.contains(k) ? opt_t(d[k]) : Tensor(); };
  s.origin = g("origin"); s.direction = g("direction"); s.depth = g("depth"); s.xyz = g("xyz"); s.ray_sdf = g("ray_sdf"); s.ridx = g("ridx");
  return s;
}

py::dict from_samples(const DepthSamples &s) {
  py::dict d;
  auto p = [&](const char *k, const Tensor &t) { if (t.defined()) d[k] = t; };
  p("origin", s.origin); p("direction", s.direction); p("depth", s.depth); p("xyz", s.xyz); p("ray_sdf", s.ray_sdf); p("ridx", s.ridx);
  return d;
}

py::dict defined_only(const std::map<std::string, Tensor> &m) {
  py::dict d;
  for (auto &kv : m)
    if (kv.second.defined()) d[py::str(kv.first)] = kv.second;
  return d;
}

// the configuration globals (params/params.h), by name
void configure(const py::dict &cfg) {
  for (auto item : cfg) {
    const std::string k = py::cast<std::string>(item.first);
    py::handle v = item.second;
#define GSDF_SET(name, type) if (k == #name) { k_##name = v.cast<type>(); continue; }
    if (k == "device") { k_device = torch::Device(v.cast<std::string>()); continue; }
    if (k == "output_path") { k_output_path = v.cast<std::string>(); continue; }
    if (k == "map_origin") { k_map_origin = v.cast<Tensor>(); continue; }
    GSDF_SET(dataset_type, int) GSDF_SET(decoder_implementation, int)
    GSDF_SET(x_max, float) GSDF_SET(x_min, float) GSDF_SET(y_max, float) GSDF_SET(y_min, float) GSDF_SET(z_max, float) GSDF_SET(z_min, float)
    GSDF_SET(inner_map_size, float) GSDF_SET(map_size, float) GSDF_SET(map_size_inv, float) GSDF_SET(leaf_size, float) GSDF_SET(octree_level, int)
    GSDF_SET(free_sample_num, int) GSDF_SET(hidden_dim, int) GSDF_SET(geo_num_layer, int) GSDF_SET(n_levels, int)
    GSDF_SET(n_features_per_level, int) GSDF_SET(log2_hashmap_size, int) GSDF_SET(bce_isigma, float) GSDF_SET(detach_sdf_grad, bool)
    GSDF_SET(numerical_grad, bool) GSDF_SET(lr_end, float) GSDF_SET(vis_attribute, int) GSDF_SET(vis_batch_pt_num, int) GSDF_SET(geo_init, bool)
    GSDF_SET(sky_init, bool) GSDF_SET(pause_refine, bool) GSDF_SET(near, float) GSDF_SET(far, float) GSDF_SET(prune_opa, float)
    GSDF_SET(grow_grad2d, float) GSDF_SET(grow_scale3d, float) GSDF_SET(grow_scale2d, float) GSDF_SET(prune_scale3d, float)
    GSDF_SET(refine_scale2d_stop_iter, int) GSDF_SET(refine_start_iter, int) GSDF_SET(refine_every, int) GSDF_SET(reset_every, int)
    GSDF_SET(use_absgrad, bool) GSDF_SET(sh_degree_interval, int) GSDF_SET(sh_degree, int) GSDF_SET(render_mode, bool) GSDF_SET(center_reg, bool)
    GSDF_SET(mesh_init, bool)
#undef GSDF_SET
    throw std::runtime_error("ref_configure: unknown parameter " + k);
  }
}
}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "the reference's in-tree host code, compiled where it lies and linked with this repository's drop-in operators (test infrastructure)";
  m.def("configure", &configure);

  // ---- optimizer/loss.h, optimizer/loss_utils/loss_utils.h
  m.def("gs_sdf_loss", [](const Tensor &s, const Tensor &w) { return loss::gs_sdf_loss(s, w); });
  m.def("gs_sdf_normal_loss", &loss::gs_sdf_normal_loss);
  m.def("rgb_loss", [](const Tensor &a, const Tensor &b, py::object mask) { return loss::rgb_loss(a, b, opt_t(mask)); }, py::arg("rgb"), py::arg("rgb_gt"),
        py::arg("mask") = py::none());
  m.def("distortion_loss", [](const Tensor &d) { return loss::distortion_loss(d); });
  m.def("dssim_loss", [](const Tensor &a, const Tensor &b, py::object mask) { return loss::dssim_loss(a, b, opt_t(mask)); }, py::arg("pred"), py::arg("gt"),
        py::arg("mask") = py::none());   // moves its window to the GPU: GPU box only
  m.def("sdf_loss", &loss::sdf_loss);
  m.def("eikonal_loss", [](const Tensor &g) { return loss::eikonal_loss(g); });
  m.def("curvate_loss", [](const Tensor &h) { return loss::curvate_loss(h); });
  m.def("gaussian", &loss_utils::gaussian);
  m.def("create_window", &loss_utils::create_window);
  m.def("_ssim", &loss_utils::_ssim, py::arg("img1"), py::arg("img2"), py::arg("window"), py::arg("window_size"), py::arg("channel"),
        py::arg("size_average") = true);
  m.def("ssim", &loss_utils::ssim, py::arg("img1"), py::arg("img2"), py::arg("window_size") = 11, py::arg("channel") = 3);   // GPU box only
  m.def("psnr", &loss_utils::psnr);

  // ---- utils/utils.h: the helpers the model classes use
  m.def("rotation_6d_to_matrix", &utils::rotation_6d_to_matrix);
  m.def("normalized_quat_to_rotmat", &utils::normalized_quat_to_rotmat);
  m.def("meshgrid_3d", [](float x0, float x1, float y0, float y1, float z0, float z1, float res, const std::string &dev) {
    torch::Device d(dev);
    return utils::meshgrid_3d(x0, x1, y0, y1, z0, z1, res, d);
  });
  m.def("sample_free_pts", [](const py::dict &s, int n) { return from_samples(utils::sample_free_pts(to_samples(s), n)); });
  m.def("sample_surface_pts", [](const py::dict &s, int n, float std) { return from_samples(utils::sample_surface_pts(to_samples(s), n, std)); });
  // utils/sensor_utils/cameras.hpp:176-226 (header-only; caches the pixel directions of the FIRST camera it sees: one image size per process)
  m.def("depth_to_normal", [](float fx, float fy, float cx, float cy, int w, int h, const Tensor &pose, const Tensor &depth) {
    sensor::Cameras cam;
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.width = w; cam.height = h;
    return sensor::depth_to_normal(cam, pose, depth);
  });

  // ---- mesher/cumcubes/include/cumcubes.hpp: the mesh file writer (host code of the reference; the kernels behind mc::marching_cubes are
  //      this repository's, reached through LocalMap.meshing_)
  m.def("save_mesh_as_ply", [](const std::string &path, Tensor v, Tensor f, Tensor c) { mc::save_mesh_as_ply(path, v, f, c); });

  // ---- optimizer/optimizer_utils/optimizer_utils.h on a torch::optim::Adam
  py::class_<AdamBox>(m, "Adam")
      .def(py::init([](std::vector<Tensor> params, std::vector<double> lrs, double eps) {
        std::vector<torch::optim::OptimizerParamGroup> groups;
        for (size_t i = 0; i < params.size(); ++i) {
          auto o = std::make_unique<torch::optim::AdamOptions>(lrs.at(i));
          o->eps(eps);
          groups.emplace_back(std::vector<Tensor>{params[i]}, std::move(o));
        }
        AdamBox b;
        b.p = std::make_shared<torch::optim::Adam>(groups, torch::optim::AdamOptions(1e-3).eps(eps));
        return b;
      }), py::arg("params"), py::arg("lrs"), py::arg("eps") = 1e-15)
      .def("step", [](AdamBox &a) { a.p->step(); })
      .def("zero_grad", [](AdamBox &a) { a.p->zero_grad(); })
      .def("n_groups", [](AdamBox &a) { return a.p->param_groups().size(); })
      .def("lr", [](AdamBox &a, int g) { return a.p->param_groups().at(g).options().get_lr(); })
      .def("param", [](AdamBox &a, int g) { return a.p->param_groups().at(g).params().at(0); })
      .def("moments", [](AdamBox &a, int g) {
        auto &t = a.p->param_groups().at(g).params().at(0);
        auto it = a.p->state().find(t.unsafeGetTensorImpl());
        if (it == a.p->state().end()) return std::vector<Tensor>{};
        auto &st = static_cast<torch::optim::AdamParamState &>(*it->second);
        return std::vector<Tensor>{st.exp_avg(), st.exp_avg_sq()};
      });
  m.def("prune_optimizer", [](AdamBox &a, const Tensor &mask, Tensor old, int pos) { prune_optimizer(a.p.get(), mask, old, pos); return old; });
  m.def("cat_tensors_to_optimizer", [](AdamBox &a, const Tensor &ext, Tensor old, int pos) { cat_tensors_to_optimizer(a.p.get(), ext, old, pos); return old; });
  m.def("prune_cat_tensors_to_optimizer", [](AdamBox &a, Tensor old, const Tensor &rest, const Tensor &ext, int pos) {
    prune_cat_tensors_to_optimizer(a.p.get(), old, rest, ext, pos);
    return old;
  });
  m.def("replace_tensors_to_optimizer", [](AdamBox &a, Tensor old, Tensor fresh, int pos) { replace_tensors_to_optimizer(a.p.get(), old, fresh, pos); return old; });

  // ---- neural_net/local_map.h
  py::class_<LocalMap, std::shared_ptr<LocalMap>>(m, "LocalMap")
      .def(py::init([](const Tensor &pos) { return std::make_shared<LocalMap>(pos, k_x_min, k_x_max, k_y_min, k_y_max, k_z_min, k_z_max); }))
      .def_readonly("pos_W_M_", &LocalMap::pos_W_M_)
      .def_readonly("xyz_min_W_", &LocalMap::xyz_min_W_)
      .def_readonly("xyz_max_W_", &LocalMap::xyz_max_W_)
      .def("named_parameters", [](LocalMap &l) {
        std::map<std::string, Tensor> out;
        for (auto &kv : l.named_parameters()) out[kv.key()] = kv.value();
        return out;
      })
      .def("update_octree_as", [](LocalMap &l, const Tensor &xyz, bool prior) { l.update_octree_as(xyz, prior); }, py::arg("xyz"), py::arg("is_prior") = false)
      .def("get_inrange_mask", &LocalMap::get_inrange_mask, py::arg("xyz"), py::arg("padding") = 0.f)
      .def("get_intersect_point", [](LocalMap &l, const Tensor &pts, const Tensor &rays, float padding) {
        Tensor a, b, c;
        l.get_intersect_point(pts, rays, a, b, c, padding);
        return std::make_tuple(a, b, c);
      }, py::arg("points"), py::arg("rays"), py::arg("padding") = 0.f)
      .def("get_valid_mask", &LocalMap::get_valid_mask, py::arg("xyz"), py::arg("level") = -1)
      .def("xyz_to_m1p1_pts", &LocalMap::xyz_to_m1p1_pts)
      .def("xyz_to_zp1_pts", &LocalMap::xyz_to_zp1_pts)
      .def("get_feat", &LocalMap::get_feat, py::arg("xyz"), py::arg("encoding_type") = 0, py::arg("normalized") = false)
      .def("get_sdf", &LocalMap::get_sdf)
      .def("get_gradient", [](LocalMap &l, const Tensor &xyz, float delta, py::object sdf, bool hessian, bool numerical) {
        Tensor s = opt_t(sdf);
        py::gil_scoped_release no_gil;   // the analytic branch runs the autograd engine
        return l.get_gradient(xyz, delta, s, hessian, numerical);
      }, py::arg("xyz"), py::arg("delta") = 0.01f, py::arg("sdf") = py::none(), py::arg("hessian") = false, py::arg("numerical_grad") = true)
      .def("sample", [](LocalMap &l, const py::dict &s, int n, bool sample_free) { return from_samples(l.sample(to_samples(s), n, sample_free)); },
           py::arg("samples"), py::arg("voxel_sample_num") = 1, py::arg("sample_free") = true)
      .def("filter_sample", [](LocalMap &l, const py::dict &s) {
        DepthSamples in = to_samples(s);
        return from_samples(l.filter_sample(in));
      })
      .def("meshing_", [](LocalMap &l, float res) {   // local_map.cpp:329-447 with _save: the chunks' vertices / faces / colours land in the mesher
        l.meshing_(res, true);
        return std::make_tuple(l.p_mesher_->vec_vertice_, l.p_mesher_->vec_face_, l.p_mesher_->vec_face_attr_);
      });

  // NeuralSLAM::export_checkpoint / load_checkpoint (neural_mapping.cpp:1331-1352): torch::save / torch::load of the LocalMap module
  m.def("save_local_map", [](LocalMap::Ptr lm, const std::string &path) { torch::save(lm, path); });
  m.def("load_local_map", [](LocalMap::Ptr lm, const std::string &path) { torch::load(lm, path); });

  // ---- neural_gaussian/neural_gaussian.h
  py::class_<NeuralGS, std::shared_ptr<NeuralGS>>(m, "NeuralGS")
      .def(py::init([](py::object lm, const Tensor &points, int num_train_data, float spatial_scale, bool sdf_enable) {
        auto l = lm.is_none() ? LocalMap::Ptr() : lm.cast<LocalMap::Ptr>();
        return std::make_shared<NeuralGS>(l, points, num_train_data, spatial_scale, sdf_enable);
      }))
      // NeuralGS(local_map, gs.ply) (neural_gaussian.cpp:456-461): the reference's own loader; no kernel involved, runs on k_device = cpu too
      .def_static("from_ply", [](py::object lm, const std::string &path) {
        auto l = lm.is_none() ? LocalMap::Ptr() : lm.cast<LocalMap::Ptr>();
        return std::make_shared<NeuralGS>(l, std::filesystem::path(path));
      })
      .def("export_gs_to_ply", [](NeuralGS &g, const std::string &path) {
        std::filesystem::path p(path);
        g.export_gs_to_ply(p);
      })
      .def("load_ply_to_gs", [](NeuralGS &g, const std::string &path) { g.load_ply_to_gs(path); })
      .def_readwrite("anchors_", &NeuralGS::anchors_)
      .def_readwrite("offsets_", &NeuralGS::offsets_)
      .def_readwrite("scaling_", &NeuralGS::scaling_)
      .def_readwrite("quaternion_", &NeuralGS::quaternion_)
      .def_readwrite("opacity_", &NeuralGS::opacity_)
      .def_readwrite("features_dc_", &NeuralGS::features_dc_)
      .def_readwrite("features_rest_", &NeuralGS::features_rest_)
      .def_property("sh_degree_to_use_", [](NeuralGS &g) { return g.sh_degree_to_use_.has_value() ? py::cast(*g.sh_degree_to_use_) : py::object(py::none()); },
                    [](NeuralGS &g, int d) { g.sh_degree_to_use_ = d; })
      .def_readwrite("gs_param_start_idx", &NeuralGS::gs_param_start_idx)
      .def_readonly("spatial_scale_", &NeuralGS::spatial_scale_)
      .def_readwrite("pause_refine_after_reset", &NeuralGS::pause_refine_after_reset)
      .def_property_readonly("state", [](NeuralGS &g) { return defined_only(g.state); })
      .def("named_parameters", [](NeuralGS &g) {
        std::map<std::string, Tensor> out;
        for (auto &kv : g.named_parameters()) out[kv.key()] = kv.value();
        return out;
      })
      .def("get_xyz", &NeuralGS::get_xyz)
      .def("get_scale", &NeuralGS::get_scale)
      .def("get_opacity", &NeuralGS::get_opacity, py::arg("training") = false)
      .def("make_optimizer", [](NeuralGS &g, py::object lm, double sdf_lr) {   // as neural_mapping.cpp:846-858 assembles it
        std::vector<torch::optim::OptimizerParamGroup> groups;
        if (!lm.is_none()) {
          auto l = lm.cast<LocalMap::Ptr>();
          for (auto &t : l->parameters()) {
            auto o = std::make_unique<torch::optim::AdamOptions>(sdf_lr);
            o->eps(1e-15);
            groups.emplace_back(std::vector<Tensor>{t}, std::move(o));
          }
        }
        g.gs_param_start_idx = (int)groups.size();
        for (auto &grp : g.optimizer_params_groups_) groups.push_back(grp);
        AdamBox box;
        box.p = std::make_shared<torch::optim::Adam>(groups, torch::optim::AdamOptions(1e-3).eps(1e-15));
        return box;
      }, py::arg("local_map") = py::none(), py::arg("sdf_lr") = 1e-3)
      .def("render", [](NeuralGS &g, const Tensor &pose, float fx, float fy, float cx, float cy, int w, int h, bool training, int bck) {
        sensor::Cameras cam;
        cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.width = w; cam.height = h;
        return defined_only(g.render(pose, cam, training, bck));
      }, py::arg("pose_cam2world"), py::arg("fx"), py::arg("fy"), py::arg("cx"), py::arg("cy"), py::arg("width"), py::arg("height"),
           py::arg("training") = false, py::arg("bck_color") = 0)
      .def("train_callback", [](NeuralGS &g, int iter, int total, AdamBox &a, std::map<std::string, Tensor> info) { g.train_callback(iter, total, a.p, info); });
  m.def("init_gs_with_sdf", [](LocalMap::Ptr lm, const Tensor &xyz, float mesh_res, bool init_opa) { return defined_only(init_gs_with_sdf(lm, xyz, mesh_res, init_opa)); });
}
