// gsplat_ops.cpp — libtorch operator layer over the C ABI for the splat path: the functions the reference's
// host code calls into gsplat_cpp (/root/reference/include/neural_gaussian/neural_gaussian.cpp:188-223),
// each a torch::autograd::Function whose forward/backward are single calls into libgsdf_hip.so.
#include <limits>

#include "gsplat_cpp/fully_fused_projection.h"
#include "gsplat_cpp/rasterize_to_pixels.h"
#include "gsplat_cpp/rendering.h"
#include "util.h"

using namespace gsdf_host;
using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

namespace {
// The fork's projection always returns its stochastic samples; the reference replaces them by the centres afterwards when k_center_reg is
// set (neural_gaussian.cpp:258-264).  So that the UNMODIFIED reference gets what it expects, stochastic is the library default; callers
// that know they will discard the samples (gsdf_model::rasterization_2dgs_sdf, gsdf_extras::JointIteration in centre mode, the Python test
// harness gs_sdf_amd.hostlib) switch the draw off for their scope.
thread_local bool g_stochastic_samples = true;   // ambient mode of the reference-signature entry point only, per thread

// a binding built against another header version must not reach the device with shifted arguments
const bool g_abi_checked = [] {
  if (gsdf_abi_version() != GSDF_ABI_VERSION) {
    fprintf(stderr, "libgsdf_torch: libgsdf_hip.so has ABI version %d, this layer was built against %d (include/gsdf_hip.h): rebuild both\n",
            gsdf_abi_version(), GSDF_ABI_VERSION);
    abort();
  }
  return true;
}();

uint64_t next_sample_seed_impl(bool stochastic) {
  if (!stochastic) return 0;
  // GSDF_SAMPLE_SEED (tests): a fixed seed instead of a draw
  static const uint64_t fixed = [] { const char *e = getenv("GSDF_SAMPLE_SEED"); return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)0; }();
  if (fixed) return fixed;
  // one draw from torch's default CPU generator: reproducible under torch::manual_seed (neural_mapping_node.cpp:26)
  const uint64_t s = (uint64_t)torch::randint(1, std::numeric_limits<int64_t>::max(), {1}, torch::kInt64).item<int64_t>();
  return s ? s : 1;
}

// ------------------------------------------------------------------------------------------ P1
struct Projection2DGS : public torch::autograd::Function<Projection2DGS> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &means_, const Tensor &quats_, const Tensor &scales_,
                             const Tensor &viewmats_, const Tensor &Ks_, int64_t width, int64_t height, double near_plane,
                             double far_plane, double radius_clip, int64_t seed) {
    Tensor means = f32c(means_, "means"), quats = f32c(quats_, "quats"), scales = f32c(scales_, "scales");
    Tensor viewmats = f32c(viewmats_, "viewmats"), Ks = f32c(Ks_, "Ks");
    const int64_t N = means.size(0), C = viewmats.size(0);
    Tensor radii_dense = empty_like_opts(means, {std::max<int64_t>(N * C, 1)}, torch::kInt32);
    Tensor ws = empty_like_opts(means, {(int64_t)gsdf_projection_2dgs_ws_bytes(N, C)}, torch::kUInt8);
    const int64_t M = gsdf_host::count_via_host_word(means, [&](int64_t *n_vis) {
      check(gsdf_projection_2dgs_cull(N, C, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (int)width, (int)height,
                                      (float)near_plane, (float)far_plane, (float)radius_clip, radii_dense.data_ptr<int32_t>(),
                                      ws.data_ptr(), n_vis, cur_stream()),
            "fully_fused_projection_2dgs(cull)");
    }, N * C);
    Tensor camera_ids = empty_like_opts(means, {M}, torch::kInt64), gaussian_ids = empty_like_opts(means, {M}, torch::kInt64);
    Tensor radii = empty_like_opts(means, {M}, torch::kInt32), means2d = empty_like_opts(means, {M, 2}, torch::kFloat32);
    Tensor depths = empty_like_opts(means, {M}, torch::kFloat32), rt = empty_like_opts(means, {M, 3, 3}, torch::kFloat32);
    Tensor normals = empty_like_opts(means, {M, 3}, torch::kFloat32), samples = empty_like_opts(means, {M, 3}, torch::kFloat32);
    Tensor sw = empty_like_opts(means, {M, 1}, torch::kFloat32);
    check(gsdf_projection_2dgs_fill(N, C, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks), (int)width, (int)height,
                                    (uint64_t)seed, radii_dense.data_ptr<int32_t>(), ws.data_ptr(), M,
                                    M ? camera_ids.data_ptr<int64_t>() : nullptr, M ? gaussian_ids.data_ptr<int64_t>() : nullptr,
                                    M ? radii.data_ptr<int32_t>() : nullptr, fpm(means2d), fpm(depths), fpm(rt), fpm(normals),
                                    fpm(samples), fpm(sw), cur_stream()),
          "fully_fused_projection_2dgs(fill)");
    ctx->save_for_backward({means, quats, scales, viewmats, Ks, camera_ids, gaussian_ids});
    ctx->saved_data["width"] = width;
    ctx->saved_data["height"] = height;
    ctx->saved_data["seed"] = seed;
    ctx->mark_non_differentiable({camera_ids, gaussian_ids, radii, sw});
    return {camera_ids, gaussian_ids, radii, means2d, depths, rt, normals, samples, sw};
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &means = saved[0], &quats = saved[1], &scales = saved[2], &viewmats = saved[3], &Ks = saved[4];
    const Tensor &camera_ids = saved[5], &gaussian_ids = saved[6];
    const int64_t N = means.size(0), C = viewmats.size(0), M = camera_ids.size(0);
    auto z = [&](const Tensor &t, at::IntArrayRef shape) { return t.defined() ? f32c(t, "grad") : zeros_like_opts(means, shape, torch::kFloat32); };
    Tensor v_means2d = z(g[3], {M, 2}), v_depths = z(g[4], {M}), v_rt = z(g[5], {M, 3, 3}), v_normals = z(g[6], {M, 3});
    Tensor v_samples = g[7].defined() ? f32c(g[7], "grad") : Tensor();
    Tensor v_means = torch::zeros_like(means), v_quats = torch::zeros_like(quats), v_scales = torch::zeros_like(scales);
    check(gsdf_projection_2dgs_bwd(N, C, M, fp(means), fp(quats), fp(scales), fp(viewmats), fp(Ks),
                                   (int)ctx->saved_data["width"].toInt(), (int)ctx->saved_data["height"].toInt(),
                                   (uint64_t)ctx->saved_data["seed"].toInt(), M ? camera_ids.data_ptr<int64_t>() : nullptr,
                                   M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fp(v_means2d), fp(v_depths), fp(v_rt),
                                   fp(v_normals), fp(v_samples), fpm(v_means), fpm(v_quats), fpm(v_scales), cur_stream()),
          "fully_fused_projection_2dgs(backward)");
    return {v_means, v_quats, v_scales, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------ P2
struct ViewColors : public torch::autograd::Function<ViewColors> {
  static Tensor forward(AutogradContext *ctx, const Tensor &viewmats_, const Tensor &means_, const Tensor &sh_,
                        const Tensor &camera_ids, const Tensor &gaussian_ids, int64_t sh_degree) {
    Tensor viewmats = f32c(viewmats_, "viewmats"), means = f32c(means_, "means"), sh = f32c(sh_, "colors");
    const int64_t M = camera_ids.size(0), K = sh.size(1);
    Tensor colors = empty_like_opts(means, {M, 3}, torch::kFloat32);
    check(gsdf_view_colors_fwd(M, K, (int)sh_degree, fp(viewmats), fp(means), fp(sh), M ? camera_ids.data_ptr<int64_t>() : nullptr,
                               M ? gaussian_ids.data_ptr<int64_t>() : nullptr, fpm(colors), cur_stream()),
          "get_view_colors");
    ctx->save_for_backward({viewmats, means, sh, camera_ids, gaussian_ids});
    ctx->saved_data["deg"] = sh_degree;
    return colors;
  }
  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const int64_t M = s[3].size(0), K = s[2].size(1);
    Tensor v_sh = torch::zeros_like(s[2]), v_means = torch::zeros_like(s[1]);
    Tensor vc = f32c(g[0], "grad");
    check(gsdf_view_colors_bwd(M, K, (int)ctx->saved_data["deg"].toInt(), fp(s[0]), fp(s[1]), fp(s[2]),
                               M ? s[3].data_ptr<int64_t>() : nullptr, M ? s[4].data_ptr<int64_t>() : nullptr, fp(vc),
                               fpm(v_sh), fpm(v_means), s[0].size(0) == 1 ? 1 : 0, cur_stream()),
          "get_view_colors(backward)");
    return {Tensor(), v_means, v_sh, Tensor(), Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------ P4
struct Rasterize2DGS : public torch::autograd::Function<Rasterize2DGS> {
  static tensor_list forward(AutogradContext *ctx, const Tensor &means2d_, const Tensor &rt_, const Tensor &colors_,
                             const Tensor &opacities_, const Tensor &normals_, const Tensor &densify,
                             const Tensor &means2d_absgrad, const at::optional<Tensor> &backgrounds,
                             const at::optional<Tensor> &masks, int64_t width, int64_t height, int64_t tile_size,
                             const Tensor &isect_offsets, const Tensor &flatten_ids) {
    Tensor means2d = f32c(means2d_, "means2d"), rt = f32c(rt_, "ray_transforms"), colors = f32c(colors_, "colors");
    Tensor opac = f32c(opacities_, "opacities"), normals = f32c(normals_, "normals");
    const int64_t C = isect_offsets.size(0), M = opac.size(0), I = flatten_ids.size(0);
    Tensor bg = backgrounds.has_value() ? f32c(backgrounds.value(), "backgrounds") : Tensor();
    Tensor mk = masks.has_value() ? masks.value().to(torch::kUInt8).contiguous() : Tensor();
    Tensor offs = isect_offsets.contiguous(), flat = flatten_ids.contiguous();
    auto e = [&](at::IntArrayRef s) { return empty_like_opts(means2d, s, torch::kFloat32); };
    Tensor rc = e({C, height, width, 3}), rd = e({C, height, width, 1}), ra = e({C, height, width, 1});
    Tensor rn = e({C, height, width, 3}), rm = e({C, height, width, 1}), vis = e({M, 1});
    Tensor last = empty_like_opts(means2d, {C, height, width}, torch::kInt32), med = empty_like_opts(means2d, {C, height, width}, torch::kInt32);
    // the transmittance each pixel ended with, saved for the backward (render_alphas = 1 - T cannot give it back once T << 1)
    Tensor fT = e({C, height, width});
    // packed splat records + reach masks of the (tile, splat) pairs (csrc/raster_quad.h): written by the forward, reused by the backward
    Tensor fws = empty_like_opts(means2d, {(int64_t)gsdf_rasterize_2dgs_fwd_ws_bytes(M, I)}, torch::kUInt8);
    check(gsdf_rasterize_2dgs_fwd(C, M, I, (int)width, (int)height, (int)tile_size, fp(means2d), fp(rt), fp(colors), fp(opac),
                                  fp(normals), fp(bg), mk.defined() ? mk.data_ptr<uint8_t>() : nullptr,
                                  offs.data_ptr<int32_t>(), I ? flat.data_ptr<int32_t>() : nullptr, fpm(rc), fpm(rd), fpm(ra),
                                  fpm(rn), fpm(rm), last.data_ptr<int32_t>(), med.data_ptr<int32_t>(), fpm(vis), fpm(fT), fws.data_ptr(),
                                  cur_stream()),
          "rasterize_to_pixels_2dgs");
    ctx->save_for_backward({means2d, rt, colors, opac, normals, bg, mk, offs, flat, ra, last, med, fT, fws});
    ctx->saved_data["w"] = width; ctx->saved_data["h"] = height; ctx->saved_data["t"] = tile_size;
    ctx->saved_data["absgrad"] = means2d_absgrad.requires_grad();
    Tensor distort = zeros_like_opts(means2d, {C, height, width, 1}, torch::kFloat32);
    ctx->mark_non_differentiable({vis, distort});
    (void)densify;
    return {rc, rd, ra, rn, distort, rm, vis};
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list g) {
    auto s = ctx->get_saved_variables();
    const Tensor &means2d = s[0], &rt = s[1], &colors = s[2], &opac = s[3], &normals = s[4], &bg = s[5], &mk = s[6];
    const Tensor &offs = s[7], &flat = s[8], &ra = s[9], &last = s[10], &med = s[11], &fT = s[12], &fws = s[13];
    const int64_t width = ctx->saved_data["w"].toInt(), height = ctx->saved_data["h"].toInt(), tile = ctx->saved_data["t"].toInt();
    const int64_t C = offs.size(0), M = opac.size(0), I = flat.size(0);
    auto z = [&](const Tensor &t, int64_t ch) {
      return t.defined() ? f32c(t, "grad") : zeros_like_opts(means2d, {C, height, width, ch}, torch::kFloat32);
    };
    Tensor v_rc = z(g[0], 3), v_rd = z(g[1], 1), v_ra = z(g[2], 1), v_rn = z(g[3], 3), v_rm = z(g[5], 1);
    auto e = [&](at::IntArrayRef sh) { return empty_like_opts(means2d, sh, torch::kFloat32); };
    Tensor v_means2d = e({M, 2}), v_rt = e({M, 3, 3}), v_colors = e({M, 3}), v_opac = e({M}), v_normals = e({M, 3}), v_dens = e({M, 2});
    Tensor v_abs = ctx->saved_data["absgrad"].toBool() ? e({M, 2}) : Tensor();
    Tensor ws = empty_like_opts(means2d, {(int64_t)gsdf_rasterize_2dgs_bwd_ws_bytes(M, I)}, torch::kUInt8);
    check(gsdf_rasterize_2dgs_bwd(C, M, I, (int)width, (int)height, (int)tile, fp(means2d), fp(rt), fp(colors), fp(opac),
                                  fp(normals), fp(bg), mk.defined() ? mk.data_ptr<uint8_t>() : nullptr, offs.data_ptr<int32_t>(),
                                  I ? flat.data_ptr<int32_t>() : nullptr, fp(ra), last.data_ptr<int32_t>(), med.data_ptr<int32_t>(),
                                  fp(v_rc), fp(v_rd), fp(v_ra), fp(v_rn), fp(v_rm), fpm(v_means2d), fpm(v_rt), fpm(v_colors),
                                  fpm(v_opac), fpm(v_normals), fpm(v_dens), fpm(v_abs), ws.data_ptr(), fp(fT), fws.data_ptr(), cur_stream()),
          "rasterize_to_pixels_2dgs(backward)");
    return {v_means2d, v_rt, v_colors, v_opac, v_normals, v_dens, v_abs, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
}  // namespace

void gsplat_cpp::set_sample_mode(bool stochastic) { g_stochastic_samples = stochastic; }
bool gsplat_cpp::get_sample_mode() { return g_stochastic_samples; }
uint64_t gsplat_cpp::next_sample_seed() { return next_sample_seed_impl(g_stochastic_samples); }
uint64_t gsplat_cpp::next_sample_seed(bool stochastic) { return next_sample_seed_impl(stochastic); }

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
fully_fused_projection_2dgs(const Tensor &means, const Tensor &quats, const Tensor &scales, const Tensor &viewmats,
                            const Tensor &Ks, int width, int height, float near_plane, float far_plane, float radius_clip,
                            bool packed, bool sparse_grad) {
  return gsplat_cpp::fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, width, height, near_plane, far_plane, radius_clip, packed,
                                                 sparse_grad, g_stochastic_samples);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
gsplat_cpp::fully_fused_projection_2dgs(const Tensor &means, const Tensor &quats, const Tensor &scales, const Tensor &viewmats,
                                        const Tensor &Ks, int width, int height, float near_plane, float far_plane, float radius_clip,
                                        bool packed, bool sparse_grad, bool stochastic_samples) {
  TORCH_CHECK(packed, "fully_fused_projection_2dgs: only packed=true is implemented (the reference uses packed)");
  TORCH_CHECK(!sparse_grad, "fully_fused_projection_2dgs: sparse_grad=true is not implemented (the reference passes false)");
  const int64_t N = means.size(0), C = viewmats.size(0);
  TORCH_CHECK(means.sizes() == torch::IntArrayRef({N, 3}) && quats.sizes() == torch::IntArrayRef({N, 4}) &&
                  scales.sizes() == torch::IntArrayRef({N, 3}), "fully_fused_projection_2dgs: invalid means/quats/scales shape");
  TORCH_CHECK(viewmats.sizes() == torch::IntArrayRef({C, 4, 4}) && Ks.sizes() == torch::IntArrayRef({C, 3, 3}),
              "fully_fused_projection_2dgs: invalid viewmats/Ks shape");
  auto o = Projection2DGS::apply(means, quats, scales, viewmats, Ks, (int64_t)width, (int64_t)height, (double)near_plane,
                                 (double)far_plane, (double)radius_clip, (int64_t)next_sample_seed_impl(stochastic_samples));
  return std::make_tuple(o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
}

Tensor gsplat_cpp::get_view_colors(const Tensor &viewmats, const Tensor &means, const Tensor &radii, const Tensor &colors,
                                   const Tensor &camera_ids, const Tensor &gaussian_ids, at::optional<int> sh_degree) {
  (void)radii;  // packed mode: every row is visible
  if (!sh_degree.has_value()) return colors.index_select(0, gaussian_ids);
  TORCH_CHECK(colors.dim() == 3 && colors.size(2) == 3 && (sh_degree.value() + 1) * (sh_degree.value() + 1) <= colors.size(1),
              "get_view_colors: invalid colors shape");
  return ViewColors::apply(viewmats, means, colors, camera_ids, gaussian_ids, (int64_t)sh_degree.value());
}

std::tuple<Tensor, Tensor, Tensor> gsplat_cpp::tile_encode(int width, int height, int tile_size, const Tensor &means2d_,
                                                           const Tensor &radii_, const Tensor &depths_, bool packed, int64_t C,
                                                           const Tensor &camera_ids, const Tensor &gaussian_ids) {
  TORCH_CHECK(packed, "tile_encode: only packed=true is implemented (the reference uses packed)");
  (void)gaussian_ids;
  torch::NoGradGuard ng;
  Tensor means2d = f32c(means2d_.detach(), "means2d"), depths = f32c(depths_.detach(), "depths"), radii = radii_.contiguous();
  const int64_t M = radii.size(0);
  const int tw = (width + tile_size - 1) / tile_size, th = (height + tile_size - 1) / tile_size;
  Tensor tpg = empty_like_opts(means2d, {M}, torch::kInt32), cum = empty_like_opts(means2d, {std::max<int64_t>(M, 1)}, torch::kInt64);
  Tensor ws = empty_like_opts(means2d, {(int64_t)gsdf_tile_count_ws_bytes(M)}, torch::kUInt8);
  const int64_t I = gsdf_host::count_via_host_word(means2d, [&](int64_t *n_is) {
    check(gsdf_tile_count(M, width, height, tile_size, fp(means2d), M ? radii.data_ptr<int32_t>() : nullptr,
                          M ? tpg.data_ptr<int32_t>() : nullptr, cum.data_ptr<int64_t>(), ws.data_ptr(), n_is, cur_stream()), "tile_encode(count)");
  }, M * (int64_t)tw * th);
  Tensor ids = empty_like_opts(means2d, {I}, torch::kInt64), flat = empty_like_opts(means2d, {I}, torch::kInt32);
  Tensor offs = empty_like_opts(means2d, {C, th, tw}, torch::kInt32);
  Tensor ws2 = empty_like_opts(means2d, {(int64_t)gsdf_tile_encode_ws_bytes(M, I)}, torch::kUInt8);
  check(gsdf_tile_encode(M, C, I, width, height, tile_size, fp(means2d), M ? radii.data_ptr<int32_t>() : nullptr, fp(depths),
                         M ? camera_ids.data_ptr<int64_t>() : nullptr, cum.data_ptr<int64_t>(), ws2.data_ptr(),
                         I ? ids.data_ptr<int64_t>() : nullptr, I ? flat.data_ptr<int32_t>() : nullptr, offs.data_ptr<int32_t>(),
                         cur_stream()), "tile_encode");
  return std::make_tuple(tpg, flat, offs);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>
rasterize_to_pixels_2dgs(const Tensor &means2d, const Tensor &ray_transforms, const Tensor &colors, const Tensor &opacities,
                         const Tensor &normals, const Tensor &densify, int width, int height, int tile_size,
                         const Tensor &isect_offsets, const Tensor &flatten_ids, at::optional<Tensor> backgrounds,
                         at::optional<Tensor> masks, bool packed, const Tensor &means2d_absgrad, bool distloss) {
  TORCH_CHECK(!distloss, "rasterize_to_pixels_2dgs: distloss=true is not implemented (the reference passes false)");
  TORCH_CHECK(packed, "rasterize_to_pixels_2dgs: only packed=true is implemented");
  TORCH_CHECK(colors.size(-1) == 3, "rasterize_to_pixels_2dgs: colors must be [M,3]");
  auto o = Rasterize2DGS::apply(means2d, ray_transforms, colors, opacities, normals, densify, means2d_absgrad, backgrounds, masks,
                                (int64_t)width, (int64_t)height, (int64_t)tile_size, isect_offsets, flatten_ids);
  return std::make_tuple(o[0], o[1], o[2], o[3], o[4], o[5], o[6]);
}
