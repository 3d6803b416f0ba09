"""The REFERENCE'S OWN model classes, unmodified, running on this repository's kernels — against gsdf_model:: (the C++ classes shipped here).

oracle/ref_link/build.py compiles the reference's in-tree host sources where they lie under /root/reference/include
(neural_net/local_map.cpp, sub_map.cpp, encoding_map.cpp, neural_gaussian/neural_gaussian.cpp, optimizer/*.cpp, utils/utils.cpp, ...)
against the drop-in headers and links them with libgsdf_torch.so / libgsdf_hip.so; the prebuilt module travels to the GPU box in
oracle/_ref/.  So what runs here is the reference's LocalMap (with its libtorch torch::nn::Linear decoder for decoder_implementation 0),
its octree sampling, its rasterization_2dgs_sdf / NeuralGS::render / train_callback with refinement and Adam surgery — every call into
the replaced submodules landing in this repository's HIP kernels.  Checked:
  * LocalMap: SDF / isigma, numerical gradient + Hessian, analytic gradient and d eikonal / d table (the reference differentiates twice
    through the drop-in encoding and its own libtorch decoder; gsdf_model uses the fused decoder and its double backward), occupancy
    masks, ray / box intersection, voxel + free-space sampling DRAW FOR DRAW (same torch generator), filter_sample;
  * NeuralGS: render outputs and every parameter gradient, 30 iterations of the training schedule with identical discrete decisions
    (splat counts, anchors bit-equal), parameters, Adam moments and densification statistics; the stochastic SDF samples of the
    reference's default configuration; SDF-aided initialisation;
  * the reference's loss::dssim_loss / rgb_loss (libtorch conv2d on the GPU) against the fused gsdf_l1_dssim kernels.
Skipped where the module was never built (a checkout without /root/reference)."""
import os
import sys

import pytest
import torch

import gs_sdf_amd.synth as synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_link import build as ref_build  # noqa: E402

pytestmark = pytest.mark.gpu
GRID = dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19)     # base 32, x2 per level: fixed in encoding_map.cpp:15-23


@pytest.fixture(scope="module")
def host():
    assert torch.cuda.is_available()
    import gs_sdf_amd.hostlib as h
    return h.load()


@pytest.fixture(scope="module")
def ref(host):
    m = ref_build.load()
    if m is None:
        pytest.skip("oracle/_ref/_gsdf_reference*.so not built (python oracle/ref_link/build.py, needs /root/reference)")
    return m


def rel(a, b):
    """max |a - b| relative to the largest magnitude of the expected tensor"""
    a, b = a.detach().double(), b.detach().double().to(a.device)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30)) if a.numel() else 0.0


def configure(host, ref, impl, **kw):
    """the reference's configuration globals (params/params.h) from gsdf_model's MapConfig, derived values as params.cpp:243-256"""
    cfg = host.MapConfig()
    cfg.leaf_size, cfg.inner_map_size, cfg.decoder_implementation = 0.25, 15.0, impl
    h = 0.5 * cfg.inner_map_size
    d = dict(device="cuda:0", decoder_implementation=impl, leaf_size=cfg.leaf_size, inner_map_size=cfg.inner_map_size, map_size=cfg.map_size(),
             map_size_inv=1.0 / cfg.map_size(), octree_level=cfg.octree_level(), x_min=-h, x_max=h, y_min=-h, y_max=h, z_min=-h, z_max=h,
             hidden_dim=cfg.hidden_dim, geo_num_layer=cfg.geo_num_layer, bce_isigma=1.0 / cfg.bce_sigma, free_sample_num=cfg.free_sample_num,
             output_path="/tmp/gsdf_reference_out", **GRID)
    d.update(kw)
    ref.configure(d)
    return cfg


def make_maps(host, ref, impl):
    """the reference's LocalMap and gsdf_model's with the reference's parameters"""
    cfg = configure(host, ref, impl)
    torch.manual_seed(11)
    rl = ref.LocalMap(torch.tensor([0.0, 0.0, 5.5]))
    cm = host.LocalMap(torch.tensor([0.0, 0.0, 5.5]), cfg)
    rp = rl.named_parameters()
    layers = (0, 2, 4, 6, 8)                      # Linear modules of the reference's Sequential (local_map.cpp:29-42)
    assert set(rp) == ({"encoder_local_map"} | {f"decoder.{i}.{w}" for i in layers for w in ("weight", "bias")} if impl == 0
                       else {"encoder_local_map", "decoder"})
    with torch.no_grad():
        rp["encoder_local_map"].mul_(200.0)       # a hash grid is initialised ~1e-4: give the SDF curvature to compare
        cm.encoder.params_.copy_(rp["encoder_local_map"])
        if impl == 0:
            cm.decoder.params_.copy_(torch.cat([rp[f"decoder.{i}.weight"].reshape(-1) for i in layers]))
            cm.decoder.biases_.copy_(torch.cat([rp[f"decoder.{i}.bias"] for i in layers]))
        else:
            cm.decoder.params_.copy_(rp["decoder"])
    return rl, cm, cfg


@pytest.mark.parametrize("impl", [0, 1])
def test_reference_local_map_on_the_dropin_matches_gsdf_model(host, ref, impl):
    dev = torch.device("cuda:0")
    rl, cm, cfg = make_maps(host, ref, impl)
    g = torch.Generator(device=dev).manual_seed(3)
    xyz = (torch.rand(30000, 3, device=dev, generator=g) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5], device=dev)
    assert torch.equal(cm.xyz_to_zp1_pts(xyz), rl.xyz_to_zp1_pts(xyz))
    a, b = cm.get_sdf(xyz), rl.get_sdf(xyz)
    assert float(b[0].detach().abs().max()) > 1e-3
    assert rel(a[0], b[0]) < 1e-5 and rel(a[1], b[1]) < 1e-5                     # measured 6e-7 (fused decoder vs libtorch GEMMs)
    a, b = cm.get_gradient(xyz, 0.02, None, True, True), rl.get_gradient(xyz, 0.02, None, True, True)
    assert rel(a[0], b[0]) < 5e-4 and rel(a[1], b[1]) < 1e-3                     # differences / delta, / delta^2: measured 5e-5, 6.5e-5
    if impl == 0:                                                                # analytic gradient: the torch decoder only (local_map.cpp:151-172)
        out = []
        for m_, table in ((cm, cm.encoder.params_), (rl, rl.named_parameters()["encoder_local_map"])):
            x = xyz[:8000].clone().requires_grad_(True)
            ga = m_.get_gradient(x, 0.02, None, False, False)[0]
            loss = ((ga.norm(dim=-1) - 1.0) ** 2).mean()
            (gt,) = torch.autograd.grad(loss, [table])
            out.append((ga.detach(), gt))
        assert float(out[1][1].abs().max()) > 0
        assert rel(out[0][0], out[1][0]) < 1e-5 and rel(out[0][1], out[1][1]) < 1e-5    # measured 4e-7, 3e-7
        # the reference's curvature term (curvate_weight > 0: neural_mapping.cpp:117-121, loss.cpp:85-90) on the ANALYTIC Hessian (local_map.cpp:163-168):
        # a third derivative of the drop-in encoder (gsdf_hashgrid_bwd_bwd_bwd) under the reference's own libtorch decoder, against gsdf_model's
        # (fused decoder + its double backward)
        out = []
        for m_, table in ((cm, cm.encoder.params_), (rl, rl.named_parameters()["encoder_local_map"])):
            x = xyz[:8000].clone()
            hess = m_.get_gradient(x, 0.02, None, True, False)[1]
            (gt,) = torch.autograd.grad(hess.sum(-1).abs().mean(), [table])
            out.append((hess.detach(), gt))
        assert float(out[1][1].abs().max()) > 0 and float(out[1][0].abs().max()) > 0
        assert rel(out[0][0], out[1][0]) < 1e-4 and rel(out[0][1], out[1][1]) < 1e-4
    # occupancy structure, intersection, sampling
    n = 4000
    origin = torch.tensor([0.0, 0.0, 5.5], device=dev).expand(n, 3).contiguous()
    direction = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g) * torch.tensor([0.5, 0.5, 0.1], device=dev)
                                              + torch.tensor([0.0, 0.0, 1.0], device=dev), dim=-1)
    depth = 3.0 + torch.rand(n, 1, device=dev, generator=g)
    pts = origin + direction * depth
    for prior in (False, True):
        cm.update_octree_as(pts, prior)
        rl.update_octree_as(pts, prior)
        q = pts + (torch.rand(n, 3, device=dev, generator=g) - 0.5) * 1.5
        va = rl.get_valid_mask(q)
        assert torch.equal(cm.get_valid_mask(q), va) and 0 < int(va.sum()) < n
    assert torch.equal(cm.get_inrange_mask(q * 2.0, 0.1), rl.get_inrange_mask(q * 2.0, 0.1))
    for x, y in zip(cm.get_intersect_point(origin, direction, 0.1), rl.get_intersect_point(origin, direction, 0.1)):
        assert torch.equal(x, y)
    rays = dict(origin=origin, direction=direction, depth=depth, xyz=pts, ray_sdf=torch.zeros(n, 1, device=dev), ridx=torch.arange(n, device=dev))
    for free in (False, True):
        torch.manual_seed(77)
        sc = cm.sample(rays, 1, free)
        torch.manual_seed(77)
        sr = rl.sample(rays, 1, free)
        assert sorted(sc.keys()) == sorted(sr.keys())
        assert sr["xyz"].shape[0] > n // 2
        assert torch.equal(sc["ridx"], sr["ridx"])
        for k in ("xyz", "ray_sdf", "depth"):
            assert sc[k].shape == sr[k].shape and rel(sc[k], sr[k]) < 1e-6, k     # draw for draw: measured 0
    fc, fr = cm.filter_sample(dict(xyz=q, ridx=torch.arange(n, device=dev))), rl.filter_sample(dict(xyz=q, ridx=torch.arange(n, device=dev)))
    assert torch.equal(fc["ridx"], fr["ridx"])


def test_reference_init_gs_with_sdf_matches_gsdf_model(host, ref):
    dev = torch.device("cuda:0")
    rl, cm, cfg = make_maps(host, ref, 0)
    g = torch.Generator(device=dev).manual_seed(9)
    pts = (torch.rand(20000, 3, device=dev, generator=g) - 0.5) * 10.0 + torch.tensor([0.0, 0.0, 5.5], device=dev)
    ref.configure(dict(vis_batch_pt_num=8192))                                  # several batches
    a = ref.init_gs_with_sdf(rl, pts, 0.5 * cfg.leaf_size, True)
    b = host.init_gs_with_sdf(cm, pts, 0.5 * cfg.leaf_size, True, 8192)
    assert sorted(a.keys()) == sorted(b.keys()) == ["curv_dom", "grad", "opacity", "quaternion"]
    for k in ("curv_dom", "grad", "opacity"):
        ok = ~(a[k].isnan().reshape(20000, -1).any(-1) | b[k].isnan().reshape(20000, -1).any(-1))
        assert int(ok.sum()) > 19000 and rel(b[k][ok], a[k][ok]) < 1e-3, k        # measured 5e-5 (second differences of the SDF)
    qa, qb = torch.nn.functional.normalize(a["quaternion"], dim=-1), torch.nn.functional.normalize(b["quaternion"], dim=-1)
    assert float(((qa * qb).sum(-1).abs() > 0.999).float().mean()) > 0.99


KW = dict(refine_start_iter=8, refine_every=4, reset_every=20, grow_grad2d=2e-7, center_reg=True, sh_degree_interval=10)
PFIELDS = ("offsets_", "scaling_", "quaternion_", "opacity_", "features_dc_", "features_rest_")
N, W, H = 6000, 320, 192


@pytest.fixture(scope="module")
def splats(host, ref):
    """the reference's NeuralGS(points) with a synthetic scene's appearance, and gsdf_model's NeuralGS on the same tensors"""
    dev = torch.device("cuda:0")
    sc = synth.make_scene(N, W, H, sh_degree=1, seed=3)
    K = sc["K"][0]
    cam = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H)
    poses = [torch.linalg.inv(v)[:3, :4].contiguous() for v in synth.make_views(4, seed=2)]
    gcfg = host.GSConfig()
    gcfg.sh_degree = 1
    for k, v in KW.items():
        setattr(gcfg, k, v)
    configure(host, ref, 0, sh_degree=1, near=gcfg.near, far=gcfg.far, use_absgrad=gcfg.use_absgrad, geo_init=False, mesh_init=False, sky_init=False,
              prune_opa=gcfg.prune_opa, grow_scale3d=gcfg.grow_scale3d, grow_scale2d=gcfg.grow_scale2d, prune_scale3d=gcfg.prune_scale3d,
              refine_scale2d_stop_iter=gcfg.refine_scale2d_stop_iter, lr_end=gcfg.lr_end, detach_sdf_grad=gcfg.detach_sdf_grad,
              vis_batch_pt_num=gcfg.vis_batch_pt_num, pause_refine=False, **KW)
    torch.manual_seed(5)
    rg = ref.NeuralGS(None, sc["means"].to(dev), 4, 1.0, False)
    assert rg.sh_degree_to_use_ == 0 and abs(rg.spatial_scale_ - 1.0) < 1e-7
    assert {k: tuple(v.shape) for k, v in rg.named_parameters().items()} == dict(
        anchors=(N, 3), offsets=(N, 3), scaling=(N, 3), quaternion=(N, 4), opacity=(N,), features_dc=(N, 1, 3), features_rest=(N, 3, 3))
    with torch.no_grad():
        rg.scaling_.copy_(sc["log_scales"].to(dev)); rg.quaternion_.copy_(sc["quats"].to(dev)); rg.opacity_.copy_(sc["logit_opacities"].to(dev))
        rg.features_dc_.copy_(sc["sh"][:, :1].to(dev)); rg.features_rest_.copy_(sc["sh"][:, 1:].to(dev))
    cg = host.NeuralGS(None, rg.anchors_.detach(), rg.scaling_.detach(), rg.quaternion_.detach(), rg.opacity_.detach(), rg.features_dc_.detach(),
                       rg.features_rest_.detach(), 4, 1.0, gcfg)
    rg.sh_degree_to_use_ = cg.sh_degree_to_use_ = 1
    return rg, cg, cam, poses


def test_reference_render_on_the_dropin_matches_gsdf_model(host, ref, splats):
    dev = torch.device("cuda:0")
    rg, cg, cam, poses = splats
    host.set_sample_mode(False)            # the reference discards the projection's samples when k_center_reg is set: no draw needed
    rr, rc = rg.render(poses[1], *cam, True, 0), cg.render(poses[1], *cam, True, 0)
    assert set(rr) == set(rc)
    for k in rr:
        a, b = rc[k], rr[k]
        assert a.shape == b.shape, k
        if a.dtype in (torch.int32, torch.int64, torch.bool):
            assert torch.equal(a.cpu(), b.cpu()), k
        else:
            assert rel(a.to(dev), b.to(dev)) < 1e-5, k                           # measured 0 / 1e-7
    g = torch.Generator(device=dev).manual_seed(1)
    tgt, w_n = torch.rand(H, W, 3, device=dev, generator=g), torch.randn(H, W, 3, device=dev, generator=g)
    for r in (rr, rc):
        loss = ((r["color"] - tgt).abs().mean() + 0.1 * r["depth"].mean() + 0.05 * (r["render_normal"] * w_n.view_as(r["render_normal"])).mean()
                + 0.02 * r["alpha"].mean() + 1e-3 * (r["samples"] * r["samples_weights"]).sum())
        loss.backward()
    for f in PFIELDS:
        a, b = getattr(cg, f).grad, getattr(rg, f).grad
        assert a is not None and b is not None and float(b.abs().max()) > 0, f
        assert rel(a, b) < 1e-5, f                                               # measured 2e-7
    assert rel(rc["gradient_2dgs"].grad, rr["gradient_2dgs"].grad) < 1e-5
    for f in PFIELDS:
        getattr(cg, f).grad = None
        getattr(rg, f).grad = None


def test_reference_training_schedule_on_the_dropin_matches_gsdf_model(host, ref, splats):
    """render -> L1 -> backward -> Adam -> train_callback for 30 iterations, refinement every 4 from 12 on, opacity reset at 20:
    NeuralGS::train_callback, grow_gs / duplicate / split / prune_gs / reset_opacity and optimizer_utils.cpp are the reference's.

    Iterations 1-11 (Adam, densification statistics, NaN / invisible pruning, SH schedule, learning-rate decay; no refinement yet) are
    compared tightly.  From the first refinement on, every decision is a threshold on accumulated gradients; the compositing backward
    accumulates with fp32 atomics, so the two runs differ in the last bits and a splat sitting on a threshold may be duplicated in one
    and not in the other (seen once in three runs, at 54 K splats).  The counts must therefore agree to 1e-3 at every iteration, and the
    element-wise comparison at the end is made when no decision flipped (the usual case)."""
    rg, cg, cam, poses = splats
    host.set_sample_mode(False)
    ropt, copt = rg.make_optimizer(None, 1e-3), cg.make_optimizer()
    assert ropt.n_groups() == copt.n_groups() == 6
    with torch.no_grad():
        target = [cg.render(p, *cam, False, 0)["color"].detach() * 0.5 + 0.25 for p in poses]
    fin = lambda t: torch.nan_to_num(t.detach(), neginf=-1e4)                    # split children: log(0) in the unused third scale

    def bulk(x, y, tol, what, worst):
        """all but 1e-3 of the elements within tol of the expected tensor's largest magnitude, none further than `worst` of it: gradients and
        visibilities are discontinuous in the parameters (one (pixel, splat) pair crossing the alpha >= 1/255 test moves a splat's gradient
        or visibility by percents), so after refinement single elements may differ while the tensors agree"""
        x, y = x.detach().double(), y.detach().double()
        assert x.shape == y.shape, what
        d = (x - y).abs() / (y.abs().max() + 1e-30)
        assert float((d > tol).double().mean()) < 1e-3 and float(d.max()) < worst, (what, float((d > tol).double().mean()), float(d.max()))

    def compare(tol_p, tol_m1, tol_m2, tol_state, worst):
        assert torch.equal(rg.anchors_, cg.anchors_)
        for k, f in enumerate(PFIELDS):
            bulk(fin(getattr(cg, f)), fin(getattr(rg, f)), tol_p, f, worst)
            mr, mc = ropt.moments(k), copt.moments(k)
            bulk(mc[0], mr[0], tol_m1, f + " exp_avg", worst)
            bulk(mc[1], mr[1], tol_m2, f + " exp_avg_sq", worst)
            assert ropt.param(k).data_ptr() == getattr(rg, f).data_ptr(), f + ": the reference's optimizer does not hold the live tensor"
        st_r, st_c = rg.state, cg.state
        for k in ("count", "grad2d", "vis"):
            bulk(st_c[k].float(), st_r[k].float(), tol_state, k, worst)

    sizes, flipped = [], None
    for it in range(1, 31):
        ropt.zero_grad(); copt.zero_grad()
        rr, rc = rg.render(poses[it % 4], *cam, True, 0), cg.render(poses[it % 4], *cam, True, 0)
        (rr["color"] - target[it % 4]).abs().mean().backward()
        (rc["color"] - target[it % 4]).abs().mean().backward()
        ropt.step(); copt.step()
        torch.manual_seed(1000 + it); rg.train_callback(it, 100, ropt, rr)
        torch.manual_seed(1000 + it); cg.train_callback(it, 100, copt, rc)
        nr, nc = rg.anchors_.shape[0], cg.anchors_.shape[0]
        assert rg.sh_degree_to_use_ == cg.sh_degree_to_use_
        assert abs(ropt.lr(0) - copt.lr(0)) <= 1e-7 * ropt.lr(0)
        if it <= 11:
            assert nr == nc, (it, nr, nc)
        else:
            assert abs(nr - nc) <= max(8, int(1e-3 * nr)), (it, nr, nc)
            if flipped is None and nr != nc:
                flipped = it
        if it == 11:
            compare(1e-4, 1e-3, 1e-4, 1e-4, 0.05)                                      # measured at 30 iterations: 9e-5, 4.5e-4, 1.6e-6, 3.5e-5
        sizes.append(nr)
    assert len(set(sizes)) > 3 and sizes[-1] > 4 * N, sizes                      # measured 6000 -> 71076
    if flipped is None:
        compare(2e-3, 5e-3, 1e-3, 1e-3, 0.5)
    else:
        print(f"a refinement decision flipped at iteration {flipped}: sizes {rg.anchors_.shape[0]} / {cg.anchors_.shape[0]}")


def test_reference_default_configuration_gets_stochastic_samples_from_the_dropin(host, ref, splats):
    """k_center_reg = 0 (config/base.yaml has no center_reg key -> params.cpp default 0): rasterization_2dgs_sdf hands the projection's own
    `samples` on (neural_gaussian.cpp:258-264); the drop-in's default mode must therefore draw them on the splats' discs"""
    rg, cg, cam, poses = splats
    ref.configure(dict(center_reg=False))
    host.set_sample_mode(True)             # the library default; the Python harness had switched it off
    try:
        rr = rg.render(poses[0], *cam, True, 0)
    finally:
        host.set_sample_mode(False)
        ref.configure(dict(center_reg=True))
    ids = rr["gaussian_ids"]
    d = (rr["samples"].detach() - rr["xyz"].detach().index_select(0, ids)).norm(dim=-1)
    w = rr["samples_weights"].detach().reshape(-1)
    scale = rg.get_scale().detach().index_select(0, ids)
    eps = torch.sqrt(-2.0 * torch.log(w))
    assert float(d.max()) > 0 and bool((w > 0).all()) and bool((w <= 1.0 + 1e-6).all())
    assert bool((d <= scale[:, :2].max(-1).values * eps * (1 + 1e-3) + 1e-6).all())
    assert bool((d >= scale[:, :2].min(-1).values * eps * (1 - 1e-3) - 1e-6).all())
    (rr["samples"].sum()).backward()
    assert float(rg.scaling_.grad.abs().sum()) > 0


def test_reference_photometric_loss_on_the_gpu_matches_the_fused_kernels(ref):
    """loss::rgb_loss / loss::dssim_loss (loss.cpp:22-47: libtorch conv2d with the reference's own window, moved to the GPU by ssim()) against
    gsdf_l1_dssim_fwd / _bwd, weights 0.8 / 0.2 as neural_mapping.cpp:237-240"""
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4)
    for (h, w) in ((192, 320), (67, 131)):
        img = torch.rand(h, w, 3, device=dev, generator=g)
        gt = (img + 0.2 * torch.randn(h, w, 3, device=dev, generator=g)).clamp(0, 1)
        out = []
        for fused in (False, True):
            x = img.clone().requires_grad_(True)
            loss = ops.l1_dssim_loss(x, gt, 0.8, 0.2) if fused else 0.8 * ref.rgb_loss(x, gt) + 0.2 * ref.dssim_loss(x, gt)
            (gx,) = torch.autograd.grad(loss, [x])
            out.append((loss.detach(), gx))
        assert rel(out[1][0], out[0][0]) < 1e-5
        assert rel(out[1][1], out[0][1]) < 1e-4


def test_reference_joint_iteration_on_the_dropin_matches_gsdf_model(host, ref):
    """The joint iteration of NeuralSLAM::gs_train (neural_mapping.cpp:400-486 — the one file of the path that cannot be compiled here: ROS,
    tf, the data loader) sequenced in Python over the REFERENCE'S compiled pieces, in the reference's default configuration (torch decoder,
    analytic eikonal): LocalMap::get_sdf / get_gradient / get_valid_mask, NeuralGS::render / train_callback, loss::sdf_loss /
    eikonal_loss / rgb_loss / dssim_loss / gs_sdf_loss, one torch::optim::Adam over the SDF network's and the splats' groups.  The same
    sequence over gsdf_model:: with this repository's mirrors of the losses must produce the same loss values, iteration by iteration,
    and the same parameters after 10 Adam steps of both networks."""
    import gs_sdf_amd.ops as ops
    import gs_sdf_amd.sdf as sdfm
    dev = torch.device("cuda:0")
    rl, cm, cfg = make_maps(host, ref, 0)
    n, w, h = 6000, 320, 192
    sc = synth.make_scene(n, w, h, sh_degree=1, seed=3)
    Kc = sc["K"][0]
    cam = (float(Kc[0, 0]), float(Kc[1, 1]), float(Kc[0, 2]), float(Kc[1, 2]), w, h)
    poses = [torch.linalg.inv(v)[:3, :4].contiguous() for v in synth.make_views(4, seed=2)]
    gcfg = host.GSConfig()
    gcfg.sh_degree, gcfg.center_reg, gcfg.refine_start_iter = 1, True, 1000
    ref.configure(dict(sh_degree=1, near=gcfg.near, far=gcfg.far, use_absgrad=gcfg.use_absgrad, geo_init=False, mesh_init=False, sky_init=False,
                       prune_opa=gcfg.prune_opa, grow_grad2d=gcfg.grow_grad2d, grow_scale3d=gcfg.grow_scale3d, grow_scale2d=gcfg.grow_scale2d,
                       prune_scale3d=gcfg.prune_scale3d, refine_scale2d_stop_iter=gcfg.refine_scale2d_stop_iter, refine_start_iter=1000,
                       refine_every=gcfg.refine_every, reset_every=gcfg.reset_every, sh_degree_interval=gcfg.sh_degree_interval, lr_end=gcfg.lr_end,
                       detach_sdf_grad=False, vis_batch_pt_num=gcfg.vis_batch_pt_num, pause_refine=False, center_reg=True))
    torch.manual_seed(5)
    rg = ref.NeuralGS(rl, sc["means"].to(dev), 1000, 1.0, False)
    with torch.no_grad():
        rg.scaling_.copy_(sc["log_scales"].to(dev)); rg.quaternion_.copy_(sc["quats"].to(dev)); rg.opacity_.copy_(sc["logit_opacities"].to(dev))
        rg.features_dc_.copy_(sc["sh"][:, :1].to(dev)); rg.features_rest_.copy_(sc["sh"][:, 1:].to(dev))
    cg = host.NeuralGS(cm, rg.anchors_.detach(), rg.scaling_.detach(), rg.quaternion_.detach(), rg.opacity_.detach(), rg.features_dc_.detach(),
                       rg.features_rest_.detach(), 1000, 1.0, gcfg)
    rg.sh_degree_to_use_ = cg.sh_degree_to_use_ = 1
    for m_ in (rl, cm):
        m_.update_octree_as(rg.anchors_.detach(), False)
    ropt, copt = rg.make_optimizer(rl, 1e-3), cg.make_optimizer(cm, 1e-3)
    assert ropt.n_groups() == 11 + 6 and copt.n_groups() == 3 + 6          # the reference's Sequential registers 10 decoder tensors
    host.set_sample_mode(False)
    g = torch.Generator(device=dev).manual_seed(21)
    pool = [(torch.rand(8192, 3, device=dev, generator=g) - 0.5) * 12.0 + torch.tensor([0.0, 0.0, 5.5], device=dev) for _ in range(4)]
    tgt_sdf = [torch.randn(8192, 1, device=dev, generator=g) * 0.05 for _ in range(4)]
    with torch.no_grad():
        target = [cg.render(p, *cam, False, 0)["color"].detach() * 0.5 + 0.25 for p in poses]

    def iteration(lm, gs, opt, L, it):
        opt.zero_grad()
        pts, ts = pool[it % 4], tgt_sdf[it % 4]
        s, isig = lm.get_sdf(pts)
        grad = lm.get_gradient(pts.clone(), 0.02, None, False, False)[0]                       # analytic (k_numerical_grad = 0)
        loss = L["sdf"](s, ts, isig) + 0.1 * L["eik"](grad)                                    # :138-188
        r = gs.render(poses[it % 4], *cam, True, 0)
        loss = loss + L["photo"](r["color"], target[it % 4])                                   # :237-240
        vis = r["visibilities"].detach()
        valid = lm.get_valid_mask(r["samples"].detach()) & (vis > 0.1).squeeze(-1)            # :420-462
        ids = valid.nonzero().squeeze(-1)
        assert ids.numel() > 100
        xs = r["samples"].index_select(0, ids)
        wts = (r["samples_weights"] * vis).detach().index_select(0, ids)
        loss = loss + 1e-3 * L["gs_sdf"](lm.get_sdf(xs)[0], wts) / ids.numel()
        loss.backward()
        opt.step()
        gs.train_callback(it, 1000, opt, r)
        return float(loss.detach()), int(ids.numel())

    L_ref = dict(sdf=ref.sdf_loss, eik=ref.eikonal_loss, gs_sdf=ref.gs_sdf_loss, photo=lambda a, b: 0.8 * ref.rgb_loss(a, b) + 0.2 * ref.dssim_loss(a, b))
    L_own = dict(sdf=sdfm.sdf_loss, eik=sdfm.eikonal_loss, gs_sdf=sdfm.gs_sdf_loss, photo=lambda a, b: ops.l1_dssim_loss(a, b, 0.8, 0.2))
    hist = []
    for it in range(1, 11):
        a = iteration(rl, rg, ropt, L_ref, it)
        b = iteration(cm, cg, copt, L_own, it)
        hist.append((a, b))
        assert a[1] == b[1], (it, a, b)                                                        # the same visible, occupancy-valid samples
        assert abs(a[0] - b[0]) <= 1e-4 * abs(a[0]), (it, a, b)
    assert hist[-1][0][0] < hist[0][0][0], hist                                                # and it trains

    def bulk(x, y, tol, what):
        d = (x.detach().double() - y.detach().double()).abs() / (y.detach().double().abs().max() + 1e-30)
        assert float((d > tol).double().mean()) < 1e-3 and float(d.max()) < 0.2, (what, float((d > tol).double().mean()), float(d.max()))
    rp = rl.named_parameters()
    layers = (0, 2, 4, 6, 8)
    bulk(cm.encoder.params_, rp["encoder_local_map"], 1e-3, "hash table after 10 steps")
    bulk(cm.decoder.params_, torch.cat([rp[f"decoder.{i}.weight"].reshape(-1) for i in layers]), 1e-3, "decoder weights")
    bulk(cm.decoder.biases_, torch.cat([rp[f"decoder.{i}.bias"] for i in layers]), 1e-3, "decoder biases")
    for f in PFIELDS:
        bulk(getattr(cg, f), getattr(rg, f), 1e-3, f)


def test_reference_meshing_on_the_dropin(host, ref):
    """LocalMap::meshing_ (local_map.cpp:329-447, the reference's) -> utils::meshgrid_3d, get_valid_mask, get_sdf, mc::marching_cubes
    (the reference's cumcubes.cpp over this repository's mc::marching_cubes_wrapper), spc_ops::points_to_neighbors for the boundary
    filter — against the same steps written with this repository's Python operators (tcnn decoder on both sides: identical SDF values,
    so the meshes must be identical)."""
    from gs_sdf_amd.mesher import marching_cubes
    dev = torch.device("cuda:0")
    rl, cm, cfg = make_maps(host, ref, 1)
    ref.configure(dict(vis_attribute=0, vis_batch_pt_num=1 << 30, dataset_type=0))          # one chunk, grey vertices, OpenCV world
    g = torch.Generator(device=dev).manual_seed(13)
    pts = (torch.rand(3000, 3, device=dev, generator=g) - 0.5) * torch.tensor([6.0, 6.0, 2.0], device=dev) + torch.tensor([0.0, 0.0, 5.5], device=dev)
    rl.update_octree_as(pts, False)
    cm.update_octree_as(pts, False)
    res = 0.125
    verts, faces, colors = rl.meshing_(res)
    assert len(verts) == 1 and faces[0].shape[0] > 1000, (len(verts), [tuple(f.shape) for f in faces])
    v_ref, f_ref = verts[0].to(dev), faces[0].to(dev)
    assert bool((colors[0] == 127).all())
    # the same steps over gsdf_model / the Python operators
    pos = cm.pos_W_M_.reshape(3)
    lo = (cm.xyz_min_W_.reshape(3) - pos + 0.5 * cfg.leaf_size).tolist()                     # xyz_min_M_margin_ (sub_map.cpp:17-18)
    hi = (cm.xyz_max_W_.reshape(3) - pos - 0.5 * cfg.leaf_size).tolist()
    c = pos.tolist()
    lower = [lo[k] + c[k] for k in range(3)]
    ax = [torch.arange(lower[k], hi[k] + c[k] + res, res, device=dev) for k in range(3)]
    grid = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1)
    shape = grid.shape[:3]
    xyz = grid.reshape(-1, 3)
    mask = cm.get_valid_mask(xyz)
    sdf = torch.full((xyz.shape[0], 1), 1e-6, device=dev)
    with torch.no_grad():
        sdf[mask] = cm.get_sdf(xyz[mask])[0]
    upper = [lower[k] + shape[k] * res for k in range(3)]
    v, f = marching_cubes(sdf.view(*shape), 0.0, lower, upper, "reference")
    q = (v / res).floor().to(torch.int16)
    nb = host.points_to_neighbors(q).view(-1, 3).to(torch.float32) * res
    ok = cm.get_valid_mask(nb).view(-1, 27).all(1)
    keep = ok[f.view(-1).long()].view(-1, 3).all(-1).nonzero().view(-1)
    f = f.index_select(0, keep)
    assert v.shape == v_ref.shape and f.shape == f_ref.shape, (v.shape, v_ref.shape, f.shape, f_ref.shape)
    assert torch.equal(v, v_ref) and torch.equal(f.to(f_ref.dtype), f_ref)


@pytest.mark.parametrize("impl", [0, 1])
def test_gsdf_model_checkpoint_loads_in_the_references_local_map_and_back(host, ref, tmp_path, impl):
    """torch::save(gsdf_model::LocalMap) -> torch::load(the reference's LocalMap) and the other way (neural_mapping.cpp:1331-1352): the C++
    model class writes the reference's archive layout in both decoder implementations."""
    rl, cm, cfg = make_maps(host, ref, impl)
    with torch.no_grad():
        cm.encoder.params_.mul_(0.5); cm.decoder.params_.mul_(-1.5)
        if impl == 0:
            cm.decoder.biases_.add_(0.125)
    p1 = str(tmp_path / "from_gsdf_model.pt")
    cm.save_checkpoint(p1)
    ref.load_local_map(rl, p1)
    rp = rl.named_parameters()
    layers = (0, 2, 4, 6, 8)
    assert torch.equal(rp["encoder_local_map"].reshape(-1), cm.encoder.params_.reshape(-1))
    if impl == 0:
        assert torch.equal(torch.cat([rp[f"decoder.{i}.weight"].reshape(-1) for i in layers]), cm.decoder.params_)
        assert torch.equal(torch.cat([rp[f"decoder.{i}.bias"] for i in layers]), cm.decoder.biases_)
    else:
        assert torch.equal(rp["decoder"].reshape(-1), cm.decoder.params_)
    with torch.no_grad():
        rp["encoder_local_map"].add_(1.0)
    p2 = str(tmp_path / "from_reference.pt")
    ref.save_local_map(rl, p2)
    cm.load_checkpoint(p2)
    assert torch.equal(cm.encoder.params_.reshape(-1), rp["encoder_local_map"].reshape(-1))
