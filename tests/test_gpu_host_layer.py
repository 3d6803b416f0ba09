"""The C++/libtorch operator layer (gs-sdf_amd/host: reference-named headers over the C ABI) driven through its
pybind11 harness: results must equal the Python mirror's (same kernels underneath), autograd must reach every
differentiable input including the leaf `densify`/`means2d_absgrad` tensors and the hash grid's second order."""
import numpy as np
import pytest
import torch

import gs_sdf_amd.synth as synth
from util import assert_close, assert_equal_int

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    assert torch.cuda.is_available()
    import gs_sdf_amd.hostlib as h
    return h.load()


def test_splat_operators_match_python_mirror(host):
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    N, W, H, deg = 20000, 480, 272, 2
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=7)
    vm = synth.make_views(2, seed=5)[1:].to(dev)
    K = sc["K"].to(dev)
    leaves = lambda: [t.to(dev).clone().requires_grad_(True) for t in (sc["means"], sc["quats"], sc["log_scales"].exp(),
                                                                        torch.sigmoid(sc["logit_opacities"]), sc["sh"])]
    res = []
    for mod in ("cpp", "py"):
        means, quats, scales, opac, sh = leaves()
        if mod == "cpp":
            o = host.fully_fused_projection_2dgs(means, quats, scales, vm, K, W, H, 0.05, 300.0, 0.0, True, False)
        else:
            o = ops.fully_fused_projection_2dgs(means, quats, scales, vm, K, W, H, 0.05, 300.0, 0.0, True, False)
        cam, gid, radii, m2d, dep, rt, nrm, smp, sw = o
        if mod == "cpp":
            col = host.get_view_colors(vm, means, radii, sh, cam, gid, deg)
            tpg, flat, offs = host.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid)
        else:
            col = ops.get_view_colors(vm, means, radii, sh, cam, gid, deg)
            tpg, flat, offs = ops.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid)
        dens = torch.zeros_like(m2d, requires_grad=True)
        absg = torch.zeros_like(m2d, requires_grad=True)
        bg = torch.tensor([[0.2, 0.1, 0.3]], device=dev)
        f = host.rasterize_to_pixels_2dgs if mod == "cpp" else ops.rasterize_to_pixels_2dgs
        rc, rd, ra, rn, rdist, rm, vis = f(m2d, rt, col, opac[gid], nrm, dens, W, H, 16, offs, flat, bg, None, True, absg, False)
        ug = synth.upstream_grads(H, W, seed=2)
        loss = sum((t * ug[k].to(dev)).sum() for t, k in ((rc, "v_render_colors"), (rd, "v_render_depths"), (ra, "v_render_alphas"),
                                                          (rn, "v_render_normals"), (rm, "v_render_median"))) + (smp * 0.1).sum()
        loss.backward()
        res.append(dict(cam=cam, gid=gid, radii=radii, m2d=m2d, rt=rt, col=col, tpg=tpg, flat=flat, offs=offs, rc=rc, rd=rd, ra=ra,
                        rn=rn, rm=rm, vis=vis, g_means=means.grad, g_quats=quats.grad, g_scales=scales.grad, g_opac=opac.grad,
                        g_sh=sh.grad, g_dens=dens.grad, g_abs=absg.grad))
    a, b = res
    for k in ("cam", "gid", "radii", "tpg", "flat", "offs"):
        assert_equal_int(a[k], b[k], k)
    for k in ("m2d", "rt", "col", "rc", "rd", "ra", "rn", "rm", "vis"):
        assert torch.equal(a[k], b[k]), f"{k}: forward differs between the C++ and Python layers (same kernels)"
    for k in ("g_means", "g_quats", "g_scales", "g_opac", "g_sh", "g_dens", "g_abs"):
        assert a[k] is not None and b[k] is not None, k
        # atomics: the accumulation order of a splat's per-tile records differs between runs; a splat seen edge-on sums cancelling
        # contributions, so a handful of rows may move by more than 1e-4 of their (small) total
        assert_close(a[k], b[k], 1e-4, k, outlier_frac=1e-4, outlier_rel=5e-2)
    with pytest.raises(RuntimeError):
        host.fully_fused_projection_2dgs(res[0]["m2d"], res[0]["m2d"], res[0]["m2d"], vm, K, W, H, 0.05, 300.0, 0.0, True, False)
    with pytest.raises(RuntimeError):
        host.tile_encode(W, H, 16, a["m2d"], a["radii"], a["m2d"][:, 0].contiguous(), False, 1, a["cam"], a["gid"])


def test_tcnn_objects_first_and_second_order(host, oracle):
    dev = torch.device("cuda:0")
    cfg = dict(n_levels=16, n_feat=2, log2_hashmap=19, base_res=32, per_level_scale=2.0)
    enc = host.TCNNEncoding(16, 2, 19, 32, 2.0)
    assert enc.get_out_dim() == 32 and enc.params_.numel() == 15_269_888 and enc.params_.is_cuda
    assert float(enc.params_.detach().abs().max()) <= 1e-4
    g = torch.Generator().manual_seed(0)
    B = 4096
    x = torch.rand(B, 3, generator=g)
    table = torch.rand(enc.params_.numel(), generator=g) * 2 - 1
    enc.params_ = table.to(dev).requires_grad_(True)
    xd = x.to(dev).requires_grad_(True)
    feat = enc.forward(xd)
    n = lambda t: t.detach().cpu().numpy()
    assert_close(feat, oracle.grid_fwd(n(x), n(table).reshape(-1, 2), cfg, prec="f32"), 1e-5, "C++ TCNNEncoding.forward")
    v = torch.randn(B, 32, generator=g)
    vd = v.to(dev).requires_grad_(True)
    v_x, v_t = torch.autograd.grad(feat, (xd, enc.params_), vd, create_graph=True)
    vt_o, vx_o = oracle.grid_bwd(n(x), n(table).reshape(-1, 2), n(v), cfg, prec="f32")
    assert_close(v_x, vx_o, 1e-4, "v_x"); assert_close(v_t.view(-1, 2), vt_o, 1e-4, "v_table")
    vv = torch.randn(B, 3, generator=g)
    g_v, g_t = torch.autograd.grad((v_x * vv.to(dev)).sum(), (vd, enc.params_))
    gv_o, gt_o, _ = oracle.grid_bwd_bwd(n(x), n(table).reshape(-1, 2), n(v), n(vv), cfg, prec="f32")
    assert_close(g_v, gv_o, 1e-4, "double backward d/dv_feat"); assert_close(g_t.view(-1, 2), gt_o, 1e-4, "double backward d/dtable")
    torch.manual_seed(1234)                  # the C++ constructor draws from the global generator (like torch::nn::Linear)
    net = host.TCNNNetwork(32, 2, 64, 3)
    W = n(net.params_)
    assert W.size == 32 * 64 + 2 * 64 * 64 + 64 * 2
    f_in = torch.randn(B, 32, generator=g).to(dev).requires_grad_(True)
    out = net.forward(f_in)
    assert_close(out, oracle.mlp_fwd(n(f_in), [32, 64, 64, 64, 2], W, None, prec="f64"), 1e-4, "C++ TCNNNetwork.forward")
    vo = torch.randn(B, 2, generator=g)
    out.backward(vo.to(dev))
    v_in, v_w, _ = oracle.mlp_bwd(n(f_in), [32, 64, 64, 64, 2], W, None, n(vo), prec="f64")
    # a pre-activation within fp32 rounding of 0 flips its ReLU mask against the fp64 oracle: a handful of elements
    assert_close(f_in.grad, v_in, 1e-4, "mlp v_in", outlier_frac=1e-5, outlier_rel=5e-2)
    assert_close(net.params_.grad, v_w, 1e-4, "mlp v_w", outlier_frac=1e-5, outlier_rel=5e-2)


def test_cpp_encoding_forward_stencil_equals_forward(host):
    """TCNNEncoding::forward_stencil (extension for the gsdf_extras edits): same features bit for bit, same gradients as
    forward() on the 7-row stencil batches of get_gradient's numerical branch — group-walking forward, stencil-merging scatter."""
    dev = torch.device("cuda:0")
    enc = host.TCNNEncoding(16, 2, 19, 32, 2.0)
    g = torch.Generator().manual_seed(3)
    enc.params_ = (torch.rand(enc.params_.numel(), generator=g) * 2 - 1).to(dev).requires_grad_(True)
    n_grp, delta = 6000, 0.02 / 16.0
    base = torch.rand(n_grp, 3, generator=g) * 0.8 + 0.1
    offs = torch.tensor([[0, 0, 0], [delta, 0, 0], [-delta, 0, 0], [0, delta, 0], [0, -delta, 0], [0, 0, delta], [0, 0, -delta]])
    x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous().to(dev)
    v = torch.randn(7 * n_grp, 32, generator=g).to(dev)
    res = []
    for stencil in (False, True):
        xd = x.clone().requires_grad_(True)
        feat = enc.forward_stencil(xd, n_grp, delta) if stencil else enc.forward(xd)
        res.append((feat.detach(), torch.autograd.grad(feat, (xd, enc.params_), v)))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1][0], res[1][1][0])                     # d/dx: the same kernel
    assert_close(res[1][1][1], res[0][1][1], 5e-5, "table gradient, stencil-merged vs plain binned scatter")
    with pytest.raises(RuntimeError):
        enc.forward_stencil(x, n_grp + 1, delta)


@pytest.mark.parametrize("delta,n", [(0.02, 4000), (0.0, 4000), (0.02, 2000)])      # 7 x 4000 rows: binned stencil scatter; 7 x 2000: atomic
def test_cpp_gs_sdf_coupling_node_matches_python_mirror(host, delta, n):
    """gsdf_extras::gs_sdf_coupling (the GS<->SDF block of neural_mapping.cpp:420-462 as one C++ autograd node with in-place
    parameter-gradient accumulation) == LocalMap.gs_sdf_coupling of the Python mirror: loss, d/d samples, table and decoder gradients."""
    import gs_sdf_amd.sdf as sdf
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    M = 9000
    pts = ((torch.rand(M, 3, generator=g) - 0.5) * 6.0).to(dev)
    w_all = torch.rand(M, 1, generator=g).to(dev)
    ids = torch.randperm(M, generator=g)[:n].sort().values.to(dev)
    lm = sdf.LocalMap([0.1, 0.2, -0.3], 8.0, decoder_implementation=1, device=dev, seed=9)
    with torch.no_grad():
        lm.encoder.params_.mul_(1e3)
    enc = host.TCNNEncoding(16, 2, 19, 32, 2.0)
    enc.params_ = lm.encoder.params_.detach().clone()
    dec = host.TCNNNetwork(32, 2, 64, 3)
    dec.params_ = lm.decoder.params_.detach().clone()
    t_grad, d_grad = torch.zeros_like(enc.params_), torch.zeros_like(dec.params_)
    xc = pts.clone().requires_grad_(True)
    loss_c = host.gs_sdf_coupling(xc, ids, w_all, enc, dec, [float(v) for v in lm._origin], float(lm.map_size_inv), 1e-3, delta, 0.1 if delta else 0.0,
                                  t_grad, d_grad)
    loss_c.backward()
    grp = lm.flatten(accumulate_table_grad_in_place=True)
    xp = pts.clone().requires_grad_(True)
    loss_p = lm.gs_sdf_coupling(xp, ids, w_all, 1e-3, delta if delta else None, 0.1 if delta else 0.0)
    with sdf.grad_sinks_armed():
        loss_p.backward()
    assert_close(loss_c, loss_p, 1e-5, "loss")
    assert_close(xc.grad, xp.grad, 1e-5, "d/d samples")
    nt = enc.params_.numel()
    assert_close(t_grad, grp.flat_grad[:nt], 1e-4 if delta else 1e-5, "table gradient (in place)")
    assert_close(d_grad, grp.flat_grad[nt:nt + d_grad.numel()], 1e-4, "decoder gradient (in place)")   # fp32 atomics: launch-to-launch order


def test_joint_sdf_node_first_order_at_forward_time_is_the_same_node(host):
    """gsdf_extras::joint_sdf_loss_analytic with first_order_in_forward runs its first-order chain (gsdf_sdf_data_term_grad -> one-pass decoder backward ->
    Jacobian contraction) inside forward(), ahead of the stencil rows' decoder pass and the loss kernel: the samples' gradient leaves earlier, nothing
    else changes.  Deterministic mode: loss, d/d samples and every parameter gradient are the SAME BITS as the node that does it all in backward(), and
    (itself compared with the Python mirror by tests/test_gpu_cpp_model.py).  Also: gsdf_sdf_data_term_grad writes the v_attr of gsdf_sdf_analytic_loss bit for bit."""
    import gs_sdf_amd.capi as capi
    import gs_sdf_amd.sdf as sdf
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(16)
    M, n_ids, n_ray = 60000, 41000, 30000
    ray = ((torch.rand(n_ray, 3, generator=g) - 0.5) * 3.6).to(dev)
    gt = (torch.randn(n_ray, 1, generator=g) * 0.05).to(dev)
    pts = ((torch.rand(M, 3, generator=g) - 0.5) * 3.6).to(dev)
    ids = torch.randperm(M, generator=g)[:n_ids].sort().values.to(dev)
    w_all = torch.rand(M, 1, generator=g).to(dev)
    lm = sdf.LocalMap([0.1, -0.2, 0.3], 4.0, bce_sigma=0.02, decoder_implementation=1, device=dev, seed=6)
    with torch.no_grad():
        lm.encoder.params_.copy_(((torch.rand(lm.encoder.params_.numel(), generator=torch.Generator().manual_seed(1)) * 2 - 1) * 0.05).to(dev))
    enc = host.TCNNEncoding(16, 2, 19, 32, 2.0)
    enc.params_ = lm.encoder.params_.detach().clone()
    dec = host.TCNNNetwork(32, 2, 64, 3)
    dec.params_ = lm.decoder.params_.detach().clone()
    res = []
    with capi.deterministic():
        for unit in (False, True):
            t_grad, d_grad = torch.zeros_like(enc.params_), torch.zeros_like(dec.params_)
            x = pts.clone().requires_grad_(True)
            loss = host.joint_sdf_loss_analytic(ray, gt, x, ids, w_all, enc, dec, [float(v) for v in lm._origin], float(lm.map_size_inv),
                                                1.0 / 0.02, 1.0, 1e-2, 0.02, 0.1, 0.1, t_grad, d_grad, None, unit_upstream=True, first_order_in_forward=unit)
            if unit:   # the first order has been accumulated already: d/d samples is complete before backward() is called
                torch.cuda.synchronize()
                assert float(d_grad.abs().sum()) > 0
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), x.grad.clone(), t_grad, d_grad))
    for k, what in enumerate(("loss", "d/d samples", "table gradient", "decoder gradient")):
        assert torch.equal(res[0][k], res[1][k]), f"{what}: the forward-time first order is not the backward-time one bit for bit"
    assert float(res[1][1].abs().sum()) > 0 and float(res[1][2].abs().sum()) > 0
    # the data-term gradient kernel against the loss kernel's v_attr
    n, ld = 50000, 2
    attr = (torch.randn(7 * n, ld, generator=g) * 0.1).to(dev)
    g0 = torch.randn(n, 32, generator=g).to(dev)
    jac = torch.randn(n, 32, 3, generator=g).to(dev)
    gts = (torch.randn(20000, generator=g) * 0.05).to(dev)
    wts = torch.rand(n - 20000, generator=g).to(dev)
    loss, va, vb = torch.zeros(1, device=dev), torch.empty(n, ld, device=dev), torch.full((n, ld), 7.0, device=dev)
    vvx, u0 = torch.empty(n, 3, device=dev), torch.empty(n, 32, device=dev)
    p = lambda t: t.data_ptr()
    L, st = capi.lib(), capi.stream()
    with capi.deterministic():      # (the loss value's reduction order is fixed: the two launches give the same bits)
        capi.check(L.gsdf_sdf_analytic_loss(n, 20000, 1, p(attr), ld, p(g0), 32, p(jac), p(gts), p(wts), None, 50.0, 1.0, 1e-2, 0.25, 0.02, 0.1, 0.1,
                                             p(loss), p(va), p(vvx), p(u0), st), "sdf_analytic_loss")
        capi.check(L.gsdf_sdf_data_term_grad(n, 20000, p(attr), ld, p(gts), p(wts), None, 50.0, 1.0, 1e-2, p(vb), st), "sdf_data_term_grad")
        torch.cuda.synchronize()
        assert torch.equal(va, vb)
        loss2 = torch.zeros(1, device=dev)
        capi.check(L.gsdf_sdf_analytic_loss(n, 20000, 1, p(attr), ld, p(g0), 32, p(jac), p(gts), p(wts), None, 50.0, 1.0, 1e-2, 0.25, 0.02, 0.1, 0.1,
                                             p(loss2), None, p(vvx), p(u0), st), "sdf_analytic_loss")      # v_attr = NULL: not written
        torch.cuda.synchronize()
    assert torch.equal(loss2, loss) and float(loss) != 0.0


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5000, 200_000])
def test_distCUDA2(host, oracle, N):
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N)
    pts = torch.rand(N, 3, generator=g) * torch.tensor([4.0, 1.0, 0.3])
    if N >= 5000:
        pts[: N // 4] = pts[: N // 4] * 0.01 + 0.5                      # a dense cluster + duplicates
        pts[10] = pts[11]
    got_cpp = host.distCUDA2(pts.to(dev))
    got_py = ops.distCUDA2(pts.to(dev))
    assert torch.equal(got_cpp, got_py)
    if N <= 5000:
        ref = oracle.knn_mean_dist2(pts.numpy(), prec="f64")
    else:
        from scipy.spatial import cKDTree
        d, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=4)
        ref = (d[:, 1:] ** 2).mean(1)
    assert_close(got_cpp, ref, 1e-4, "distCUDA2")


@pytest.mark.parametrize("kind", ["planar", "collinear", "single_point_cloud"])
def test_distCUDA2_degenerate_clouds(oracle, kind):
    """Exactly planar / collinear / coincident point sets: the uniform grid must be sized from the non-degenerate extents
    (a cell size derived from a zero extent would ask for ~1e12 cells and overflow the cell tables)."""
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    N = 3000
    pts = torch.rand(N, 3, generator=g) * torch.tensor([4.0, 2.0, 1.0])
    if kind == "planar":
        pts[:, 2] = 0.25
    elif kind == "collinear":
        pts[:, 1:] = torch.tensor([0.5, -1.0])
    else:
        pts[:] = torch.tensor([1.0, 2.0, 3.0])
    got = ops.distCUDA2(pts.to(dev))
    ref = oracle.knn_mean_dist2(pts.numpy(), prec="f64")
    assert torch.isfinite(got).all()
    assert_close(got, ref, 1e-4, f"distCUDA2 ({kind})")


def test_octree_as_cpp_matches_python_mirror_and_oracle(host, oracle):
    """kaolin_wisp_cpp drop-in headers (spc_ops.h, octree_as.h) driven the way sub_map.cpp:22-35 / local_map.cpp:467-476
    do: quantize -> unique -> neighbours -> clamp -> from_quantized_points; query; voxel ray march."""
    from gs_sdf_amd.occupancy import OctreeAS, spc_ops
    dev = torch.device("cuda:0")
    L = 7
    rng = np.random.default_rng(5)
    u = rng.standard_normal((20000, 3))
    pts = (0.55 * u / np.linalg.norm(u, axis=1, keepdims=True)).astype(np.float32)
    x = torch.from_numpy(pts).to(dev)
    q = host.quantize_points(x, L)
    assert q.dtype == torch.int16 and torch.equal(q, spc_ops.quantize_points(x, L))
    q = torch.unique(q.contiguous(), dim=0)
    nb = host.points_to_neighbors(q)
    assert nb.shape == (q.shape[0], 27, 3) and torch.equal(nb, spc_ops.points_to_neighbors(q))
    assert torch.equal(host.points_to_corners(q), spc_ops.points_to_corners(q))
    qn = nb.view(-1, 3).clamp(0, 2 ** L - 1)
    acc = host.OctreeAS.from_quantized_points(qn, L)
    ref = oracle.occ_build(L, pts, True)                            # same set of voxels: dilated quantised points
    assert np.array_equal(acc.grid_.cpu().numpy().view(np.uint32), ref)
    py = OctreeAS.from_points(x, L, dilate27=True)
    assert torch.equal(acc.grid_, py.grid)
    qp = (torch.rand(50000, 3, generator=torch.Generator().manual_seed(1)) * 2.2 - 1.1).to(dev)
    for level in (-1, 4):
        assert torch.equal(acc.query(qp, level), py.query(qp, level).pidx)
    vox = acc.get_quantized_points()
    assert torch.equal(vox, py.get_quantized_points()) and vox.dtype == torch.int16
    assert torch.equal(host.quantized_points_to_fpoints(vox, L), spc_ops.quantized_points_to_fpoints(vox, L))
    o = (torch.rand(3000, 3, generator=torch.Generator().manual_seed(2)) * 0.4 - 0.2).to(dev)
    d = torch.nn.functional.normalize(torch.randn(3000, 3, generator=torch.Generator().manual_seed(3)), dim=1).to(dev)
    ridx, samples, depth = acc.raymarch(o, d, "voxel", 2)
    r2 = py.raymarch(o, d, "voxel", 2)
    assert ridx.numel() > 6000 and torch.equal(ridx, r2.ridx) and torch.equal(samples, r2.samples) and torch.equal(depth, r2.depth_samples)
    first = host.mark_pack_boundaries(ridx)
    assert int(first.sum()) == int(torch.unique(ridx).numel()) and bool(first[0])
    with pytest.raises(RuntimeError):
        acc.raymarch(o, d, "ray", 2)


def test_cpp_fused_extras_match_python_mirror(host):
    """gsdf_extras/gsdf_extras.h (libgsdf_torch.so): the fused pieces of the training step that are not submodule symbols —
    photometric loss, SDF query points / ray loss / GS-sample loss with eikonal, update_state, splat activations, fused
    Adam — give the Python mirror's results (same kernels underneath), values and gradients."""
    import gs_sdf_amd.ops as ops
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.neural_gs import update_densify_state
    from gs_sdf_amd.trainer import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    H, W = 72, 130
    img = torch.rand(H, W, 3, generator=g).to(dev)
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    a, b = img.clone().requires_grad_(True), img.clone().requires_grad_(True)
    la, lb = host.l1_dssim_loss(a, gt, 0.8, 0.2), ops.l1_dssim_loss(b, gt, 0.8, 0.2)
    (2.0 * la).backward(); (2.0 * lb).backward()
    assert torch.allclose(la.detach(), lb.detach(), rtol=1e-6) and torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-12)
    # SDF pieces
    n = 3000
    xyz = ((torch.rand(n, 3, generator=g) - 0.5) * 6.0).to(dev)
    lm = sdfm.LocalMap([0.1, 0.2, -0.3], 8.0, decoder_implementation=1, device=dev, seed=9)
    q_py = lm.query_points(xyz, 0.02)
    q_cpp = host.query_points(xyz, [0.1, 0.2, -0.3], 1.0 / 8.0, True, 0.02)
    assert torch.equal(q_py, q_cpp)
    attr = torch.randn(7 * n, 2, generator=g).to(dev)
    gts = (torch.randn(n, 1, generator=g) * 0.05).to(dev)
    a1, a2 = attr.clone().requires_grad_(True), attr.clone().requires_grad_(True)
    l1 = host.sdf_ray_loss(a1, gts, n, 50.0, 0.02, 0.1)
    l2 = sdfm._SdfRayLoss.apply(a2, gts, 50.0, 0.02, 0.1, n)
    l1.backward(); l2.backward()
    assert torch.allclose(l1.detach(), l2.detach(), rtol=1e-5) and torch.equal(a1.grad, a2.grad)   # the scalar is a sum of per-wave atomics
    w_all = torch.rand(5000, generator=g).to(dev)
    ids = torch.randperm(5000, generator=g)[:n].sort().values.to(dev)
    a1, a2 = attr.clone().requires_grad_(True), attr.clone().requires_grad_(True)
    l1 = host.gs_sdf_eik_loss(a1, w_all, ids, n, 1e-3, 0.02, 0.1)
    l1.backward()
    import ctypes as C
    import gs_sdf_amd.capi as capi
    loss = torch.empty((), device=dev); v = torch.empty_like(attr)
    capi.check(capi.lib().gsdf_gs_sdf_eik_loss(n, 1, capi.f32(attr), 2, capi.f32(w_all), capi.ptr(ids, torch.int64), 1e-3, 0.02, 0.1,
                                               capi.f32(loss), capi.f32(v), capi.stream()), "gs_sdf_eik_loss")
    assert torch.allclose(l1.detach(), loss, rtol=1e-5) and torch.equal(a1.grad, v)
    # update_state
    N, M = 4000, 2500
    gid = torch.randperm(N, generator=g)[:M].sort().values.to(dev)
    dens = torch.zeros(M, 2, device=dev, requires_grad=True)
    dens.grad = torch.randn(M, 2, generator=g).to(dev) * 1e-3
    info = dict(gradient_2dgs=dens, n_cameras=torch.tensor([1]), width=torch.tensor([W]), height=torch.tensor([H]), gaussian_ids=gid,
                visibilities=torch.rand(M, 1, generator=g).to(dev), radii=torch.randint(1, 30, (M,), generator=g, dtype=torch.int32).to(dev))
    st_py = {}
    update_densify_state(st_py, info, N, False, True)
    st_cpp = host.update_state({}, dens.grad, gid, info["visibilities"], info["radii"], N, 1, W, H, True)
    for k in ("grad2d", "count", "vis", "radii"):
        assert torch.equal(st_py[k], st_cpp[k]), k
    # activations + fused Adam
    anchors, offsets = torch.randn(N, 3, generator=g).to(dev), (torch.randn(N, 3, generator=g) * 0.01).to(dev).requires_grad_(True)
    scaling, opacity = torch.randn(N, 3, generator=g).to(dev).requires_grad_(True), torch.randn(N, generator=g).to(dev).requires_grad_(True)
    xyz2, sc2, op2 = host.splat_activations(anchors, offsets, scaling, opacity)
    assert torch.equal(xyz2, anchors + offsets)
    assert_close(sc2, torch.exp(scaling), 1e-6, "scales"); assert_close(op2, torch.sigmoid(opacity), 1e-6, "opacities")
    (xyz2.sum() + (sc2 * 2).sum() + (op2 * 3).sum()).backward()
    assert_close(scaling.grad, 2 * torch.exp(scaling.detach()), 1e-6, "d/d log-scales")
    o = torch.sigmoid(opacity.detach())
    assert_close(opacity.grad, 3 * o * (1 - o), 1e-6, "d/d logit-opacity")
    sizes, lrs = [1003, 4096, 7, 2501], [1.6e-4, 5e-3, 5e-2, 1e-3]
    flat = torch.randn(sum(sizes), generator=g).to(dev)
    f1, f2 = flat.clone(), flat.clone()
    g1, g2 = torch.zeros_like(flat), torch.zeros_like(flat)
    o1 = host.FusedAdam(0.9, 0.999, 1e-15); o1.add_group(f1, g1, sizes, lrs)
    o2 = FusedAdam(eps=1e-15); o2.add_group(f2, g2, list(zip(sizes, lrs)))
    for it in range(4):
        gr = torch.randn(flat.numel(), generator=g).to(dev)
        g1.copy_(gr); g2.copy_(gr)
        o1.step(bool(it % 2)); o2.step(zero_grad=bool(it % 2))      # odd steps: the launch also zeroes the gradient it consumed
        assert torch.equal(g1, g2) and bool((g1 == 0).all()) == bool(it % 2)
    assert torch.equal(f1, f2)
    # the step in two launches (JointIteration: the offsets' gradient arrives last): everything behind the first segment(s), then the head — the same bits,
    # whatever the alignment of the boundary (1003 and 1003 + 4096 are not multiples of 4) and with empty segments in the list
    for sizes2, heads in (([1003, 4096, 7, 2501], (1, 2)), ([1203, 1203, 1604, 401, 1203, 0], (1,)), ([8, 0, 5], (1, 2))):
        lrs2 = [1e-3 * (k + 1) for k in range(len(sizes2))]
        flat = torch.randn(sum(sizes2), generator=g).to(dev)
        for head in heads:
            fa, fb = flat.clone(), flat.clone()
            ga, gb = torch.zeros_like(flat), torch.zeros_like(flat)
            oa = host.FusedAdam(0.9, 0.999, 1e-15); oa.add_group(fa, ga, sizes2, lrs2)
            ob = host.FusedAdam(0.9, 0.999, 1e-15); ob.add_group(fb, gb, sizes2, lrs2)
            for it in range(3):
                gr = torch.randn(flat.numel(), generator=g).to(dev)
                ga.copy_(gr); gb.copy_(gr)
                oa.step(bool(it % 2))
                ob.step_tail(head, bool(it % 2)); ob.step_head(head, bool(it % 2))
                assert torch.equal(ga, gb)
            assert torch.equal(fa, fb), (sizes2, head)


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_spherical_harmonics_encoding_is_the_view_colour_basis(host, oracle, degree):
    """SHEncoding (the reference's include/neural_net/encodings/encodings.h:6-27: TCNNEncoding with otype SphericalHarmonics): degree^2 outputs,
    the basis of the splat colours (SPEC A.2: oracle view_colors_fwd with one-hot coefficients evaluates it), differentiable in its input."""
    dev = torch.device("cuda:0")
    enc = host.TCNNEncoding.spherical_harmonics(degree)
    assert enc.get_out_dim() == degree * degree and enc.params_.numel() == 0
    g = torch.Generator().manual_seed(degree)
    d = torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=-1)
    x = ((d + 1.0) * 0.5).to(dev).requires_grad_(True)                       # tiny-cuda-nn's convention: the input is (direction + 1) / 2
    y = enc.forward(x)
    assert y.shape == (500, degree * degree)
    # oracle: colour = max(sum_k basis_k(dir) c_k + 0.5, 0) with dir = mean - camera position (camera at the origin: viewmat = I)
    K = degree * degree
    vm = np.eye(4, dtype=np.float32)[None]
    cam, gid = np.zeros(500, np.int64), np.arange(500, dtype=np.int64)
    for k in range(K):
        sh = np.zeros((500, K, 3), np.float32)
        sh[:, k, 0] = 1.0
        col = oracle.view_colors_fwd(vm, d.numpy().astype(np.float32), sh, cam, gid, degree - 1, prec="f64")[:, 0]
        want = np.maximum(y[:, k].detach().cpu().double().numpy() + 0.5, 0.0)
        np.testing.assert_allclose(col, want, rtol=0, atol=2e-6, err_msg=f"basis {k}")
    if degree == 1:
        return                                                               # a constant: nothing to differentiate
    (gx,) = torch.autograd.grad((y * torch.arange(1, K + 1, device=dev)).sum(), x, create_graph=True)
    assert gx.shape == x.shape and bool(torch.isfinite(gx).all())
    if degree > 2:
        (ggx,) = torch.autograd.grad(gx.square().sum(), x)                   # second order exists (autograd of the composition)
        assert float(ggx.abs().sum()) > 0
