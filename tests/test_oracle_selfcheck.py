"""The oracle checks itself (parity is unpinned: the reference vendors neither the kernels nor
any golden vector — SURVEY.md section 0/8c).  Three independent lines of evidence:
  1. an autograd restatement in torch fp64 (tests/torch_ref.py) agrees with the C oracle's forward
     AND with every hand-derived VJP;
  2. the f32 and f64 builds of the oracle agree;
  3. analytic special cases (fronto-parallel splat -> isotropic footprint; invariants of the bins).
"""
import numpy as np
import pytest
import torch

import gs_sdf_amd.synth as synth
import torch_ref as tr


def _scene(N=300, W=64, H=48, sh_degree=2, seed=0):
    sc = synth.make_scene(N, W, H, sh_degree=sh_degree, seed=seed, sigma_px=(0.7, 5.0))
    vm = synth.make_views(2, seed=1)[1:2]
    return sc, vm


def _np(t):
    return t.detach().numpy()


def test_projection_fwd_bwd_matches_autograd(oracle):
    sc, vm = _scene()
    W, H = sc["W"], sc["H"]
    means, quats, scales = sc["means"].double(), sc["quats"].double(), sc["log_scales"].exp().double()
    K, vmd = sc["K"].double(), vm.double()
    seed = 1234
    p = oracle.projection_2dgs_fwd(_np(means), _np(quats), _np(scales), _np(vmd), _np(K), W, H, seed=seed, prec="f64")
    gid = torch.from_numpy(p["gaussian_ids"])
    M = gid.numel()
    assert 0 < M <= means.shape[0]
    # recover the eps used by the oracle from samples (linear in eps) is awkward: use seed=0 path for
    # autograd of everything else, and check samples separately below.
    p0 = oracle.projection_2dgs_fwd(_np(means), _np(quats), _np(scales), _np(vmd), _np(K), W, H, seed=0, prec="f64")
    a = [t.clone().requires_grad_(True) for t in (means, quats, scales)]
    m2d, dep, Wm, nrm, smp = tr.project(a[0], a[1], a[2], vmd[0], K[0])
    np.testing.assert_allclose(_np(m2d[gid]), p0["means2d"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(_np(dep[gid]), p0["depths"], rtol=1e-12)
    np.testing.assert_allclose(_np(Wm[gid]), p0["ray_transforms"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(_np(nrm[gid]), p0["normals"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(_np(smp[gid]), p0["samples"], rtol=1e-12)
    g = torch.Generator().manual_seed(5)
    v = [torch.randn(M, *s, generator=g, dtype=torch.float64) for s in ((2,), (), (3, 3), (3,), (3,))]
    loss = (m2d[gid] * v[0]).sum() + (dep[gid] * v[1]).sum() + (Wm[gid] * v[2]).sum() + (nrm[gid] * v[3]).sum() + (smp[gid] * v[4]).sum()
    loss.backward()
    vm_, vq_, vs_ = oracle.projection_2dgs_bwd(_np(means), _np(quats), _np(scales), _np(vmd), _np(K), W, H,
                                               p0["camera_ids"], p0["gaussian_ids"], _np(v[0]), _np(v[1]), _np(v[2]),
                                               _np(v[3]), _np(v[4]), seed=0, prec="f64")
    np.testing.assert_allclose(vm_, _np(a[0].grad), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(vq_, _np(a[1].grad), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(vs_, _np(a[2].grad), rtol=1e-8, atol=1e-9)
    # stochastic samples: finite differences on the oracle itself (eps is a fixed function of (seed, id))
    eps = 1e-6
    vsmp = _np(v[4])
    base = (p["samples"] * vsmp).sum()
    gm, gq, gs = oracle.projection_2dgs_bwd(_np(means), _np(quats), _np(scales), _np(vmd), _np(K), W, H,
                                            p["camera_ids"], p["gaussian_ids"], np.zeros((M, 2)), np.zeros(M),
                                            np.zeros((M, 3, 3)), np.zeros((M, 3)), vsmp, seed=seed, prec="f64")
    for arr, grad in ((means, gm), (quats, gq), (scales, gs)):
        for trial in range(4):
            i, j = int(p["gaussian_ids"][trial * 7 % M]), trial % arr.shape[1]
            pert = arr.clone(); pert[i, j] += eps
            args = [means, quats, scales]
            args[[means, quats, scales].index(arr) if False else (0 if arr is means else 1 if arr is quats else 2)] = pert
            pp = oracle.projection_2dgs_fwd(_np(args[0]), _np(args[1]), _np(args[2]), _np(vmd), _np(K), W, H, seed=seed, prec="f64")
            if pp["gaussian_ids"].shape[0] != M:
                continue
            fd = ((pp["samples"] * vsmp).sum() - base) / eps
            assert abs(fd - grad[i, j]) <= 1e-4 * max(1.0, abs(fd)), (fd, grad[i, j])
    # weights are exp(-0.5|eps|^2) in (0,1]
    assert np.all(p["samples_weights"] > 0) and np.all(p["samples_weights"] <= 1)
    assert np.all(p0["samples_weights"] == 1)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_view_colors_matches_autograd(oracle, deg):
    sc, vm = _scene(sh_degree=3)
    means, sh, vmd = sc["means"].double(), sc["sh"].double(), vm.double()
    N = means.shape[0]
    gid = torch.arange(0, N, 2)
    cam = torch.zeros_like(gid)
    out = oracle.view_colors_fwd(_np(vmd), _np(means), _np(sh), _np(cam), _np(gid), deg, prec="f64")
    a_m, a_sh = means.clone().requires_grad_(True), sh.clone().requires_grad_(True)
    campos = torch.linalg.inv(vmd[0])[:3, 3]
    ref = tr.sh_colors(deg, a_m[gid] - campos, a_sh[gid])
    np.testing.assert_allclose(out, _np(ref), rtol=1e-10, atol=1e-12)
    v = torch.randn(gid.numel(), 3, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    (ref * v).sum().backward()
    v_sh, v_means = oracle.view_colors_bwd(_np(vmd), _np(means), _np(sh), _np(cam), _np(gid), deg, _np(v), prec="f64")
    np.testing.assert_allclose(v_sh, _np(a_sh.grad), rtol=1e-9, atol=1e-12)
    ref_vm = np.zeros_like(v_means) if a_m.grad is None else _np(a_m.grad)   # degree 0 is view independent
    np.testing.assert_allclose(v_means, ref_vm, rtol=1e-8, atol=1e-11)


def _pipeline(oracle, sc, vm, prec, seed=0):
    W, H = sc["W"], sc["H"]
    dt = np.float32 if prec == "f32" else np.float64
    means, quats = _np(sc["means"]).astype(dt), _np(sc["quats"]).astype(dt)
    scales, opac = np.exp(_np(sc["log_scales"]).astype(dt)), 1 / (1 + np.exp(-_np(sc["logit_opacities"]).astype(dt)))
    p = oracle.projection_2dgs_fwd(means, quats, scales, _np(vm), _np(sc["K"]), W, H, seed=seed, prec=prec)
    col = oracle.view_colors_fwd(_np(vm), means, _np(sc["sh"]), p["camera_ids"], p["gaussian_ids"], sc["sh_degree"], prec=prec)
    tpg, ids, flat, offs = oracle.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1, prec=prec)
    return p, col, opac[p["gaussian_ids"]], tpg, ids, flat, offs


def test_binning_invariants(oracle):
    sc, vm = _scene(N=2000, W=160, H=112)
    p, col, opa, tpg, ids, flat, offs = _pipeline(oracle, sc, vm, "f32")
    I = flat.shape[0]
    assert tpg.sum() == I and I > 0
    assert np.all(np.diff(ids.view(np.uint64)) >= 0)                    # sorted by (tile, depth bits)
    o = offs.reshape(-1)
    assert np.all(np.diff(o) >= 0) and o[0] == 0 and o[-1] <= I          # offsets monotone
    tile_of = (ids >> 32)
    for t in (0, 7, len(o) - 1):
        lo, hi = o[t], (o[t + 1] if t + 1 < len(o) else I)
        assert np.all(tile_of[lo:hi] == t)
    # stable: equal keys keep emission (= packed index) order
    same = np.diff(ids) == 0
    assert np.all(flat[1:][same] > flat[:-1][same])
    # depth bits of the key are the fp32 depth
    np.testing.assert_array_equal((ids & 0xFFFFFFFF).astype(np.uint32), p["depths"].astype(np.float32).view(np.uint32)[flat])


def test_raster_fwd_bwd_matches_autograd(oracle):
    sc, vm = _scene(N=400, W=48, H=40, sh_degree=1, seed=3)
    W, H = sc["W"], sc["H"]
    p, col, opa, tpg, ids, flat, offs = _pipeline(oracle, sc, vm, "f64")
    bg = np.array([[0.3, 0.5, 0.7]])
    fw = oracle.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat,
                                   backgrounds=bg, prec="f64")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double().requires_grad_(True)
    a = [t(p["means2d"]), t(p["ray_transforms"]), t(col), t(opa), t(p["normals"])]
    rc, rd, ra, rn, rm = tr.rasterize(*a, W, H, 16, torch.from_numpy(offs[0]), torch.from_numpy(flat), background=torch.from_numpy(bg[0]))
    for ref, key in ((rc, "render_colors"), (rd, "render_depths"), (ra, "render_alphas"), (rn, "render_normals"), (rm, "render_median")):
        np.testing.assert_allclose(fw[key][0], _np(ref), rtol=1e-9, atol=1e-11, err_msg=key)
    assert (fw["render_alphas"] > 0.5).mean() > 0.2          # the scene actually covers pixels
    ug = synth.upstream_grads(H, W, seed=2)
    v = {k: _np(x.double()) for k, x in ug.items()}
    loss = ((rc * ug["v_render_colors"][0].double()).sum() + (rd * ug["v_render_depths"][0].double()).sum()
            + (ra * ug["v_render_alphas"][0].double()).sum() + (rn * ug["v_render_normals"][0].double()).sum()
            + (rm * ug["v_render_median"][0].double()).sum())
    loss.backward()
    g = oracle.rasterize_2dgs_bwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat,
                                  fw["render_alphas"], fw["last_ids"], fw["median_ids"], v["v_render_colors"],
                                  v["v_render_depths"], v["v_render_alphas"], v["v_render_normals"], v["v_render_median"],
                                  backgrounds=bg, prec="f64")
    for got, ref, name in ((g["v_means2d"], a[0].grad, "means2d"), (g["v_ray_transforms"], a[1].grad, "ray_transforms"),
                           (g["v_colors"], a[2].grad, "colors"), (g["v_opacities"], a[3].grad, "opacities"),
                           (g["v_normals"], a[4].grad, "normals")):
        ref = _np(ref)
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-9 * scale, err_msg=name)
    # densification signal == dL/dM[.,2] * depth (2DGS convention), abs-grad >= |grad|
    np.testing.assert_allclose(g["v_densify"], g["v_ray_transforms"][:, :2, 2] * p["ray_transforms"][:, 2, 2:3], rtol=1e-9, atol=1e-12)
    assert np.all(g["v_means2d_abs"] + 1e-12 >= np.abs(g["v_means2d"]))
    # visibilities: max blending weight, in [0, 0.999]
    assert fw["visibilities"].max() <= 0.999 + 1e-9 and fw["visibilities"].min() >= 0


def test_f32_vs_f64_builds_agree(oracle):
    sc, vm = _scene(N=1500, W=96, H=80, sh_degree=0, seed=4)
    W, H = sc["W"], sc["H"]
    r = {}
    for prec in ("f32", "f64"):
        p, col, opa, tpg, ids, flat, offs = _pipeline(oracle, sc, vm, prec)
        fw = oracle.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, prec=prec)
        r[prec] = (p, fw, flat)
    p32, f32, fl32 = r["f32"]; p64, f64, fl64 = r["f64"]
    if p32["gaussian_ids"].shape == p64["gaussian_ids"].shape and np.array_equal(p32["radii"], p64["radii"]):
        np.testing.assert_allclose(p32["ray_transforms"], p64["ray_transforms"], rtol=2e-5, atol=1e-5)
    # images agree to fp32 round-off even if a radius flipped by one pixel
    np.testing.assert_allclose(f32["render_colors"], f64["render_colors"], atol=2e-4)
    np.testing.assert_allclose(f32["render_alphas"], f64["render_alphas"], atol=2e-4)


def test_fronto_parallel_splat_is_isotropic(oracle):
    """A splat facing the camera at depth z with scale s projects to an isotropic Gaussian of
    sigma_px = s*f/z: alpha(centre)=o, alpha(r)=o*exp(-r^2/(2 sigma^2))."""
    W = H = 64
    f, z, s, o = 50.0, 4.0, 0.4, 0.8
    K = np.array([[[f, 0, 31.5], [0, f, 31.5], [0, 0, 1]]])
    means = np.array([[0.0, 0.0, z]])
    quats = np.array([[1.0, 0, 0, 0]])
    scales = np.array([[s, s, s]])
    vm = np.eye(4)[None]
    p = oracle.projection_2dgs_fwd(means, quats, scales, vm, K, W, H, prec="f64")
    sig = s * f / z
    assert p["radii"][0] == int(np.ceil(3 * sig))
    np.testing.assert_allclose(p["means2d"][0], [31.5, 31.5], atol=1e-9)
    tpg, ids, flat, offs = oracle.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1, prec="f64")
    fw = oracle.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], np.ones((1, 3)), np.array([o]), p["normals"], W, H, 16,
                                   offs, flat, prec="f64")
    yy, xx = np.mgrid[0:H, 0:W]
    r2 = (xx + 0.5 - 31.5) ** 2 + (yy + 0.5 - 31.5) ** 2
    # exact ray-splat intersection == screen distance / sigma for a fronto-parallel splat on the axis
    expect = o * np.exp(-0.5 * np.minimum(r2 / sig ** 2, 2 * r2))
    expect[expect < 1 / 255] = 0
    # only the tiles hit by the 3-sigma box [16.5,46.5] are composited (binning is part of the operator)
    cov = np.zeros_like(expect, dtype=bool); cov[16:48, 16:48] = True
    expect[~cov] = 0
    np.testing.assert_allclose(fw["render_alphas"][0, :, :, 0], expect, atol=1e-9)
    np.testing.assert_allclose(fw["render_normals"][0, 32, 32] / max(fw["render_alphas"][0, 32, 32, 0], 1e-9), [0, 0, -1], atol=1e-9)
    np.testing.assert_allclose(fw["render_depths"][0, 32, 32, 0] / fw["render_alphas"][0, 32, 32, 0], z, rtol=1e-6)


def test_threaded_projection_equals_the_serial_row_loop(oracle):
    """The projection runs on all cores (two passes forward; camera blocks backward) for the cpu_baseline leg.  Forward: rows in the serial
    loop's order (camera-major, gaussian ascending), one thread or many.  Backward: rows handed over in any OTHER order take the serial loop;
    with two cameras both orders add the same two terms per gaussian, so the sums must agree bit for bit."""
    sc = synth.make_scene(4000, 160, 96, sh_degree=0, seed=5, sigma_px=(0.7, 5.0))
    vm = synth.make_views(3, seed=2)[1:3]
    n = lambda t: t.detach().numpy()
    means, quats, scales, K = n(sc["means"]), n(sc["quats"]), n(sc["log_scales"].exp()), n(sc["K"]).repeat(2, 0)
    oracle.set_threads(4)
    p4 = oracle.projection_2dgs_fwd(means, quats, scales, n(vm), K, 160, 96)
    oracle.set_threads(1)
    p1 = oracle.projection_2dgs_fwd(means, quats, scales, n(vm), K, 160, 96)
    for k in p1:
        assert np.array_equal(p1[k], p4[k]), k
    cam, gid = p1["camera_ids"], p1["gaussian_ids"]
    M = len(gid)
    assert len(set(cam.tolist())) == 2 and M > 4000
    order = np.lexsort((gid, cam))
    assert np.array_equal(order, np.arange(M))                       # camera-major, gaussian ascending
    g = np.random.default_rng(0)
    v2d, vd = g.standard_normal((M, 2)).astype(np.float32), g.standard_normal(M).astype(np.float32)
    vrt, vn = g.standard_normal((M, 3, 3)).astype(np.float32), g.standard_normal((M, 3)).astype(np.float32)
    oracle.set_threads(4)
    a = oracle.projection_2dgs_bwd(means, quats, scales, n(vm), K, 160, 96, cam, gid, v2d, vd, vrt, vn)
    r = np.arange(M)[::-1].copy()                                    # reversed rows: the serial loop
    b = oracle.projection_2dgs_bwd(means, quats, scales, n(vm), K, 160, 96, cam[r].copy(), gid[r].copy(), v2d[r].copy(), vd[r].copy(),
                                   vrt[r].copy(), vn[r].copy())
    oracle.set_threads(int(__import__("os").environ.get("GSDF_TEST_THREADS", "0")) or 8)      # what the rest of the session runs on
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
