"""tests/golden/*.npz: byte-stable outputs of the repo's OWN CPU oracle on tiny seeded inputs (tests/golden/make_golden.py;
NOT reference outputs — the reference has none for this path, parity is unpinned).
  * CPU: today's oracle still reproduces them (catches silent semantic drift of the checker between rounds);
  * GPU: the HIP path, through the C ABI, agrees with the frozen files (integer tensors bit-exact, fp32 within 1e-4)."""
import os

import numpy as np
import pytest
import torch

from util import assert_close, assert_equal_int

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def test_oracle_reproduces_splat_golden(oracle):
    g = _load("splat_cfg_tiny")
    W, H, deg = int(g["W"]), int(g["H"]), int(g["deg"])
    p = oracle.projection_2dgs_fwd(g["means"], g["quats"], g["scales"], g["viewmat"], g["K"], W, H, seed=0, prec="f32")
    for k, v in p.items():
        if v.dtype.kind in "iu":
            assert np.array_equal(v, g["p_" + k]), k
        else:
            np.testing.assert_allclose(v, g["p_" + k], rtol=1e-6, atol=1e-7, err_msg=k)
    cols = oracle.view_colors_fwd(g["viewmat"], g["means"], g["sh"], p["camera_ids"], p["gaussian_ids"], deg, prec="f32")
    np.testing.assert_allclose(cols, g["view_colors"], rtol=1e-6, atol=1e-7)
    tpg, ids, flat, offs = oracle.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1)
    for got, key in ((tpg, "tiles_per_gauss"), (ids, "isect_ids"), (flat, "flatten_ids"), (offs, "isect_offsets")):
        assert np.array_equal(got, g[key]), key
    r = oracle.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], cols, g["opacities"][p["gaussian_ids"]], p["normals"],
                                  W, H, 16, offs, flat)
    for k, v in r.items():
        if v.dtype.kind in "iu":
            assert (v != g["r_" + k]).mean() < 1e-3, k          # a threshold flip moves a last/median id
        else:
            assert_close(v, g["r_" + k], 1e-5, k, outlier_frac=1e-4)


def test_oracle_reproduces_sdf_and_occupancy_golden(oracle):
    g = _load("sdf_tiny")
    total = oracle.grid_offsets(dict(oracle.GRID_DEFAULT))[-1]
    idx = np.arange(total * 2, dtype=np.uint64)
    table = (((idx * np.uint64(2654435761)) % np.uint64(1 << 20)).astype(np.float32) / np.float32(1 << 20) - 0.5).astype(np.float32) * 2e-1
    feat = oracle.grid_fwd(g["x"], table, dict(oracle.GRID_DEFAULT), prec="f32")
    np.testing.assert_allclose(feat, g["feat"], rtol=1e-6, atol=1e-8)
    out = oracle.mlp_fwd(feat, [int(d) for d in g["dims"]], g["mlp_w"], None, prec="f32")
    np.testing.assert_allclose(out, g["mlp_out"], rtol=1e-5, atol=1e-7)
    o = _load("occ_tiny")
    L = int(o["L"])
    grid = oracle.occ_build(L, o["pts"], True)
    assert np.array_equal(grid, o["grid"]) and np.array_equal(oracle.occ_list(L, grid), o["voxels"])
    assert np.array_equal(oracle.occ_query(L, grid, o["q"]), o["q_mask"]) and np.array_equal(oracle.occ_query(L, grid, o["q"], 3), o["q_mask_l3"])
    counts, ridx, samples, depth = oracle.occ_raymarch(L, grid, o["origins"], o["dirs"], 2)
    assert np.array_equal(counts, o["counts"]) and np.array_equal(ridx, o["ridx"])
    assert np.array_equal(depth, o["depth"]) and np.array_equal(samples, o["samples"])


@pytest.mark.gpu
def test_hip_path_matches_splat_golden():
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    g = _load("splat_cfg_tiny")
    W, H, deg = int(g["W"]), int(g["H"]), int(g["deg"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    means, quats, scales, vm, K, sh = (t(g[k]) for k in ("means", "quats", "scales", "viewmat", "K", "sh"))
    Kd = K.expand(1, 3, 3).contiguous()
    cam, gid, radii, m2d, dep, rt, nrm, smp, sw = ops.fully_fused_projection_2dgs(means, quats, scales, vm, Kd, W, H, 0.05, 300.0,
                                                                                  0.0, True, False, 0)
    assert_equal_int(cam, g["p_camera_ids"], "camera_ids"); assert_equal_int(gid, g["p_gaussian_ids"], "gaussian_ids")
    assert_equal_int(radii, g["p_radii"], "radii")
    for got, key in ((m2d, "means2d"), (dep, "depths"), (rt, "ray_transforms"), (nrm, "normals")):
        assert np.array_equal(got.cpu().numpy(), g["p_" + key]), key
    col = ops.get_view_colors(vm, means, radii, sh, cam, gid, deg)
    assert_close(col, g["view_colors"], 1e-5, "view colors")
    tpg, flat, offs, ids = ops.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid, return_isect_ids=True)
    assert_equal_int(tpg, g["tiles_per_gauss"], "tiles_per_gauss"); assert_equal_int(ids, g["isect_ids"], "isect_ids")
    assert_equal_int(flat, g["flatten_ids"], "flatten_ids"); assert_equal_int(offs, g["isect_offsets"], "isect_offsets")
    opa = t(g["opacities"])[gid]
    z = torch.zeros_like(m2d)
    rc, rd, ra, rn, _, rm, vis = ops.rasterize_to_pixels_2dgs(m2d, rt, col, opa, nrm, z, W, H, 16, offs, flat, None, None, True, z, False)
    for got, key in ((rc, "render_colors"), (rd, "render_depths"), (ra, "render_alphas"), (rn, "render_normals"), (vis, "visibilities")):
        assert_close(got, g["r_" + key], 1e-4, key, outlier_frac=1e-4)


@pytest.mark.gpu
def test_hip_path_matches_sdf_and_occupancy_golden():
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.occupancy import OctreeAS
    dev = torch.device("cuda:0")
    g = _load("sdf_tiny")
    enc = sdfm.TCNNEncoding(3, None, "enc", dev, seed=0)
    n = enc.params_.numel()
    idx = np.arange(n, dtype=np.uint64)
    table = (((idx * np.uint64(2654435761)) % np.uint64(1 << 20)).astype(np.float32) / np.float32(1 << 20) - 0.5).astype(np.float32) * 2e-1
    with torch.no_grad():
        enc.params_.copy_(torch.from_numpy(table).to(dev))
    feat = enc.forward(torch.from_numpy(g["x"]).to(dev))
    assert_close(feat, g["feat"], 1e-5, "hash-grid features")
    dec = sdfm.TCNNNetwork(32, 2, dict(n_neurons=64, n_hidden_layers=3), "dec", dev, seed=0)
    with torch.no_grad():
        dec.params_.copy_(torch.from_numpy(g["mlp_w"]).to(dev))
    assert_close(dec.forward(torch.from_numpy(g["feat"]).to(dev)), g["mlp_out"], 1e-4, "decoder output")
    o = _load("occ_tiny")
    L = int(o["L"])
    acc = OctreeAS.from_points(torch.from_numpy(o["pts"]).to(dev), L, dilate27=True)
    assert np.array_equal(acc.grid.cpu().numpy().view(np.uint32), o["grid"])
    assert np.array_equal(acc.get_quantized_points().cpu().numpy(), o["voxels"])
    assert np.array_equal(acc.query(torch.from_numpy(o["q"]).to(dev)).pidx.cpu().numpy() > -1, o["q_mask"].astype(bool))
    assert np.array_equal(acc.query(torch.from_numpy(o["q"]).to(dev), 3).pidx.cpu().numpy() > -1, o["q_mask_l3"].astype(bool))
    rm = acc.raymarch(torch.from_numpy(o["origins"]).to(dev), torch.from_numpy(o["dirs"]).to(dev), "voxel", 2)
    assert np.array_equal(rm.ridx.cpu().numpy(), o["ridx"].astype(np.int64))
    assert np.array_equal(rm.depth_samples.cpu().numpy(), o["depth"]) and np.array_equal(rm.samples.cpu().numpy(), o["samples"])
