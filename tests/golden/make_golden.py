"""Generates tests/golden/*.npz.  THESE ARE NOT REFERENCE OUTPUTS: the reference vendors neither the kernels of this path nor
any fixture (parity unpinned, oracle/README.md).  They are outputs of the repo's own CPU oracle (f32 build) on tiny seeded
inputs, frozen so that (1) an accidental change of the oracle's semantics between rounds is caught by the CPU suite and
(2) the HIP path is also compared against a byte-stable file, not only against an oracle rebuilt from today's sources.
Regenerate ONLY together with a deliberate SPEC change:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc          # noqa: E402
import gs_sdf_amd.synth as synth          # noqa: E402  (CPU-only helpers: scene generator)


def splat_case():
    N, W, H, deg = 300, 64, 48, 1
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=11, sigma_px=(0.7, 5.0))
    vm = synth.make_views(2, seed=1)[1:2].numpy()
    means, quats, scales = sc["means"].numpy(), sc["quats"].numpy(), np.exp(sc["log_scales"].numpy())
    opac = 1.0 / (1.0 + np.exp(-sc["logit_opacities"].numpy()))
    K = sc["K"].numpy()
    p = orc.projection_2dgs_fwd(means, quats, scales, vm, K, W, H, seed=0, prec="f32")
    cols = orc.view_colors_fwd(vm, means, sc["sh"].numpy(), p["camera_ids"], p["gaussian_ids"], deg, prec="f32")
    tpg, ids, flat, offs = orc.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1)
    r = orc.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], cols, opac[p["gaussian_ids"]], p["normals"], W, H, 16, offs, flat)
    out = dict(W=W, H=H, deg=deg, means=means, quats=quats, scales=scales, opacities=opac, sh=sc["sh"].numpy(), viewmat=vm, K=K,
               view_colors=cols, tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=flat, isect_offsets=offs)
    out.update({"p_" + k: v for k, v in p.items()})
    out.update({"r_" + k: v for k, v in r.items()})
    return out


def sdf_case():
    rng = np.random.default_rng(21)
    cfg = dict(orc.GRID_DEFAULT)
    total = orc.grid_offsets(cfg)[-1]
    x = rng.random((64, 3), dtype=np.float32)
    # a sparse, reproducible table: value = hash of the entry index (no 61 MB fixture)
    idx = np.arange(total * 2, dtype=np.uint64)
    table = (((idx * np.uint64(2654435761)) % np.uint64(1 << 20)).astype(np.float32) / np.float32(1 << 20) - 0.5).astype(np.float32) * 2e-1
    feat = orc.grid_fwd(x, table, cfg, prec="f32")
    dims = [32, 64, 64, 64, 2]
    w = (rng.random(sum(a * b for a, b in zip(dims[:-1], dims[1:])), dtype=np.float32) - 0.5) * 0.5
    out = orc.mlp_fwd(feat, dims, w, None, prec="f32")
    return dict(x=x, feat=feat, mlp_w=w, mlp_out=out, dims=np.array(dims))


def occ_case():
    rng = np.random.default_rng(31)
    L = 6
    u = rng.standard_normal((1500, 3))
    pts = (0.6 * u / np.linalg.norm(u, axis=1, keepdims=True)).astype(np.float32)
    grid = orc.occ_build(L, pts, True)
    o = (rng.random((200, 3)) * 2.4 - 1.2).astype(np.float32)
    d = rng.standard_normal((200, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    counts, ridx, samples, depth = orc.occ_raymarch(L, grid, o, d, 2)
    q = (rng.random((500, 3)) * 2.2 - 1.1).astype(np.float32)
    return dict(L=L, pts=pts, grid=grid, origins=o, dirs=d, counts=counts, ridx=ridx, samples=samples, depth=depth, q=q,
                q_mask=orc.occ_query(L, grid, q), q_mask_l3=orc.occ_query(L, grid, q, 3), voxels=orc.occ_list(L, grid))


if __name__ == "__main__":
    orc.build()
    for name, fn in (("splat_cfg_tiny", splat_case), ("sdf_tiny", sdf_case), ("occ_tiny", occ_case)):
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fn())
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
