"""Diagnostic (GPU): error distribution of the HIP compositing vs the f32 and f64 oracles."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gs_sdf_amd.synth as synth, gs_sdf_amd.ops as ops
from oracle import oracle as orc
orc.build()
dev = torch.device("cuda:0")
n = lambda t: t.detach().cpu().numpy()

def scaled(a, r):
    a = np.asarray(a, np.float64); r = np.asarray(r, np.float64)
    floor = np.abs(r).mean() + 1e-30
    return np.abs(a - r) / np.maximum(np.abs(r), floor)

def q(e):
    return "n>1e-4=%d n>1e-3=%d max=%.2e (size %d)" % ((e > 1e-4).sum(), (e > 1e-3).sum(), e.max(), e.size)

for (N, W, H, deg, V, seed) in [(10_000, 256, 256, 0, 1, 0), (40_000, 640, 368, 0, 1, 3)]:
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=seed, sigma_px=(0.5, 6.0))
    vm = synth.make_views(V + 1, seed=seed + 10)[1:]
    means, quats, scales = n(sc["means"]), n(sc["quats"]), n(sc["log_scales"].exp())
    opac = n(torch.sigmoid(sc["logit_opacities"]))
    p = orc.projection_2dgs_fwd(means, quats, scales, n(vm), n(sc["K"]), W, H)
    col = orc.view_colors_fwd(n(vm), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg)
    opa = opac[p["gaussian_ids"]]
    tpg, ids, flat, offs = orc.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], V)
    ug = synth.upstream_grads(H, W, seed=2, C=V)
    res = {}
    for prec in ("f32", "f64"):
        fw = orc.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, prec=prec)
        g = orc.rasterize_2dgs_bwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat,
                                   fw["render_alphas"], fw["last_ids"], fw["median_ids"], *[n(ug[k]) for k in
                                   ("v_render_colors", "v_render_depths", "v_render_alphas", "v_render_normals", "v_render_median")], prec=prec)
        res[prec] = (fw, g)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(True)
    a = [t(p["means2d"]), t(p["ray_transforms"]), t(col), t(opa), t(p["normals"])]
    dens = torch.zeros_like(a[0], requires_grad=True); absg = torch.zeros_like(a[0], requires_grad=True)
    rc, rd, ra, rn, _, rm, vis = ops.rasterize_to_pixels_2dgs(a[0], a[1], a[2], a[3], a[4], dens, W, H, 16,
                                                             torch.from_numpy(offs).to(dev), torch.from_numpy(flat).to(dev), None, None, True, absg, False)
    loss = sum((o * ug[k].to(dev)).sum() for o, k in ((rc, "v_render_colors"), (rd, "v_render_depths"), (ra, "v_render_alphas"),
                                                      (rn, "v_render_normals"), (rm, "v_render_median")))
    loss.backward()
    gpu_f = dict(render_colors=n(rc), render_depths=n(rd), render_alphas=n(ra), render_normals=n(rn), visibilities=n(vis))
    gpu_g = dict(v_colors=n(a[2].grad), v_normals=n(a[4].grad), v_opacities=n(a[3].grad), v_ray_transforms=n(a[1].grad),
                 v_means2d=n(a[0].grad), v_densify=n(dens.grad), v_means2d_abs=n(absg.grad))
    print(f"=== N={N} {W}x{H}  M={opa.shape[0]} I={flat.shape[0]}")
    for k in gpu_f:
        print(f"{k:18s} gpu-f64: {q(scaled(gpu_f[k], res['f64'][0][k]))} | f32-f64: {q(scaled(res['f32'][0][k], res['f64'][0][k]))}")
    print("last_ids differ gpu/f32:", (n(ops_last := None) if False else 0))
    for k in gpu_g:
        print(f"{k:18s} gpu-f64: {q(scaled(gpu_g[k], res['f64'][1][k]))} | f32-f64: {q(scaled(res['f32'][1][k], res['f64'][1][k]))}")
