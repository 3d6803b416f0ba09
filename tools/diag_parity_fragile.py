"""Diagnostic (GPU): HIP compositing forward / backward against the oracle's fp64 build, on ALL elements and on the
decision-robust ("clean") elements of oracle.rasterize_2dgs_fragility.  Usage: python tools/diag_parity_fragile.py [shape ...]
Writes gpurun_out/diag_parity_fragile.json and, per shape, the worst clean offenders."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gs_sdf_amd.synth as synth, gs_sdf_amd.ops as ops
from oracle import oracle as orc
orc.build(); orc.set_threads(os.cpu_count() or 1)
dev = torch.device("cuda:0")
n = lambda t: t.detach().cpu().numpy()
SHAPES = {"small": (50_000, 640, 368, 0, False), "cfg1": (300_000, 1200, 680, 0, True), "cfg3": (1_000_000, 1920, 1080, 0, False),
          "cfg4": (3_000_000, 640, 512, 3, False)}
KM = float(os.environ.get("KMARGIN", "16")); COND = float(os.environ.get("COND_ABS", "2e-6"))

def scaled(a, r):
    a = np.asarray(a, np.float64); r = np.asarray(r, np.float64)
    floor = np.abs(r).mean() + 1e-30
    return np.abs(a - r) / np.maximum(np.abs(r), floor)

out = {}
for name in (sys.argv[1:] or ["small", "cfg1"]):
    N, W, H, deg, replica = SHAPES[name]
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica)
    vm = synth.make_views(2, seed=1)[1:2]
    means, quats, scales = n(sc["means"]), n(sc["quats"]), n(sc["log_scales"].exp())
    opac = n(torch.sigmoid(sc["logit_opacities"]))
    p = orc.projection_2dgs_fwd(means, quats, scales, n(vm), n(sc["K"]), W, H)
    col = orc.view_colors_fwd(n(vm), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg)
    opa = opac[p["gaussian_ids"]]
    tpg, ids, flat, offs = orc.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1)
    ug = synth.upstream_grads(H, W, seed=2)
    t0 = time.time()
    fw = orc.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, prec="f64")
    g = orc.rasterize_2dgs_bwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat,
                               fw["render_alphas"], fw["last_ids"], fw["median_ids"], *[n(ug[k]) for k in
                               ("v_render_colors", "v_render_depths", "v_render_alphas", "v_render_normals", "v_render_median")], prec="f64", abs_sums=True)
    pf, sf, cnt = orc.rasterize_2dgs_fragility(p["means2d"], p["ray_transforms"], opa, W, H, 16, offs, flat, kmargin=KM, cond_abs=COND)
    t_or = time.time() - t0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(True)
    a = [t(p["means2d"]), t(p["ray_transforms"]), t(col), t(opa), t(p["normals"])]
    dens = torch.zeros_like(a[0], requires_grad=True)
    rc, rd, ra, rn, _, rm, vis = ops.rasterize_to_pixels_2dgs(a[0], a[1], a[2], a[3], a[4], dens, W, H, 16,
                                                             torch.from_numpy(offs).to(dev), torch.from_numpy(flat).to(dev))
    loss = sum((o * ug[k].to(dev)).sum() for o, k in ((rc, "v_render_colors"), (rd, "v_render_depths"), (ra, "v_render_alphas"),
                                                      (rn, "v_render_normals"), (rm, "v_render_median")))
    loss.backward()
    M = opa.shape[0]
    rec = dict(N=N, W=W, H=H, M=M, I=int(flat.shape[0]), oracle_s=round(t_or, 1), pairs=cnt, kmargin=KM, cond_abs=COND,
               pix_excluded=float((pf != 0).mean()), splat_excluded=float((sf != 0).mean()),
               by_flag={nm: [float(((pf & b) != 0).mean()), float(((sf & b) != 0).mean())] for b, nm in
                        ((1, "alpha"), (2, "term"), (4, "median"), (8, "branch"), (16, "clamp"), (32, "cond"), (64, "edge-on"))})
    print(f"=== {name}: N={N} {W}x{H} M={M} I={flat.shape[0]} oracle {t_or:.1f}s  excluded pixels {rec['pix_excluded']:.5f} splats {rec['splat_excluded']:.5f}")
    print("    ", rec["by_flag"])
    pm = pf[0] == 0
    for k, got in (("render_colors", rc), ("render_depths", rd), ("render_alphas", ra), ("render_normals", rn), ("render_median", rm)):
        e = scaled(n(got), fw[k])[0].max(-1)
        rec[k] = dict(all_above=int((e > 1e-4).sum()), all_max=float(e.max()), clean_above=int((e[pm] > 1e-4).sum()), clean_max=float(e[pm].max()))
        print(f"{k:18s} all: >1e-4 {rec[k]['all_above']:6d} max {rec[k]['all_max']:.2e} | clean: >1e-4 {rec[k]['clean_above']:6d} max {rec[k]['clean_max']:.2e}")
    sm = sf == 0
    for k, got in (("visibilities", vis), ("v_colors", a[2].grad), ("v_opacities", a[3].grad), ("v_normals", a[4].grad), ("v_means2d", a[0].grad),
                   ("v_ray_transforms", a[1].grad), ("v_densify", dens.grad)):
        ref = fw[k] if k == "visibilities" else g[k]
        e = scaled(n(got), ref).reshape(M, -1).max(-1)
        d = (np.asarray(n(got), np.float64) - ref).reshape(M, -1)
        rl2 = float(np.linalg.norm(d[sm]) / (np.linalg.norm(np.asarray(ref).reshape(M, -1)[sm]) + 1e-30))
        rec[k] = dict(all_above=int((e > 1e-4).sum()), all_max=float(e.max()), clean_above=int((e[sm] > 1e-4).sum()), clean_max=float(e[sm].max()),
                      clean_rel_l2=rl2)
        print(f"{k:18s} all: >1e-4 {rec[k]['all_above']:6d} max {rec[k]['all_max']:.2e} | clean: >1e-4 {rec[k]['clean_above']:6d} max {rec[k]['clean_max']:.2e} relL2 {rl2:.2e}")
        if rec[k]["clean_above"] and k in ("v_ray_transforms", "v_densify", "v_colors"):
            idx = np.where(sm & (e > 1e-4))[0]
            idx = idx[np.argsort(-e[idx])][:6]
            for m in idx:
                print(f"      splat {m}: err {e[m]:.2e} radius {p['radii'][m]} opac {opa[m]:.3f} depth {p['depths'][m]:.2f} got {n(got).reshape(M,-1)[m][:3]} ref {np.asarray(ref).reshape(M,-1)[m][:3]}")
    # conditioning of the geometry sums: error in units of eps32 x sum|contribution|
    for k, ak, got in (("v_ray_transforms", "abs_ray_transforms", a[1].grad), ("v_densify", "abs_densify", dens.grad)):
        d = np.abs(np.asarray(n(got), np.float64) - g[k]).reshape(M, -1)
        ab = g[ak].reshape(M, -1)
        r = d / (6e-8 * ab + 1e-300)
        e = scaled(n(got), g[k]).reshape(M, -1)
        sel = sm[:, None] & (e > 1e-4)
        rr = r[sm]
        print(f"{k}: error / (eps32 * sum|terms|) on clean splats: median {np.median(rr):.2f} p99 {np.percentile(rr, 99):.1f} p99.99 {np.percentile(rr, 99.99):.1f} max {rr.max():.1f}; "
              f"among elements above 1e-4: min {r[sel].min() if sel.any() else 0:.1f} median {np.median(r[sel]) if sel.any() else 0:.1f} max {r[sel].max() if sel.any() else 0:.1f}; "
              f"cancellation (sum|terms| / max(|ref|, floor)) of those: median {np.median((ab / np.maximum(np.abs(g[k].reshape(M, -1)), np.abs(g[k]).mean()))[sel]) if sel.any() else 0:.0f}")
        rec[k]["err_over_eps_abs_sum_p9999"] = float(np.percentile(rr, 99.99)); rec[k]["err_over_eps_abs_sum_max"] = float(rr.max())
    if os.environ.get("DIAG_DUMP_LITE") == name:
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"diag_dump_{name}.npz"), v_ray_transforms=n(a[1].grad), v_densify=n(dens.grad),
                            in_means2d=p["means2d"], in_ray_transforms=p["ray_transforms"], in_colors=col, in_opacities=opa, in_normals=p["normals"],
                            in_offs=offs, in_flat=flat, in_radii=p["radii"], W=W, H=H)
    if os.environ.get("DIAG_DUMP") == name:
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"diag_dump_{name}.npz"), v_ray_transforms=n(a[1].grad), v_densify=n(dens.grad),
                            v_colors=n(a[2].grad), v_means2d=n(a[0].grad), v_opacities=n(a[3].grad), v_normals=n(a[4].grad),
                            render_normals=n(rn), render_colors=n(rc), render_alphas=n(ra),
                            in_means2d=p["means2d"], in_ray_transforms=p["ray_transforms"], in_colors=col, in_opacities=opa, in_normals=p["normals"],
                            in_offs=offs, in_flat=flat, in_radii=p["radii"])
    out[name] = rec
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_parity_fragile.json"), "w"), indent=1)
