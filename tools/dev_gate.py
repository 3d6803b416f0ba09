"""Offline development loop of the decision-matched gate (tests/util.py): the oracle's fp32 build stands in for the implementation
under test.  python tools/dev_gate.py [N W H seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gs_sdf_amd.synth as synth
from oracle import oracle as orc
import util

PREC = os.environ.get("PREC", "f32fma")
N, W, H, seed = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (10000, 256, 256, 0)
n = lambda t: t.detach().cpu().numpy()
sc = synth.make_scene(N, W, H, sh_degree=0, seed=seed, sigma_px=(0.5, 6.0))
vm = synth.make_views(2, seed=seed + 10)[1:]
p = orc.projection_2dgs_fwd(n(sc["means"]), n(sc["quats"]), n(sc["log_scales"].exp()), n(vm), n(sc["K"])[None], W, H, prec="f32")
col = orc.view_colors_fwd(n(vm), n(sc["means"]), n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], 0, prec="f32")
opa = n(torch.sigmoid(sc["logit_opacities"]))[p["gaussian_ids"]]
tpg, ids, flat, offs = orc.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1)
ug = synth.upstream_grads(H, W, seed=2)
t0 = time.time()
f32 = orc.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, prec=PREC)
g32 = orc.rasterize_2dgs_bwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, f32["render_alphas"],
                             f32["last_ids"], f32["median_ids"], *(n(ug[k]) for k in ("v_render_colors", "v_render_depths", "v_render_alphas",
                                                                                      "v_render_normals", "v_render_median")), prec=PREC)
trace = lambda rows, stride: orc.rasterize_2dgs_trace(p["means2d"], p["ray_transforms"], opa, W, H, 16, offs, flat, rows, stride, prec=PREC)
ref = util.matched_reference(orc, p, col, opa, W, H, offs, flat, ug, trace)
print("info", {k: v for k, v in ref["info"].items()}, f"{time.time()-t0:.1f}s")
got = {**f32, **g32}
print("last_ids equal", np.array_equal(got["last_ids"], ref["last_ids"]), "median_ids equal", np.array_equal(got["median_ids"], ref["median_ids"]))
for key in list(orc.PIX_BOUND_COLS) + ["visibilities"] + list(orc.COND_SLICES):
    st = util.matched_stats(got[key], ref[key], util.bound_of(ref, key, orc))
    print(f"{key:18s}", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
