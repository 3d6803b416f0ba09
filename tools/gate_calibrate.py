"""Calibration run of the decision-matched gate on the GPU box: per case and tensor the statistics of tests/util.py: matched_stats
(c_needed = the smallest safety factor on the conditioning bound that would pass).  No asserts.  python tools/gate_calibrate.py [small|all]"""
import json, math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import gs_sdf_amd.ops as ops, gs_sdf_amd.synth as synth
from oracle import oracle as orc
import util

dev = torch.device("cuda:0")
n = lambda t: t.detach().cpu().numpy()
orc.set_threads(os.cpu_count() or 1)
which = sys.argv[1] if len(sys.argv) > 1 else "small"
CASES = [("10k_256", 10_000, 256, 256, 0, 1, 0, False), ("3k_200x120_sh3", 3_000, 200, 120, 3, 1, 1, False), ("5k_160x96_3cam", 5_000, 160, 96, 1, 3, 2, False),
         ("40k_640x368", 40_000, 640, 368, 0, 1, 3, False)]
if which == "all":
    CASES += [("cfg1_300k_1200x680", 300_000, 1200, 680, 0, 1, 0, True), ("cfg3_1M_1080p", 1_000_000, 1920, 1080, 0, 1, 0, False)]
out = {}


def run(name, sc, vm, W, H, deg, V, bg):
    t0 = time.time()
    Kd = sc["K"].reshape(-1, 3, 3)[:1].expand(V, 3, 3).contiguous()
    p = orc.projection_2dgs_fwd(n(sc["means"]), n(sc["quats"]), n(sc["log_scales"].exp()), n(vm), n(Kd), W, H, prec="f32")
    col = orc.view_colors_fwd(n(vm), n(sc["means"]), n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, prec="f32")
    opa = n(torch.sigmoid(sc["logit_opacities"]))[p["gaussian_ids"]]
    tpg, ids, flat, offs = orc.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], V)
    ug = synth.upstream_grads(H, W, seed=2, C=V)
    got, trace_fn = util.hip_compositing(ops, p, col, opa, W, H, offs, flat, ug, dev, backgrounds=bg)
    ref = util.matched_reference(orc, p, col, opa, W, H, offs, flat, ug, trace_fn, backgrounds=bg)
    if "last_ids" not in got:
        trace_fn(np.full(ref["last_ids"].shape, -1, np.int32), 1)
    rec = dict(info=ref["info"], last_ids_identical=bool(np.array_equal(n(got["last_ids"]), ref["last_ids"])),
               median_ids_identical=bool(np.array_equal(n(got["median_ids"]), ref["median_ids"])), seconds=round(time.time() - t0, 1))
    print(f"== {name}: M={opa.shape[0]} I={flat.shape[0]} traced={ref['info']['traced_pixels']} stride={ref['info']['trace_stride']} flips={ref['info']['flips']} "
          f"last_ids={rec['last_ids_identical']} median_ids={rec['median_ids_identical']} ({rec['seconds']} s)", flush=True)
    for key in util.RASTER_TENSORS:
        st = util.matched_stats(n(got[key]), ref[key], util.bound_of(ref, key, orc))
        rec[key] = st
        print(f"   {key:18s} n={st['elements']:9d} needed={st['needed']:6d} relaxed={st['relaxed']:8d} c_needed={st['c_needed']:8.3f} over_tol={st['worst_over_tol']:7.3f} "
              f"over_base={st['worst_over_base']:8.2f} rel_l2={st['rel_l2']:.2e}", flush=True)
    out[name] = rec
    if which != "all" or opa.shape[0] < 50_000:    # keep the small cases' HIP outputs for offline work on the bound
        os.makedirs(os.path.join(ROOT, "gpurun_out", "gate"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "gate", name + ".npz"), trace_rows=ref["trace_rows"], trace_bits=ref["trace_bits"],
                            **{k: n(v) for k, v in got.items()})


for name, N, W, H, deg, V, seed, replica in CASES:
    big = N >= 300_000
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0 if big else seed, replica=replica) if big else synth.make_scene(N, W, H, sh_degree=deg, seed=seed, sigma_px=(0.5, 6.0))
    vm = synth.make_views(2, seed=1)[1:2] if big else synth.make_views(V + 1, seed=seed + 10)[1:]
    bg = np.array([[0.1, 0.4, 0.8]] * V, np.float32) if (seed % 2 and not big) else None
    run(name, sc, vm, W, H, deg, V, bg)

# the adversarial scene of tests/test_gpu_splat_parity.py::test_pathological_splats_keep_parity
W, H, N = 160, 112, 1500
g = torch.Generator().manual_seed(12)
sc = synth.make_scene(N, W, H, sh_degree=0, seed=12, sigma_px=(0.3, 2.0))
sc["log_scales"][:300] += math.log(60.0)
sc["means"][300:600, 2] = 0.06 + 0.3 * torch.rand(300, generator=g)
sc["means"][300:600, :2] *= 0.02
sc["quats"][600:900] = torch.tensor([0.7071, 0.7071, 0.0, 0.0]) + 0.01 * torch.randn(300, 4, generator=g)
sc["logit_opacities"][900:1200] = torch.logit(torch.full((300,), 1.0 / 255.0) + 0.002 * torch.rand(300, generator=g))
run("pathological", sc, synth.make_views(2, seed=3)[1:], W, H, 0, 1, None)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"gate_calibration_{which}.json"), "w"), indent=1)
