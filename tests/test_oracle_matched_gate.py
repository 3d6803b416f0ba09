"""CPU self-check of the decision-matched parity gate (tests/util.py, oracle/splat_oracle.c "DECISION-MATCHED"): the oracle's fp32 build
compiled with FMA contraction stands in for the implementation under test (it decides and rounds like a GPU kernel would), the fp64
build evaluated under ITS decisions is the reference.  The gate must (a) accept it on every element with no pixel or splat excluded,
(b) reproduce its last_ids / median_ids exactly, (c) refuse a decision flipped outside the rounding noise, (d) refuse a perturbed value."""
import numpy as np
import pytest
import torch

import gs_sdf_amd.synth as synth
import util

n = lambda t: t.detach().cpu().numpy()
UP = ("v_render_colors", "v_render_depths", "v_render_alphas", "v_render_normals", "v_render_median")


def _problem(oracle, N, W, H, seed, sigma=(0.5, 6.0)):
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=seed, sigma_px=sigma)
    vm = synth.make_views(2, seed=seed + 10)[1:]
    p = oracle.projection_2dgs_fwd(n(sc["means"]), n(sc["quats"]), n(sc["log_scales"].exp()), n(vm), n(sc["K"])[None], W, H, prec="f32")
    col = oracle.view_colors_fwd(n(vm), n(sc["means"]), n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], 0, prec="f32")
    opa = n(torch.sigmoid(sc["logit_opacities"]))[p["gaussian_ids"]]
    tpg, ids, flat, offs = oracle.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1)
    return p, col, opa, flat, offs, synth.upstream_grads(H, W, seed=2)


def _standin(oracle, p, col, opa, W, H, offs, flat, ug, bg=None):
    fw = oracle.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, backgrounds=bg, prec="f32fma")
    g = oracle.rasterize_2dgs_bwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, fw["render_alphas"],
                                  fw["last_ids"], fw["median_ids"], *(n(ug[k]) for k in UP), backgrounds=bg, prec="f32fma")
    trace = lambda rows, stride: oracle.rasterize_2dgs_trace(p["means2d"], p["ray_transforms"], opa, W, H, 16, offs, flat, rows, stride, prec="f32fma")
    return {**fw, **g}, trace


@pytest.mark.parametrize("N,W,H,seed,bg", [(6000, 192, 128, 0, False), (2500, 100, 72, 3, True)])
def test_gate_accepts_a_correct_fp32_evaluation_on_every_element(oracle, N, W, H, seed, bg):
    p, col, opa, flat, offs, ug = _problem(oracle, N, W, H, seed)
    bgv = np.array([[0.1, 0.4, 0.8]], np.float32) if bg else None
    got, trace = _standin(oracle, p, col, opa, W, H, offs, flat, ug, bgv)
    ref = util.matched_reference(oracle, p, col, opa, W, H, offs, flat, ug, trace, backgrounds=bgv, recovers_final_T=True)
    assert ref["info"]["traced_pixels"] > 0, "the scene has no decision-fragile pixel: the test would not exercise the trace"
    # (the stand-in is the plain direct form h_u x h_v with 1 - render_alphas as final transmittance: its aggregate error is larger than
    # the HIP kernels', whose tests keep the 1e-5 relative-L2 bar)
    stats = util.assert_all_matched(got, ref, oracle, case=f"standin N={N}", keys=util.RASTER_TENSORS, max_rel_l2=1e-4, max_needed=1e-2)
    assert set(stats) == set(util.RASTER_TENSORS)
    # nothing is excluded: every tensor reports all of its elements
    assert stats["v_ray_transforms"]["elements"] == 9 * opa.shape[0] and stats["render_colors"]["elements"] == 3 * W * H


def test_unmatched_reference_differs_where_decisions_flip(oracle):
    """Why the trace exists: in the flagged pixels the fp64 evaluation's OWN decisions give a different image than the implementation's
    (a flip moves a pixel by O(alpha T)); under the traced decisions the same pixels agree to 1e-4."""
    N, W, H = 6000, 192, 128
    p, col, opa, flat, offs, ug = _problem(oracle, N, W, H, 0)
    got, trace = _standin(oracle, p, col, opa, W, H, offs, flat, ug)
    ref = util.matched_reference(oracle, p, col, opa, W, H, offs, flat, ug, trace, recovers_final_T=True)
    own = oracle.rasterize_2dgs_fwd_matched(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, prec="f64")
    flips = sum(c for c, _ in ref["info"]["flips"].values())
    same_ids = np.array_equal(own["last_ids"], ref["last_ids"]) and np.array_equal(own["median_ids"], ref["median_ids"])
    assert (flips == 0) == (same_ids and np.array_equal(own["render_colors"], ref["render_colors"]))
    util.assert_flips_inside_noise(ref["info"]["flips"])


def test_gate_refuses_a_wrong_decision_and_a_wrong_value(oracle):
    N, W, H = 4000, 128, 96
    p, col, opa, flat, offs, ug = _problem(oracle, N, W, H, 1)
    got, trace = _standin(oracle, p, col, opa, W, H, offs, flat, ug)

    def bad_trace(rows, stride):          # an implementation that drops a clearly visible splat in a traced pixel
        bits = trace(rows, stride).copy()
        r = int(np.argmax((bits & 1).sum(1)))
        k = int(np.flatnonzero(bits[r] & 1)[0])
        bits[r, k] = 0
        return bits
    ref = util.matched_reference(oracle, p, col, opa, W, H, offs, flat, ug, bad_trace)
    with pytest.raises(AssertionError, match="decision differs"):
        util.assert_flips_inside_noise(ref["info"]["flips"])
    ref = util.matched_reference(oracle, p, col, opa, W, H, offs, flat, ug, trace, recovers_final_T=True)
    g = dict(got)
    g["v_opacities"] = got["v_opacities"].copy()
    i = int(np.argmax(np.abs(g["v_opacities"])))
    g["v_opacities"][i] *= 1.0 + 1e-3     # 10 x the bar on one element of one splat
    with pytest.raises(AssertionError, match="v_opacities"):
        util.assert_all_matched(g, ref, oracle, keys=("v_opacities",))
    g = dict(got)
    g["last_ids"] = got["last_ids"].copy(); g["last_ids"][0, 5, 5] += 1
    with pytest.raises(AssertionError, match="last_ids"):
        util.assert_all_matched(g, ref, oracle, keys=())


def test_independent_fp32_builds_land_as_far_from_fp64_as_the_stand_in(oracle):
    """CPU self-check of the cross-check (tests/util.py "INDEPENDENT leg"): the stand-in implementation (f32fma build, double accumulation) and the
    two fully-fp32 builds are three correct fp32 evaluations; wherever the stand-in leaves the plain 1e-4 bar the others must be comparably
    far out, and a value pushed 50 x further out than any of them must be refused."""
    N, W, H = 6000, 192, 128
    p, col, opa, flat, offs, ug = _problem(oracle, N, W, H, 0)
    got, trace = _standin(oracle, p, col, opa, W, H, offs, flat, ug)
    ref = util.matched_reference(oracle, p, col, opa, W, H, offs, flat, ug, trace, recovers_final_T=True)
    alts = util.independent_fp32_evaluations(oracle, ref, p, col, opa, W, H, offs, flat, ug)
    for prec in util.INDEP_BUILDS:      # the builds see the same decisions: identical ids
        assert np.array_equal(alts[prec]["last_ids"], ref["last_ids"]) and np.array_equal(alts[prec]["median_ids"], ref["median_ids"])
    seen_needed = 0
    for key in util.RASTER_TENSORS:
        st = util.independent_stats(got[key], ref[key], [alts[b][key] for b in util.INDEP_BUILDS])
        util.check_independent(st, key)
        seen_needed += st["needed"]
        assert st["rel_l2_alt"] < 1e-3
    # float accumulation is really on in the builds: their gradients differ from the double-accumulating f32 build's
    g32 = oracle.rasterize_2dgs_bwd_matched(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, alts["f32acc"]["render_alphas"],
                                            ref["last_ids"], ref["median_ids"], *(n(ug[k]) for k in UP), trace_rows=ref["trace_rows"], trace_bits=ref["trace_bits"],
                                            prec="f32", recovers_final_T=True)
    assert not np.array_equal(g32["v_colors"], alts["f32acc"]["v_colors"])
    bad = np.array(got["v_ray_transforms"], np.float64).copy()
    i = int(np.argmax(np.abs(bad - ref["v_ray_transforms"]).reshape(-1)))
    base = 1e-4 * max(abs(ref["v_ray_transforms"].reshape(-1)[i]), np.abs(ref["v_ray_transforms"]).mean())
    worst_alt = max(util.independent_stats(got["v_ray_transforms"], ref["v_ray_transforms"], [alts[b]["v_ray_transforms"] for b in util.INDEP_BUILDS])["worst_over_base_alt"], 1.0)
    bad.reshape(-1)[i] = ref["v_ray_transforms"].reshape(-1)[i] + 50.0 * worst_alt * base
    st = util.independent_stats(bad, ref["v_ray_transforms"], [alts[b]["v_ray_transforms"] for b in util.INDEP_BUILDS])
    with pytest.raises(AssertionError, match="v_ray_transforms"):
        util.check_independent(st, "v_ray_transforms")
