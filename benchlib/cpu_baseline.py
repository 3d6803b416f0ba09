"""cpu_baseline leg of bench.py: the oracle timed on the host cores + the oracle-vs-HIP parity leg.  The ONLY bench module that imports oracle/."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _usable_cores():
    """the cores this process may really use: CPU affinity, capped by the container's CFS quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def _err_stats(got, ref, clean=None):
    """Scaled error (|got-ref| / max(|ref|, mean|ref|)) per row: rows above 1e-4, worst, relative L2 — over all rows and, when
    `clean` (bool over the leading dims) is given, over those rows too ("masked": the decoder's points away from a ReLU kink)."""
    import numpy as np
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return {"n": 0}
    R = ref.shape[0] if clean is None else int(np.prod(clean.shape))
    g2, r2 = got.reshape(R, -1), ref.reshape(R, -1)
    floor = np.abs(r2).mean() + 1e-30
    e = (np.abs(g2 - r2) / np.maximum(np.abs(r2), floor)).max(1)
    out = {"rows": int(R), "worst": float(e.max()), "rel_l2": float(np.linalg.norm(g2 - r2) / (np.linalg.norm(r2) + 1e-30)),
           "rows_above_1e-4": int((e > 1e-4).sum())}
    if clean is not None:
        c = np.asarray(clean).reshape(R)
        out["masked"] = {"rows": int(c.sum()), "rows_above_1e-4": int((e[c] > 1e-4).sum()), "worst": float(e[c].max()) if c.any() else 0.0,
                                  "rel_l2": float(np.linalg.norm((g2 - r2)[c]) / (np.linalg.norm(r2[c]) + 1e-30))}
    return out


def cpu_baseline(sc, views, params, N, W, H, deg, n_sdf_points, dev):
    """The oracle ("port": the reference has no CPU rasteriser and none of its kernels are vendored) timed on this box's
    host cores on a BOUNDED sample of the step: the splat half of ONE iteration in full (projection, SH, binning,
    compositing forward + backward, projection / SH backward at the workload's own size) + the SDF half (hash grid +
    decoder forward and backward) on at most 300 000 of the step's query points, scaled linearly to all of them.
    The oracle's outputs are then compared with the HIP path's on the same inputs (the parity leg of the bench line)."""
    import numpy as np
    import gs_sdf_amd.ops as ops
    import gs_sdf_amd.synth as synth
    from oracle import oracle as orc
    orc.build()
    cores = _usable_cores()
    orc.set_threads(cores)
    n = lambda t: t.detach().cpu().numpy()
    view = views[0:1].cpu()
    means, quats = n(sc["means"]), n(sc["quats"])
    t0 = time.perf_counter()
    scales, opac = np.exp(n(sc["log_scales"])), 1.0 / (1.0 + np.exp(-n(sc["logit_opacities"])))
    p = orc.projection_2dgs_fwd(means, quats, scales, n(view), n(sc["K"]), W, H)
    col = orc.view_colors_fwd(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg)
    tpg, ids, flat, offs = orc.tile_encode(W, H, 16, p["means2d"], p["radii"], p["depths"], p["camera_ids"], 1)
    opa = opac[p["gaussian_ids"]]
    fw = orc.rasterize_2dgs_fwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat)
    ug = synth.upstream_grads(H, W, seed=2)
    g = orc.rasterize_2dgs_bwd(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat,
                               fw["render_alphas"], fw["last_ids"], fw["median_ids"], n(ug["v_render_colors"]),
                               n(ug["v_render_depths"]), n(ug["v_render_alphas"]), n(ug["v_render_normals"]),
                               n(ug["v_render_median"]), absgrad=False)
    M = p["gaussian_ids"].shape[0]
    pb = orc.projection_2dgs_bwd(means, quats, scales, n(view), n(sc["K"]), W, H, p["camera_ids"], p["gaussian_ids"],
                                 g["v_means2d"].astype(np.float32), np.zeros(M, np.float32),
                                 g["v_ray_transforms"].astype(np.float32), g["v_normals"].astype(np.float32))
    orc.view_colors_bwd(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, g["v_colors"].astype(np.float32))
    t_splat = time.perf_counter() - t0
    t_sdf, n_s = 0.0, 0
    if n_sdf_points:
        # SDF leg: hash-grid + decoder forward and backward on a bounded sample of the step's query points
        n_s = min(n_sdf_points, 300_000)
        rng = np.random.default_rng(4)
        _, total = orc.grid_offsets()
        table = ((rng.random((total, 2), dtype=np.float32) * 2 - 1) * 1e-4).astype(np.float32)
        dims = [32, 64, 64, 64, 2]
        Wm = (rng.standard_normal(sum(i * o for i, o in zip(dims[:-1], dims[1:]))) * 0.1).astype(np.float32)
        xs = rng.random((n_s, 3), dtype=np.float32)
        t1 = time.perf_counter()
        feat = orc.grid_fwd(xs, table)
        o = orc.mlp_fwd(feat, dims, Wm, None)
        v_in, v_w, _ = orc.mlp_bwd(feat, dims, Wm, None, np.ones_like(o))
        vt_o, _ = orc.grid_bwd(xs, table, v_in)
        t_sdf = (time.perf_counter() - t1) * (n_sdf_points / n_s)
    dt = t_splat + t_sdf
    # ---- parity leg (not timed): the HIP operators on the same inputs against the oracle ---------------------------------
    # integers against the fp32 build just timed (bit-exact contract); floats against the fp64 build of the compositing
    # forward / backward and of the projection backward (truth: the fp32 CPU build itself is 1e-2 off on these gradients,
    # profiles/parity_r02.json)
    f64 = lambda a: np.asarray(a, np.float64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # DECISION-MATCHED reference (tests/util.py, oracle/splat_oracle.c): the kernel's decisions in every decision-fragile pixel are
    # traced (instrumented instantiation of the same kernel on the same inputs) and the fp64 oracle is evaluated under them: no
    # pixel and no splat is excluded from the comparison below
    pf, sf, _ = orc.rasterize_2dgs_fragility(p["means2d"], p["ray_transforms"], opa, W, H, 16, offs, flat)
    rows, stride, n_rows = orc.trace_plan(pf, offs, flat.shape[0])
    tr = ops.rasterize_fwd_instr(t(p["means2d"]), t(p["ray_transforms"]), t(col), t(opa), t(p["normals"]), W, H, t(offs), t(flat),
                                 trace_rows=t(rows), trace_stride=stride)
    bits = n(tr["trace_bits"])
    fw64 = orc.rasterize_2dgs_fwd_matched(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat, trace_rows=rows,
                                          trace_bits=bits, prec="f64")
    g64 = orc.rasterize_2dgs_bwd_matched(p["means2d"], p["ray_transforms"], col, opa, p["normals"], W, H, 16, offs, flat,
                                         fw64["render_alphas"], fw64["last_ids"], fw64["median_ids"], n(ug["v_render_colors"]),
                                         n(ug["v_render_depths"]), n(ug["v_render_alphas"]), n(ug["v_render_normals"]),
                                         n(ug["v_render_median"]), trace_rows=rows, trace_bits=bits, prec="f64")
    pb64 = orc.projection_2dgs_bwd(means, quats, scales, n(view), n(sc["K"]), W, H, p["camera_ids"], p["gaussian_ids"],
                                   f64(g64["v_means2d"]), np.zeros(M, np.float64), f64(g64["v_ray_transforms"]), f64(g64["v_normals"]),
                                   prec="f64")
    vsh64 = orc.view_colors_bwd(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, f64(g64["v_colors"]), prec="f64")   # (v_sh, v_means)
    vop64 = np.zeros(N)
    np.add.at(vop64, p["gaussian_ids"], f64(g64["v_opacities"]))
    # End-to-end parameter gradients: bound(leaf) = the compositing gradients' first-order error bounds (g64["cond"], eps32 units) pushed through
    # the ABSOLUTE SHADOW of the projection / SH backward (oracle: orc_projection_2dgs_bwd_bound — every term enters with its absolute value,
    # internal cancellations included) + the backward's own evaluation error (PROJ_COND_C eps32 x the shadow of |upstream|, the rule of
    # tests/test_gpu_baseline_shapes.py); gated like the compositing's own: 1e-4 max(|ref|, mean|ref|) + COND_C eps32 bound
    cnd = g64["cond"]
    PROJ_OVER_COND = 16.0 / 2.0          # PROJ_COND_C / COND_C
    zM = np.zeros(M, np.float64)
    pa = (means, quats, scales, n(view), n(sc["K"]), W, H, p["camera_ids"], p["gaussian_ids"])
    b1 = orc.projection_2dgs_bwd_bound(*pa, cnd[:, 0:2], zM, cnd[:, 2:11].reshape(M, 3, 3), cnd[:, 15:18])
    b2 = orc.projection_2dgs_bwd_bound(*pa, np.abs(f64(g64["v_means2d"])), zM, np.abs(f64(g64["v_ray_transforms"])), np.abs(f64(g64["v_normals"])))
    s1 = orc.view_colors_bwd_bound(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, cnd[:, 11:14])
    s2 = orc.view_colors_bwd_bound(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, np.abs(f64(g64["v_colors"])))
    b_means = b1[0] + s1[1] + PROJ_OVER_COND * (b2[0] + s2[1])
    b_quats, b_scales = b1[1] + PROJ_OVER_COND * b2[1], b1[2] + PROJ_OVER_COND * b2[2]
    b_sh = s1[0] + PROJ_OVER_COND * s2[0]
    b_opac = np.zeros(N)
    np.add.at(b_opac, p["gaussian_ids"], cnd[:, 14])
    leaves = [t(a).requires_grad_(True) for a in (means, quats, scales, opac, n(sc["sh"]))]
    colors, alphas, meta = ops.rasterization_2dgs_sdf(*leaves, view.to(dev), sc["K"].to(dev), W, H, "RGB+D", 0.05, 300.0, 0.0, deg)
    ugd = {k: v.to(dev) for k, v in ug.items()}
    # RGB+D keeps the accumulated depth (the oracle's render_depths); normals go back to the camera frame for the comparison
    R = view[0, :3, :3].to(dev)
    rn_cam = meta["render_normal"] @ R.t()
    loss = ((colors[..., :3] * ugd["v_render_colors"]).sum() + (colors[..., 3:4] * ugd["v_render_depths"]).sum()
            + (alphas * ugd["v_render_alphas"]).sum() + (rn_cam * ugd["v_render_normals"]).sum()
            + (meta["render_median"] * ugd["v_render_median"]).sum())
    loss.backward()
    torch.cuda.synchronize()
    EPS32, COND_C = 2.0 ** -24, 2.0

    def matched(got, ref, bound):
        """|got - ref| <= 1e-4 max(|ref|, mean|ref|) + COND_C eps32 bound for every element (tests/util.py: matched_stats)"""
        got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
        b = np.asarray(bound, np.float64)
        b = b.reshape(ref.shape) if b.size == ref.size else np.broadcast_to(b.reshape(b.shape + (1,) * (ref.ndim - b.ndim)), ref.shape)
        base = 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean() + 1e-30)
        err = np.abs(got - ref)
        return {"elements": int(err.size), "above_1e-4": int((err > base).sum()), "worst_over_1e-4_bar": float((err / base).max()),
                "worst_over_tolerance": float((err / (base + COND_C * EPS32 * b)).max()),
                "rel_l2": float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))}
    pb = fw64["pix_bound"]
    par = {"integer_outputs_bit_exact": bool(np.array_equal(n(meta["gaussian_ids"]), p["gaussian_ids"]) and np.array_equal(n(meta["radii"]), p["radii"])
                                             and np.array_equal(n(meta["flatten_ids"]), flat) and np.array_equal(n(meta["isect_offsets"]), offs)
                                             and np.array_equal(n(meta["tiles_per_gauss"]), tpg)),
           "decision_matching": {"traced_pixels": n_rows, "traced_fraction": n_rows / max(pf.size, 1), "excluded_pixels": 0, "excluded_splats": 0,
                                 "flips (count, worst margin in fp32-evaluation errors)": fw64["flips"],
                                 "last_ids_identical": bool(np.array_equal(n(tr["last_ids"]), fw64["last_ids"])),
                                 "median_ids_identical": bool(np.array_equal(n(tr["median_ids"]), fw64["median_ids"])),
                                 "instrumented_forward_bit_identical_to_the_product_kernel": bool(torch.equal(tr["render_alphas"], alphas.detach()))},
           "render_colors": matched(n(colors[..., :3]), fw64["render_colors"], pb[..., 0]), "render_depths": matched(n(colors[..., 3:4]), fw64["render_depths"], pb[..., 1]),
           "render_alphas": matched(n(alphas), fw64["render_alphas"], pb[..., 2]), "render_normals": matched(n(rn_cam), fw64["render_normals"], pb[..., 3]),
           "render_median": matched(n(meta["render_median"]), fw64["render_median"], pb[..., 4]),
           "visibilities": matched(n(meta["visibilities"]), fw64["visibilities"], fw64["vis_bound"]),
           "v_densify": matched(n(meta["gradient_2dgs"].grad), g64["v_densify"], g64["cond"][:, orc.COND_SLICES["v_densify"]]),
           "v_means (compositing + projection + SH backward)": matched(n(leaves[0].grad), pb64[0] + vsh64[1], b_means),
           "v_quats (compositing + projection backward)": matched(n(leaves[1].grad), pb64[1], b_quats), "v_scales": matched(n(leaves[2].grad), pb64[2], b_scales),
           "v_opacities": matched(n(leaves[3].grad), vop64, b_opac), "v_sh": matched(n(leaves[4].grad), vsh64[0], b_sh),
           "note": "HIP path vs the oracle on the bench workload's first view, NO pixel or splat excluded: ids / radii / bins / offsets bit-exact against "
                   "the fp32 build; floats against the fp64 build evaluated under the kernel's own traced decisions (oracle.rasterize_2dgs_*_matched). "
                   "Compositing outputs: every element against 1e-4 max(|ref|, mean|ref|) + 2 eps32 x the oracle's first-order conditioning bound "
                   "(worst_over_tolerance <= 1 is the gate of tests/util.py; above_1e-4 = elements that needed the second term). End-to-end "
                   "parameter gradients (compositing -> projection / SH backward): the same element-wise comparison, the compositing bounds "
                   "pushed through the fp64 projection / SH backward by absolute values "
                   "(tests/test_gpu_baseline_shapes.py runs the comparison at every BASELINE shape)"}
    if n_sdf_points:
        # SDF half: the HIP encoder / decoder / scatter on the sample the oracle was timed on.  Features and table gradient
        # against the fp32 build (pos = fma(scale, x, 0.5) in fp32 IS the function), decoder against the fp64 build
        import ctypes as C
        import gs_sdf_amd.capi as capi
        L = capi.lib()
        gcfg = (16, 2, 19, 32, 2.0)
        nl, dims_c = len(dims) - 1, (C.c_int * len(dims))(*dims)
        xd, td, Wd, fd = t(xs), t(table), t(Wm), t(feat)
        feat_h = torch.empty(n_s, 32, device=dev)
        capi.check(L.gsdf_hashgrid_fwd(n_s, *gcfg, capi.f32(xd), capi.f32(td), capi.f32(feat_h), capi.stream()), "hashgrid_fwd")
        out_h = torch.empty(n_s, dims[-1], device=dev)
        acts = torch.empty(L.gsdf_mlp_acts_floats(n_s, nl), device=dev)
        capi.check(L.gsdf_mlp_fwd(n_s, nl, dims_c, capi.f32(Wd), None, capi.f32(fd), capi.f32(out_h), capi.f32(acts), capi.stream()), "mlp_fwd")
        vin_h, vw_h = torch.empty_like(fd), torch.zeros_like(Wd)
        ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes_for(n_s, nl, dims_c, 1), dtype=torch.uint8, device=dev)
        capi.check(L.gsdf_mlp_bwd(n_s, nl, dims_c, capi.f32(Wd), None, capi.f32(fd), capi.f32(acts), capi.f32(torch.ones_like(out_h)), capi.f32(vin_h),
                                  capi.f32(vw_h), None, capi.ptr(ws) if ws.numel() else None, capi.stream()), "mlp_bwd")
        nb = L.gsdf_hashgrid_bwd_binned_ws_bytes(n_s, *gcfg)
        bws = torch.empty(nb, dtype=torch.uint8, device=dev)
        vt_h = torch.zeros(table.shape[0], 2, device=dev)
        capi.check(L.gsdf_hashgrid_bwd_binned(n_s, *gcfg, capi.f32(xd), capi.f32(t(v_in.astype(np.float32))), capi.f32(vt_h), capi.ptr(bws), nb, capi.stream()), "scatter")
        torch.cuda.synchronize()
        o64 = orc.mlp_fwd(feat, dims, Wm, None, prec="f64")
        vin64, vw64, _ = orc.mlp_bwd(feat, dims, Wm, None, np.ones_like(o64), prec="f64")
        # points with a hidden pre-activation within 1e-5 (of the layer's rms) of zero may take the other ReLU branch than the fp64 evaluation (a decision, like
        # the compositing's): the same mask as tests/test_gpu_sdf_parity.py::_near_relu_kink; both figures are printed
        hcur, off_, away = feat.astype(np.float64), 0, np.ones(n_s, bool)
        for l_ in range(len(dims) - 2):
            z_ = hcur @ Wm[off_:off_ + dims[l_] * dims[l_ + 1]].astype(np.float64).reshape(dims[l_ + 1], dims[l_]).T
            off_ += dims[l_] * dims[l_ + 1]
            away &= ~(np.abs(z_) < 1e-5 * np.sqrt((z_ * z_).mean())).any(axis=1)      # relative to the layer's pre-activation scale (here ~1e-4: the table is U(-1e-4, 1e-4))
            hcur = np.maximum(z_, 0.0)
        par["sdf"] = {"hashgrid_features (vs f32 build)": _err_stats(n(feat_h), feat), "decoder_out (vs f64 build)": _err_stats(n(out_h), o64),
                      "decoder_v_in (vs f64 build)": _err_stats(n(vin_h), vin64, away), "decoder_v_weights (vs f64 build)": _err_stats(n(vw_h), vw64),
                      "points_within_1e-5_rms_of_a_relu_kink": int((~away).sum()),
                      "table_gradient (vs f32 build)": _err_stats(n(vt_h), vt_o),
                      "note": f"{n_s} uniformly random points, table U(-1e-4, 1e-4), 4-layer bias-free decoder; the decoder runs on the bf16 MFMA pipe with "
                              "exact 3-term operand splits (GSDF_MLP_MFMA=f32 selects the fp32 pipe); a point whose pre-activation is within "
                              "rounding of zero may take the other ReLU branch than the fp64 evaluation: those are the elements above 1e-4"}
    return {"value": 1.0 / dt, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample_short": f"splat half of 1 iteration in full ({t_splat:.1f} s) + SDF half on {n_s} of {n_sdf_points} points scaled ({t_sdf:.1f} s); OpenMP over tiles, "
                            "projection threaded over the splats, sort single-threaded, hash grid + decoder threaded over the points",
            "sample": f"splat half of 1 iteration in full (oracle/splat_oracle.c f32 build, OpenMP over tiles on {cores} threads for "
                      f"compositing, projection threaded over the splats, sort single-threaded): {t_splat:.1f} s" +
                      (f"; SDF half (oracle/sdf_oracle.c fwd+bwd, OpenMP over the points) on {n_s} of the step's {n_sdf_points} query points, scaled linearly: "
                       f"{t_sdf:.1f} s" if n_sdf_points else ""),
            "parity": par}

