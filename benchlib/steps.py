"""Step constructors and the secondary step loops of bench.py (C++ JointIteration, the zero-edit reference loop)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_cpp_iteration(args, sc, params, dev, W, H, deg, views):
    """gsdf_extras::JointIteration on the bench's scene (same initial parameters as the Python step) + its per-step inputs."""
    import gs_sdf_amd.hostlib as hostlib
    import gs_sdf_amd.sdf as sdfm
    host = hostlib.load()
    analytic, ref_terms = args.sdf_config == "default", args.step_terms == "reference"
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0 if analytic else 1, device=dev, seed=5)
    enc = host.TCNNEncoding(16, 2, 19, 32, 2.0)
    dec = host.TCNNNetwork(32, 2, 64, 4 if analytic else 3, analytic)       # default: the torch decoder's topology (biases, 4 hidden matmuls)
    enc.params_, dec.params_ = lm.encoder.params_.detach().clone(), lm.decoder.params_.detach().clone()
    if analytic:
        dec.biases_ = lm.decoder.biases_.detach().clone()
    fields = [params.views[k].detach().clone() for k in ("offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest")]
    ji = host.JointIteration(params.anchors, fields, enc, dec, [0.0, 0.0, 5.5], 16.0, 0.02, 8, W, H, deg, not args.no_overlap, analytic, ref_terms,
                            args.sample_mode == "center", getattr(args, "hashgrid_resident", -1), getattr(args, "samples_grad_first", -1))   # level 8: 1/16 m leaves in 16 m
    gq = torch.Generator().manual_seed(4)
    pool = [((torch.rand(32768, 3, generator=gq) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev) for _ in range(8)]
    ray_sdf = [(torch.randn(32768, 1, generator=gq) * 0.02).to(dev) for _ in range(8)]
    Kh = [float(v) for v in (sc["K"][0, 0, 0], sc["K"][0, 1, 1], sc["K"][0, 0, 2], sc["K"][0, 1, 2])]
    cams = [Kh + [float(v) for v in torch.linalg.inv(vw.double())[:3, :4].reshape(-1)] for vw in views.cpu()]      # host values, known ahead
    return ji, pool, ray_sdf, cams, list(lm.decoder.dims)


def cpp_step(args, sc, views, K, ug6, target, N, W, H, deg, dev):
    """gsdf_extras::JointIteration (gs-sdf_amd/host/src/joint_step.cpp) on the bench's scene: the joint iteration in C++/libtorch,
    same configuration (--sdf-config, --step-terms), initial parameters, views and ray batches as the Python step."""
    import gs_sdf_amd.hostlib as hostlib
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.trainer import SplatParams, morton_order
    host = hostlib.load()
    analytic, ref_terms = args.sdf_config == "default", args.step_terms == "reference"
    params = SplatParams.from_scene(sc, dev, morton_order(sc["means"]) if args.splat_order == "morton" else None)
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0 if analytic else 1, device=dev, seed=5)
    enc = host.TCNNEncoding(16, 2, 19, 32, 2.0)
    dec = host.TCNNNetwork(32, 2, 64, 4 if analytic else 3, analytic)       # default: the torch decoder's topology (biases, 4 hidden matmuls)
    enc.params_, dec.params_ = lm.encoder.params_.detach().clone(), lm.decoder.params_.detach().clone()
    if analytic:
        dec.biases_ = lm.decoder.biases_.detach().clone()
    fields = [params.views[k].detach().clone() for k in ("offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest")]
    two = not args.no_overlap
    ji = host.JointIteration(params.anchors, fields, enc, dec, [0.0, 0.0, 5.5], 16.0, 0.02, 8, W, H, deg, two, analytic, ref_terms,
                            args.sample_mode == "center")   # level 8: 1/16 m leaves in 16 m
    gq = torch.Generator().manual_seed(4)
    pool = [((torch.rand(32768, 3, generator=gq) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev) for _ in range(8)]
    ray_sdf = [(torch.randn(32768, 1, generator=gq) * 0.02).to(dev) for _ in range(8)]
    up = [] if ref_terms else [ug6[k] for k in ("v_render_depths", "v_render_alphas", "v_render_normals", "v_render_median")]
    Kh = [float(v) for v in (sc["K"][0, 0, 0], sc["K"][0, 1, 1], sc["K"][0, 0, 2], sc["K"][0, 1, 2])]
    cams = [Kh + [float(v) for v in torch.linalg.inv(vw.double())[:3, :4].reshape(-1)] for vw in views.cpu()]      # host values, known ahead
    nv = views.shape[0]
    if args.dump_grads:
        sizes = ji.step(views[0][None], K, target, pool[0], ray_sdf[0], up, False, cams[0])
        torch.cuda.synchronize()
        torch.save({"splat": ji.splat_flat_grad().cpu(), "sdf": [ji.sdf_flat_grad().cpu()], "sizes": dict(sizes)}, args.dump_grads)
        return {"dumped": args.dump_grads}
    n_sdf = []
    for i in range(args.warmup):
        ji.step(views[i % nv][None], K, target, pool[i % 8], ray_sdf[i % 8], up, True, cams[i % nv])
    torch.cuda.synchronize()
    i = args.warmup
    while i < args.warmup + max(0, int(os.environ.get("GSDF_BENCH_WARM_STEPS", "300"))):     # steady state, as the headline step: a fixed count, so that the timed region is the same views every run
        ji.step(views[i % nv][None], K, target, pool[i % 8], ray_sdf[i % 8], up, True, cams[i % nv])
        i += 1
        if i % 10 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    first = i
    marks = []
    t0 = time.perf_counter()
    for i in range(first, first + args.steps):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append(ev)
        n_sdf.append(ji.step(views[i % nv][None], K, target, pool[i % 8], ray_sdf[i % 8], up, True, cams[i % nv])["n_gs_sdf"])
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append(ev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gaps = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
    return {"metric": "train iters/sec, the joint iteration in C++/libtorch (gsdf_extras::JointIteration, " + ("two streams" if two else "one stream") + ")",
            "value": args.steps / el, "unit": "iters/s", "ms_per_step": el / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
            "internal_warmup_steps": first - args.warmup, "n_gpus": 1, "sdf_config": args.sdf_config, "step_terms": args.step_terms,
            "step_ms_hip_events": {"p10": gaps[len(gaps) // 10], "p50": gaps[len(gaps) // 2], "p90": gaps[(len(gaps) * 9) // 10], "max": gaps[-1]},
            "params_finite": bool(torch.isfinite(ji.splat_flat()).all() and torch.isfinite(ji.sdf_flat()).all()),
            "nan_splats_seen_by_prune_test": int(ji.nan_splats_seen().item()),
            "config": {"workload": args.workload, "sdf_points_per_step": 7 * (32768 + sum(n_sdf) / max(1, len(n_sdf)))}}


def reference_loop(args, sc, views, K, ug6, target, N, W, H, deg, dev):
    """The joint iteration (neural_mapping.cpp:400-486) written the way the reference writes it, on top of the drop-in operator
    layer only: what `neural_mapping_node` gets when it is linked against libgsdf_torch.so WITHOUT the gsdf_extras edits of
    INTEGRATION.md section 5 (no fused losses, no fused coupling node, no fused Adam, no second stream).  Python stands in for
    the reference's C++ here: every call below is one libtorch call there."""
    import torch.nn.functional as F
    import gs_sdf_amd.ops as ops
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.neural_gs import update_densify_state
    from gs_sdf_amd.trainer import SplatParams, inject_grads, morton_order
    params = SplatParams.from_scene(sc, dev, morton_order(sc["means"]) if args.splat_order == "morton" else None)
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=1, device=dev, seed=5)
    lm.set_bounds(16.0 - 2 * 0.0625, 0.0625)
    lm.update_octree_as(params.anchors)
    gq = torch.Generator().manual_seed(4)
    pool = [((torch.rand(32768, 3, generator=gq) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev) for _ in range(8)]
    ray_sdf = [(torch.randn(32768, 1, generator=gq) * 0.02).to(dev) for _ in range(8)]
    lrs = dict(offsets=1.6e-4, scaling=5e-3, quaternion=1e-3, opacity=5e-2, features_dc=2.5e-3, features_rest=2.5e-3 / 20)
    opt = torch.optim.Adam([{"params": [params.views[k]], "lr": lrs[k]} for k in params.views] +
                           [{"params": lm.parameters(), "lr": 1e-4}], eps=1e-15)
    # loss_utils.cpp:6-21, 71-117: the reference's 11-tap window (sigma 1.5, its floor((x - 11) / 2) form), per-channel convolutions
    g1 = torch.tensor(ops.ssim_window(), dtype=torch.float32)
    win = (g1[:, None] * g1[None, :]).to(dev)[None, None].expand(3, 1, 11, 11).contiguous()

    def ssim(a, b):
        a, b = a.permute(2, 0, 1)[None], b.permute(2, 0, 1)[None]
        mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
        s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1 * mu1
        s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
        return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()

    gs_state, sizes = {}, []

    def step(i):
        view = views[i % views.shape[0]][None]
        opt.zero_grad()
        # sdf_train_batch_iter (:138-188): sdf_loss + eikonal on get_gradient's numerical branch
        pts, tgt = pool[i % 8], ray_sdf[i % 8]
        s, isig = lm.get_sdf(pts)
        loss = sdfm.sdf_loss(s, tgt, isig) + 0.1 * sdfm.eikonal_loss(lm.get_gradient(pts, 0.02, s, False, True)[0])
        # gs_train_batch_iter (:195-300): generate_gaussian() activations, render, 0.8 L1 + 0.2 D-SSIM
        v = params.views
        xyz, scales, opacity = params.anchors + v["offsets"], torch.exp(v["scaling"]), torch.sigmoid(v["opacity"]).reshape(N)
        dc = v["features_dc"].reshape(N, 1, 3)
        sh = dc if params.n_rest == 0 else torch.cat([dc, v["features_rest"].reshape(N, params.n_rest, 3)], 1)
        colors, alphas, meta = ops.rasterization_2dgs_sdf(xyz, v["quaternion"], scales, opacity, sh, view, K, W, H, near_plane=0.05,
                                                          far_plane=300.0, sh_degree=deg, center_reg=True)
        img = meta["color"][0]
        loss = loss + 0.8 * (img - target).abs().mean() + 0.2 * (1.0 - ssim(img, target)) + inject_grads(
            [(meta["depth"], ug6["v_render_depths"]), (alphas, ug6["v_render_alphas"]),
             (meta["render_normal"], ug6["v_render_normals"]), (meta["render_median"], ug6["v_render_median"])])
        # GS <-> SDF (:420-462): gs_sdf_loss at the visible splats' samples + eikonal at the same (detached) samples
        vis = meta["visibilities"].detach()
        w_all = (meta["samples_weights"] * vis).detach()
        valid = lm.get_valid_mask(meta["samples"].detach()) & (vis > 0.1).squeeze(-1)
        ids = valid.nonzero().squeeze(-1)
        if ids.numel() > 0:
            xs = meta["samples"].index_select(0, ids)
            loss = loss + 1e-3 * sdfm.gs_sdf_loss(lm.get_sdf(xs)[0], w_all.index_select(0, ids))
            loss = loss + 0.1 * sdfm.eikonal_loss(lm.get_gradient(xs.detach(), 0.02, None, False, True)[0])
        loss.backward()
        opt.step()
        update_densify_state(gs_state, meta, N, eager=True)
        sizes.append(int(ids.numel()))

    steps, warm = min(args.steps, 30), min(args.warmup, 5)
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"metric": "train iters/sec, reference loop body on the drop-in operators with zero source edits (NOT the headline)",
            "value": steps / el, "unit": "iters/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warm, "n_gpus": 1,
            "config": {"workload": args.workload, "sdf_points_per_step": 7 * 32768 + 7 * sum(sizes[-steps:]) / steps,
                       "what": "drop-in rasterization_2dgs_sdf / TCNNEncoding / TCNNNetwork + eager torch losses, SSIM, activations, "
                               "numerical get_gradient, update_state, torch.optim.Adam; one stream"}}



def refine_amortised(args, sc, views, K, target, N, W, H, deg, dev):
    """Refinement INSIDE a measured run (NeuralGS::train_callback, neural_gaussian.cpp:568-624): the headline's step
    (gsdf_extras::JointIteration, two streams, per-iteration ray batches) followed every iteration by JointIteration::train_callback —
    prune_invisible_gs every num_train_data iterations, grow_gs (duplicate + split) + prune_gs + zero_state with the Adam-state surgery
    every refine_every iterations after refine_start_iter — with the reference's thresholds (config/base.yaml:60-74) on the synthetic scene.
    The splat count follows the policy (reported); the run stops growing at GSDF_REFINE_MAX_SPLATS (default 4 M) so that a synthetic
    target image that keeps every gradient high cannot exhaust the box."""
    import gs_sdf_amd.hostlib as hostlib
    from benchlib.raybatch import RayBatcher
    from gs_sdf_amd.trainer import SplatParams
    host = hostlib.load()
    params = SplatParams.from_scene(sc, dev, None)
    ji, pool, ray_sdf, cams, _ = make_cpp_iteration(args, sc, params, dev, W, H, deg, views)
    batcher = RayBatcher(host, sc, views, dev, seed=7)
    rc = host.RefineConfig()
    rc.num_train_data = int(views.shape[0])
    steps = int(os.environ.get("GSDF_REFINE_STEPS", "1000"))
    cap = int(os.environ.get("GSDF_REFINE_MAX_SPLATS", "4000000"))
    total_iter = 4 * steps                        # refine_stop_iter = total_iter / 2 lies beyond the run
    nv = views.shape[0]

    def one(i):
        rp, rs = batcher.take(torch.cuda.current_stream())
        ji.step(views[i % nv][None], K, target, rp, rs, [], True, cams[i % nv])
        batcher.issue()

    for i in range(1, 21):                        # warm-up (no refinement below refine_start_iter anyway)
        one(i)
    torch.cuda.synchronize()
    refine_ms, n_hist, events = [], [int(ji.n_splats())], []
    t0 = time.perf_counter()
    for i in range(21, 21 + steps):
        one(i)
        will_refine = i > rc.refine_start_iter and i % rc.refine_every == 0 and ji.n_splats() < cap
        will_prune = i % rc.num_train_data == 0
        if will_refine or will_prune:
            torch.cuda.synchronize()              # so that the refinement step is timed alone (costs the run one drain per refinement)
            t1 = time.perf_counter()
            out = ji.train_callback(i, total_iter, rc)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) * 1e3
            if will_refine:
                refine_ms.append(dt)
            events.append({"iter": i, "ms": round(dt, 3), **{k: int(v) for k, v in out.items()}})
            n_hist.append(int(out["N"]))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    refine_ms.sort()
    step_ms = el / steps * 1e3
    return {"metric": "train iters/sec over a run WITH refinement (train_callback every iteration; grow + split + prune + Adam-state surgery on the flat buffers)",
            "value": steps / el, "unit": "iters/s", "ms_per_step": step_ms, "steps": steps, "n_gpus": 1,
            "refine_steps": len(refine_ms), "refine_ms_median": refine_ms[len(refine_ms) // 2] if refine_ms else None,
            "refine_ms_max": refine_ms[-1] if refine_ms else None,
            "refine_over_normal_step": (refine_ms[len(refine_ms) // 2] / step_ms) if refine_ms else None,
            "splats_first_last_max": [n_hist[0], n_hist[-1], max(n_hist)], "events": events[:12],
            "config": {"refine_start_iter": rc.refine_start_iter, "refine_every": rc.refine_every, "reset_every": rc.reset_every,
                       "num_train_data": rc.num_train_data, "grow_grad2d": rc.grow_grad2d, "prune_opa": rc.prune_opa, "max_splats": cap,
                       "what": "a refinement step = JointIteration::train_callback alone between two device synchronisations: plan (flags + scan + 4 totals in "
                               "host-visible words), randn, apply (every row of parameters, both Adam moments, anchors, statistics written once), rebind"}}
