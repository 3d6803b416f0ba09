#!/usr/bin/env python
"""bench.py — train iters/s of the GS-SDF hot path on MI355X (BASELINE.json metric).

One "step" = one training iteration on one view: activations (exp/sigmoid, a2) -> projection (P1) -> SH colours (P2)
-> tile binning (P3) -> compositing (P4) -> synthetic loss on every rasteriser output; hash-grid SDF leg (per-ray batch
with numerical eikonal + GS<->SDF coupling on the visible splats); backward of everything (P4', P2', P1', S1', S2')
[-> RCCL all-reduce of the flat gradient buffers when N>1] -> fused Adam step on all parameters.
Workload at N=1: BASELINE.json configs[3] shape, "Synthetic 1M Gaussians, 1920x1080" (SURVEY 8d inputs).
N>1: view-parallel (rank r renders view step*N+r), weak scaling, value = views/s of the job.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N, W, H, sh_degree, replica intrinsics)
    "cfg3_1M_1080p": (1_000_000, 1920, 1080, 0, False),
    "cfg1_replica_300k": (300_000, 1200, 680, 0, True),
    "cfg0_10k_256": (10_000, 256, 256, 0, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3_1M_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sdf", action="store_true", help="splat path only (no hash-grid SDF leg)")
    ap.add_argument("--scatter-xcds", type=int, default=2, help="XCDs reserved for the hash-grid backward (overlap mode)")
    ap.add_argument("--dump-grads", default=None, help="test hook: run ONE step without the optimizer update, save the flat "
                                                        "gradient buffers to this file and exit")
    ap.add_argument("--ray-leg-on-scatter-xcds", type=int, default=1, help="run the per-ray SDF leg on the scatter stream's XCDs")
    ap.add_argument("--ray-weights-aux", type=int, default=1, help="decoder weight gradients of the ray leg on the aux stream")
    ap.add_argument("--no-overlap", action="store_true", help="issue the SDF leg on the same HIP stream as the splat leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    # one process per GPU.  (GSDF_BENCH_BACKEND=gloo is a test hook: it lets N ranks share one GPU so that the
    # multi-rank control flow can be exercised on a single-GPU box; RCCL itself refuses two ranks per device.)
    backend = os.environ.get("GSDF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    import gs_sdf_amd.ops as ops
    import gs_sdf_amd.synth as synth
    from gs_sdf_amd.trainer import FusedAdam, GradGate, SplatParams, ViewParallel, inject_grads

    N, W, H, deg, replica = WORKLOADS[args.workload]
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica)
    views = synth.make_views(200, seed=1).to(dev)
    K = sc["K"].to(dev)
    params = SplatParams.from_scene(sc, dev)
    ug = {k: v.to(dev) for k, v in synth.upstream_grads(H, W, seed=2).items()}
    ug6 = {k: (1e-6 * v).contiguous() for k, v in ug.items()}       # the 1e-6 N(0,1) op-level upstream gradients, fixed
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)    # target image U(0,1) seed 3
    groups = []
    if not args.no_sdf:
        # hash-grid SDF (2^19 table, 16 levels x 2) + fused MFMA decoder; the reference's numerical-gradient
        # configuration (params.cpp:396-399 forces it for the tcnn decoder); ray batch 32768 (base.yaml:24)
        import gs_sdf_amd.sdf as sdfm
        lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=1, device=dev, seed=5)
        groups.append(lm.flatten(accumulate_table_grad_in_place=True))
        # occupancy structure of the map (SubMap::update_octree_as, sub_map.cpp:22-35): leaf 1/16 m -> level 8 in the 16 m
        # cube, built from the splat centres (the reference builds it from the depth point cloud the splats start from)
        lm.set_bounds(16.0 - 2 * 0.0625, 0.0625)
        lm.update_octree_as(params.anchors)
        gq = torch.Generator().manual_seed(4)
        pool = [((torch.rand(32768, 3, generator=gq) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev) for _ in range(8)]
        ray_sdf = [(torch.randn(32768, 1, generator=gq) * 0.02).to(dev) for _ in range(8)]
    vp = ViewParallel(params, dist, groups)
    # optimizer: fused Adam over the flat buffers, the reference's groups / learning rates (neural_gaussian.cpp:434-453;
    # SDF groups at min(xyz_lr, lr_end) during the joint stage, :619-623), eps 1e-15
    adam, adam_sdf = FusedAdam(eps=1e-15), FusedAdam(eps=1e-15)       # one per leg: each steps on its leg's stream
    lrs = dict(offsets=1.6e-4, scaling=5e-3, quaternion=1e-3, opacity=5e-2, features_dc=2.5e-3, features_rest=2.5e-3 / 20)
    adam.add_group(params.flat, params.flat_grad, [(params.views[k].numel(), lrs[k]) for k in params.views])
    for gsdf_group in groups:
        adam_sdf.add_group(gsdf_group.flat, gsdf_group.flat_grad, [(gsdf_group.flat.numel(), 1e-4)])

    sizes = {}

    # Two legs on two HIP streams.  The SDF leg (hash grid + MLP; its scatter is bound by the memory-side fp32 atomic units)
    # runs beside the splat leg (rasteriser; bound by VALU issue).  They are the reference's own loss groups
    # (neural_mapping.cpp:138-188 and :420-462 against :195-300) and touch in two places only: the visible splats' sample
    # points go splat -> SDF after compositing, and their gradient comes back SDF -> splat where it enters the projection
    # backward (trainer.join_grad).  Each leg owns its parameters, gradients and optimizer, so step i+1's splat leg does
    # not wait for step i's hash-grid scatter.  --no-overlap issues the identical work on one stream.
    # The hash-grid scatter kernel gets XCDs of its own (default 2 of 8) and both legs stay off them: beside it, any
    # kernel that shares an XCD with it finishes only when it does (gs_sdf_amd/streams.py has the measurements).
    overlap = not args.no_sdf and not args.no_overlap
    main = torch.cuda.current_stream()
    side = scatter = aux = main
    if not args.no_sdf:
        lm.encoder.save_jacobian = True      # d/dx of the sample points from the forward's Jacobian (first order only)
    if overlap:
        from gs_sdf_amd.streams import xcd_partition_streams
        try:
            (main, side, aux), scatter = xcd_partition_streams(args.scatter_xcds, 3)
        except Exception as e:      # CU masks unavailable: same schedule on ordinary HIP streams (slower, still correct)
            print(f"[bench] XCD-partitioned streams unavailable ({e}); using unmasked streams", file=sys.stderr, flush=True)
            main, side, aux, scatter = (torch.cuda.Stream() for _ in range(4))
        lm.encoder.scatter_stream = scatter
        lm.decoder.aux_stream = aux          # decoder weight gradients: off the chain that leads back to the splat leg
        main.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main)
    gate = GradGate()

    def release_streams():
        if overlap:
            from gs_sdf_amd.streams import destroy_all
            torch.cuda.synchronize()
            lm.encoder.scatter_stream = lm.decoder.aux_stream = None
            torch.cuda.set_stream(torch.cuda.default_stream())
            destroy_all()

    host = [] if os.environ.get("GSDF_BENCH_HOST_TIMES") else None   # debugging aid: host-side issue times per segment

    def stamp(tag):
        if host is not None:
            host.append((tag, time.perf_counter()))

    def step(i, update=True):
        stamp("begin")
        view = views[(i * world + rank) % views.shape[0]][None]
        if not args.no_sdf:
            # per-ray SDF batch (neural_mapping.cpp:138-188): BCE on the SDF head + eikonal on the numerical gradient.
            # Independent of the render.  In overlap mode the WHOLE leg (encoder, decoder, loss, backward, scatter) runs on
            # the scatter stream's two XCDs: they would otherwise idle until the first scatter of the step, and the
            # six XCDs of the splat leg are relieved of ~0.7 ms of kernels.
            ray_stream = scatter if args.ray_leg_on_scatter_xcds else side
            if ray_stream is not side:
                ray_stream.wait_stream(side)              # the SDF parameters of step i-1 (Adam ran on `side`)
            aux_saved, lm.decoder.aux_stream = lm.decoder.aux_stream, (None if (ray_stream is scatter and not args.ray_weights_aux) else lm.decoder.aux_stream)
            with torch.cuda.stream(ray_stream):
                pts, tgt = pool[i % 8], ray_sdf[i % 8]
                lm.ray_loss(pts, tgt, 0.02, 0.1).backward()
            lm.decoder.aux_stream = aux_saved
        stamp("ray leg issued")
        xyz, quat, scales, opacity, sh = params.activated()
        colors, alphas, meta = ops.rasterization_2dgs_sdf(xyz, quat, scales, opacity, sh, view, K, W, H, near_plane=0.05,
                                                          far_plane=300.0, sh_degree=deg, center_reg=True, samples_gate=gate)
        if not args.no_sdf:
            # GS <-> SDF coupling (neural_mapping.cpp:420-462): SDF at the visible splats' samples.  This is the longest
            # dependency chain of the step (compositing -> visible set -> encoder -> decoder -> back), so it is issued first.
            stamp("render issued (2 syncs)")
            vis = meta["visibilities"].detach()
            w_all = (meta["samples_weights"] * vis).detach()
            valid = lm.get_valid_mask(meta["samples"].detach()) & (vis > 0.1).squeeze(-1)      # neural_mapping.cpp:430-432
            ids = valid.nonzero().squeeze(-1)
            stamp("visible set (sync)")
            fwd_done = main.record_event()
            samples = meta["samples"]                                 # already behind join_grad(gate)
            samples_cut = samples.detach().requires_grad_(True)      # graph cut: same maths, two backward legs
            sizes.update(n_gs_sdf=int(ids.numel()))
            side.wait_event(fwd_done)
            if side is not main:
                for t in (samples_cut, w_all, ids):     # allocated on `main`, read by kernels on `side`: keep the blocks
                    t.record_stream(side)               # out of main's allocator until side has passed this point
            with torch.cuda.stream(side):
                if ids.numel() > 0:
                    lm.gs_sdf_coupling(samples_cut, ids, w_all, 1e-3).backward()
                gate.event = side.record_event() if side is not main else None     # d loss / d samples is complete
            stamp("samples leg issued")
        # colour: the reference's photometric loss 0.8 L1 + 0.2 D-SSIM (neural_mapping.cpp:237-240), fused HIP kernel;
        # depth / alpha / normal / median: op-level 1e-6 N(0,1) upstream gradients so that every backward path is live
        loss = ops.l1_dssim_loss(meta["color"][0], target, 0.8, 0.2) + inject_grads(
            [(meta["depth"], ug6["v_render_depths"]), (alphas, ug6["v_render_alphas"]),
             (meta["render_normal"], ug6["v_render_normals"]), (meta["render_median"], ug6["v_render_median"])])
        stamp("loss issued")
        if not args.no_sdf and samples_cut.grad is not None:
            if side is not main:
                samples_cut.grad.record_stream(main)    # allocated on `side`, read by the projection backward on `main`
            torch.autograd.backward([loss, samples], [None, samples_cut.grad])
        else:
            loss.backward()
        stamp("backward issued")
        if world > 1:
            main.wait_stream(scatter)      # RCCL's workgroups land on every XCD: do not run them beside the scatter
        vp.all_reduce_group(params)
        if update:
            adam.step()
            params.flat_grad.zero_()
        if not args.no_sdf:
            # the SDF network's gradients are final once the scatter stream has drained: all-reduce (61 MB table + MLP)
            # and optimizer step of this leg on its own stream, beside the splat leg
            with torch.cuda.stream(side):
                side.wait_stream(scatter)
                side.wait_stream(aux)
                vp.all_reduce_group(groups[0])
                if update:
                    adam_sdf.step()
                    groups[0].flat_grad.zero_()
        stamp("optimizers issued")
        sizes.update(M=int(meta["gaussian_ids"].shape[0]), I=int(meta["flatten_ids"].shape[0]))

    vp.zero_grad()
    if args.dump_grads:
        step(0, update=False)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"splat": params.flat_grad.cpu(), "sdf": [g.flat_grad.cpu() for g in groups], "sizes": dict(sizes)},
                       args.dump_grads)
        release_streams()
        if dist is not None:
            dist.destroy_process_group()
        return
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP-event timing of the roofline kernels over the timed region (the full per-operator table comes from a short
    # separate pass below: two events per launch on all ~25 operators cost ~2 % of the step in host time)
    ROOF = {"hashgrid_bwd", "hashgrid_fwd", "rasterize_2dgs_fwd", "rasterize_2dgs_bwd"}
    ops.TIMERS.enable(only=ROOF)
    if host is not None:
        host.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if host is not None and rank == 0:
        import collections
        acc, n = collections.OrderedDict(), 0
        for (ta, a), (tb, b) in zip(host[:-1], host[1:]):
            if tb != "begin":
                acc[tb] = acc.get(tb, 0.0) + (b - a)
            n += tb == "optimizers issued"
        print("host ms/step: " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items()), file=sys.stderr, flush=True)
    kern = ops.TIMERS.summary_ms()
    calls = ops.TIMERS.calls()
    ops.TIMERS.enable()                       # every operator, outside the timed region
    for i in range(min(10, args.steps)):
        step(args.warmup + args.steps + i)
    kern_all = ops.TIMERS.summary_ms()
    ops.TIMERS.disable()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        M, I = sizes["M"], sizes["I"]
        P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        Kb = (deg + 1) ** 2
        # algorithmic bytes per launch.  The dominant kernel is the one with the
        # largest total time per step (average launch x launches per step).
        alg = {"rasterize_2dgs_bwd": 80 * I + 48 * P + 88 * M, "rasterize_2dgs_fwd": 80 * I + 48 * P}
        if not args.no_sdf:
            # S1 per query point: fwd 12 + 1024 (16 levels x 8 corners x 8 B) + 128; bwd 8 + 128 + 1024 scatter; averaged
            # over the launches of a step (7 x 32768 ray + stencil points in one launch, the visible splat samples in another)
            pts = (7 * 32768 + sizes.get("n_gs_sdf", 0)) / max(1.0, calls.get("hashgrid_bwd", 0) / args.steps)
            alg["hashgrid_bwd"] = int(1160 * pts)
            alg["hashgrid_fwd"] = int(1164 * pts)
        per_step = {k: kern.get(k, 0.0) * calls.get(k, 0) / args.steps for k in alg}
        dom = max(alg, key=lambda k: per_step[k])
        dur_ms = kern.get(dom, float("nan"))
        achieved = alg[dom] / (dur_ms * 1e-3) / 1e9
        b_splat = (80 + 12 * Kb) * N + (364 + 12 * Kb) * M + 204 * I + 96 * P + 4 * T
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
        out = {
            "metric": "train iters/sec (splat raster + SDF fwd+bwd), 1M Gaussians @1080p" if not args.no_sdf
                      else "train iters/sec (splat raster fwd+bwd only), 1M Gaussians @1080p",
            "value": args.steps * world / elapsed, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} random Gaussians, {W}x{H}, sh_degree {deg}, 1 view/GPU/step, "
                                   f"M={M} I={I} L={I / T:.0f}" + ("" if args.no_sdf else f"; + hash-grid SDF (2^19 x16x2, fused 64-wide MLP): 32768 ray "
                                   f"points x7 (numerical eikonal) + {sizes.get('n_gs_sdf', 0)} splat samples"), "parallelism": f"view-parallel x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "algorithmic_bytes": alg[dom],
                         "avg_launch_ms": dur_ms, "launches_per_step": calls.get(dom, 0) / args.steps,
                         "ms_per_step_by_kernel": {k: round(v, 4) for k, v in per_step.items()},
                         # the same figure for the other large kernels (the scatter runs beside the rest of the step on
                         # XCDs of its own; the compositing backward is the largest kernel on the splat leg)
                         "others": {k: {"achieved": alg[k] / (kern[k] * 1e-3) / 1e9, "frac": alg[k] / (kern[k] * 1e-3) / 8e12,
                                        "avg_launch_ms": kern[k]} for k in alg if k != dom and kern.get(k)},
                         # the scatter is bound by the memory-side fp32 atomic units, not by HBM bytes: cache-line requests per
                         # launch (16 levels x 4 (y,z) corner pairs x 9/8 lines: the two x-neighbours share a 64 B line unless
                         # x0 % 8 == 7) against the measured chip-wide ceiling (tools/ubench/atomic_rate.hip)
                         "atomic_line_rate": (None if dom != "hashgrid_bwd" else {
                             "achieved_G_per_s": 72.0 * (alg[dom] / 1160.0) / (dur_ms * 1e-3) / 1e9, "ceiling_G_per_s": 21.0,
                             "frac": 72.0 * (alg[dom] / 1160.0) / (dur_ms * 1e-3) / 21e9,
                             "note": "kernel confined to 2 of 8 XCDs in the two-leg step"}),
                         "step_B_splat_bytes": b_splat,
                         "step_hbm_frac": b_splat / (elapsed / args.steps) / 8e12},
            "params_finite": bool(torch.isfinite(params.flat).all()) and all(bool(torch.isfinite(g.flat).all()) for g in groups),
            "hbm_gb": {"allocated_peak": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                       "reserved": round(torch.cuda.memory_reserved() / 2 ** 30, 2)},
            "kernel_ms": kern_all, "kernel_ms_note": "average launch duration per operator over 10 extra steps after the timed region",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, views[0:1].cpu(), N, W, H, deg,
                                               0 if args.no_sdf else 7 * 32768 + sizes.get("n_gs_sdf", 0))
        print(json.dumps(out), flush=True)
    release_streams()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(sc, view, N, W, H, deg, n_sdf_points=0):
    """The oracle ("port": the reference has no CPU rasteriser and none of its kernels are vendored) timed on
    this box's host cores: ONE full iteration (fwd + bwd) of the same workload."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    n = lambda t: t.detach().cpu().numpy()
    import gs_sdf_amd.synth as synth
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
    orc.projection_2dgs_bwd(means, quats, scales, n(view), n(sc["K"]), W, H, p["camera_ids"], p["gaussian_ids"],
                            g["v_means2d"].astype(np.float32), np.zeros(M, np.float32),
                            g["v_ray_transforms"].astype(np.float32), g["v_normals"].astype(np.float32))
    orc.view_colors_bwd(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, g["v_colors"].astype(np.float32))
    if n_sdf_points:
        # SDF leg: hash-grid + decoder forward and backward on the same number of query points
        rng = np.random.default_rng(4)
        offs, total = orc.grid_offsets()
        table = ((rng.random((total, 2), dtype=np.float32) * 2 - 1) * 1e-4).astype(np.float32)
        dims = [32, 64, 64, 64, 2]
        Wm = (rng.standard_normal(sum(i * o for i, o in zip(dims[:-1], dims[1:]))) * 0.1).astype(np.float32)
        xs = rng.random((n_sdf_points, 3), dtype=np.float32)
        feat = orc.grid_fwd(xs, table)
        o = orc.mlp_fwd(feat, dims, Wm, None)
        v_in, v_w, _ = orc.mlp_bwd(feat, dims, Wm, None, np.ones_like(o))
        orc.grid_bwd(xs, table, v_in)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"1 full iteration (fwd+bwd) of the same workload, oracle/splat_oracle.c f32 build, OpenMP over tiles "
                      f"on {cores} threads for compositing (projection/sort single-threaded)"
                      + (f" + SDF oracle fwd+bwd on {n_sdf_points} query points" if n_sdf_points else "") + f", {dt:.1f} s"}


if __name__ == "__main__":
    main()
