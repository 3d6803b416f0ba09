#!/usr/bin/env python
"""bench.py — train iters/s of the GS-SDF hot path on MI355X (BASELINE.json metric).

One "step" = one training iteration on one view: activations (exp/sigmoid, a2) -> projection (P1) -> SH colours (P2)
-> tile binning (P3) -> compositing (P4) -> synthetic loss on every rasteriser output; hash-grid SDF leg (per-ray batch
with numerical eikonal + GS<->SDF coupling on the visible splats); backward of everything (P4', P2', P1', S1', S2')
[-> RCCL all-reduce of the flat gradient buffers when N>1] -> fused Adam step on all parameters.
Workload at N=1: BASELINE.json configs[3] shape, "Synthetic 1M Gaussians, 1920x1080" (SURVEY 8d inputs).
N>1: view-parallel (rank r renders view step*N+r), weak scaling, value = views/s of the job.

Rank 0 prints ONE compact JSON line (<= 4 KB, benchlib/report.py: compact_line) as the LAST stdout line, with `roofline` and `cpu_baseline`;
secondary lines are printed before it as `[secondary] name {...}`; the full record (per-kernel tables, the whole parity report, every secondary
line) goes to gpurun_out/bench_detail.json.  The helpers live in benchlib/.
"""
import argparse
import contextlib
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
    "cfg1_replica_300k": (300_000, 1200, 680, 0, True),         # configs[1] with --no-sdf, configs[2] (joint train) without
    "cfg4_3M_640x512_K16": (3_000_000, 640, 512, 3, False),     # configs[4] shape on ONE GPU (FAST-LIVO2: 640x512, sh_degree 3)
    "cfg0_10k_256": (10_000, 256, 256, 0, False),
    # larger footprints than the SURVEY 8d scene (projected 1-sigma 2-12 px instead of 0.5-4: ~12 M (tile, splat) pairs, L ~ 1500): the hard end
    # for the list kernels; a secondary line, never the headline
    "stress_1M_1080p_sigma2_12": (1_000_000, 1920, 1080, 0, False),
}
WORKLOAD_SIGMA_PX = {"stress_1M_1080p_sigma2_12": (2.0, 12.0)}      # synth.make_scene(sigma_px=...); default (0.5, 4.0)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run with N ranks on this
    node (127.0.0.1 rendezvous, a free port), one process per GPU.  Fails loudly when the box has fewer than N GPUs (RCCL needs
    a device per rank; GSDF_BENCH_BACKEND=gloo is the test hook that lets ranks share a device).  Returns the exit code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if os.environ.get("GSDF_BENCH_BACKEND", "nccl") == "nccl" and n_dev < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs (one process per GPU over RCCL), torch.cuda.device_count() = {n_dev}; "
              "refusing to report a smaller job as an N-GPU number", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3_1M_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sdf", action="store_true", help="splat path only (no hash-grid SDF leg)")
    ap.add_argument("--scatter-xcds", type=int, default=0, help="XCDs reserved for the hash-grid backward (overlap mode); 0 = no CU "
                                                                "masks, every stream sees the whole chip (the binned scatter is "
                                                                "bandwidth-bound, not atomic-bound: it wants all XCDs)")
    ap.add_argument("--dump-grads", default=None, help="test hook: run ONE step without the optimizer update, save the flat "
                                                        "gradient buffers to this file and exit")
    ap.add_argument("--ray-leg-on-scatter-xcds", type=int, default=1, help="run the per-ray SDF leg on the scatter stream's XCDs")
    ap.add_argument("--ray-weights-aux", type=int, default=1, help="decoder weight gradients of the ray leg on the aux stream")
    ap.add_argument("--no-overlap", action="store_true", help="issue the SDF leg on the same HIP stream as the splat leg")
    ap.add_argument("--sdf-config", default="default", choices=["default", "tcnn"],
                    help="default = the reference's shipped configuration (config/base.yaml:12-13: decoder_implementation 0, biased "
                         "5-layer decoder; numerical_grad 0, eikonal on the ANALYTIC gradient by double backward + align_weight 0.1 against "
                         "the detached numerical gradient); tcnn = decoder_implementation 1 (bias-free FullyFusedMLP, for which the "
                         "reference forces the numerical gradient, params.cpp:396-399)")
    ap.add_argument("--step-terms", default="reference", choices=["reference", "round2"],
                    help="reference = every loss term of the reference's iteration (0.8 L1 + 0.2 D-SSIM, render_normal_weight 0.01 x "
                         "depth->normal consistency, isotropic_weight 0.05, prune_nan test); round2 = the round-2 step (L1 + D-SSIM and "
                         "1e-6 N(0,1) op-level gradients on depth / alpha / normal / median)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra lines (other SDF configuration, zero-edit loop, C++ step)")
    ap.add_argument("--step-impl", default="cpp", choices=["cpp", "python"],
                    help="who issues the step: cpp = gsdf_extras::JointIteration (C++/libtorch over libgsdf_torch.so: the host language of the "
                         "reference, the step its node reaches with the INTEGRATION.md section 5 edits; THE HEADLINE), python = the Python "
                         "mirror of the same operators (tests, and the secondary line `python_mirror_step`)")
    ap.add_argument("--cpp-step", action="store_true",
                    help="time gsdf_extras::JointIteration (the same joint iteration in C++/libtorch over libgsdf_torch.so, one stream, driven "
                         "through the pybind test harness): the step the reference's node reaches with the INTEGRATION.md section 5 edits")
    ap.add_argument("--reference-loop", action="store_true",
                    help="time the loop body as neural_mapping_node would run it linked against the drop-in submodules with ZERO "
                         "source edits: the drop-in operators (rasterization_2dgs_sdf, TCNNEncoding, TCNNNetwork) composed with "
                         "eager torch for everything the reference does in libtorch (losses, SSIM, activations, get_gradient's "
                         "numerical branch, update_state, torch.optim.Adam), one stream.  NOT the headline; reported as a line of its own")
    ap.add_argument("--sample-mode", default="center", choices=["center", "stochastic"],
                    help="SDF samples of the visible splats: center = k_center_reg 1 (splat centres, weight 1: the fully specified mode the "
                         "parity / benchmark runs use, SURVEY appendix A.1); stochastic = the reference's default (center_reg absent from "
                         "config/base.yaml): one random point on every visible splat's disc, weight exp(-|eps|^2/2)")
    ap.add_argument("--ray-batch", default="sampled", choices=["sampled", "pool"],
                    help="sampled (default, C++ step): the reference's per-iteration ray-batch construction inside the timed step — random rays "
                         "gathered from a host-side depth-ray pool, H2D copy, octree ray march + free / surface / end-point samples, truncation, "
                         "in-range filter, throttled to ~32768 points (neural_mapping.cpp:138-164, 73-104, 324-330); pool: 8 pre-generated batches "
                         "of 32768 uniform points (rounds 1-3; what --dump-grads and the Python mirror use)")
    ap.add_argument("--splat-order", default="as-given", choices=["morton", "as-given"],
                    help="memory order of the splat set: the synthetic scene's random order, or Morton order of the centres "
                         "(trainer.morton_order; measured: no gain, the fine hash-grid levels scatter either way, DESIGN.md section 11)")
    ap.add_argument("--deterministic", action="store_true", help="gsdf_deterministic(1): order-independent accumulation (fixed-point compositing "
                    "gradients, ordered loss reductions, partial-buffer decoder gradients) — a validation mode, ~2 ms per step")
    ap.add_argument("--step-trace", default="", help="write a device timeline of the C-ABI entry points (begin us, end us, duration, name; HIP events, no "
                    "profiler) of six steps after the timed region to this file")
    ap.add_argument("--hashgrid-resident", type=int, default=-1,
                    help="C++ step, two streams: workgroups per CU of the stencil hash-grid forward's resident grid (JointConfig::hashgrid_resident; "
                         "-1 = its default, 0 = the full grid)")
    ap.add_argument("--samples-grad-first", type=int, default=-1,
                    help="C++ step: 1 = the SDF node sends the samples' gradient on its way before the regularisers' launches "
                         "(JointConfig::samples_grad_first, the default), 0 = the whole first order at backward time")
    ap.add_argument("--collective", default="all_reduce", choices=["all_reduce", "reduce_scatter_all_gather"],
                    help="N > 1: how a parameter family's flat gradient buffer is summed over the ranks")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked as `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU
        sys.exit(spawn_ranks(args.gpus))
    if args.deterministic:
        import gs_sdf_amd.capi as _capi
        _capi.lib().gsdf_deterministic(1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    n_dev = torch.cuda.device_count()
    if os.environ.get("GSDF_BENCH_BACKEND", "nccl") == "nccl" and n_dev < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible GPUs (one process per GPU over RCCL), "
                         f"torch.cuda.device_count() = {n_dev}")
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
    from gs_sdf_amd.neural_gs import update_densify_state

    N, W, H, deg, replica = WORKLOADS[args.workload]
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica, sigma_px=WORKLOAD_SIGMA_PX.get(args.workload, (0.5, 4.0)))
    views = synth.make_views(200, seed=1).to(dev)
    K = sc["K"].to(dev)
    from gs_sdf_amd.trainer import morton_order
    order = morton_order(sc["means"]) if args.splat_order == "morton" else None
    params = SplatParams.from_scene(sc, dev, order)
    ug = {k: v.to(dev) for k, v in synth.upstream_grads(H, W, seed=2).items()}
    ug6 = {k: (1e-6 * v).contiguous() for k, v in ug.items()}       # the 1e-6 N(0,1) op-level upstream gradients, fixed
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)    # target image U(0,1) seed 3
    from benchlib.raybatch import RayBatcher
    from benchlib.report import CABI_OPS, ROOF_OPS as ROOF, cabi_timing_to_ops
    from benchlib.steps import cpp_step, make_cpp_iteration, reference_loop
    if args.reference_loop:
        print(json.dumps(reference_loop(args, sc, views, K, ug6, target, N, W, H, deg, dev)), flush=True)
        return
    if args.cpp_step:
        print(json.dumps(cpp_step(args, sc, views, K, ug6, target, N, W, H, deg, dev)), flush=True)
        return
    impl = "python" if (args.no_sdf or args.scatter_xcds > 0) else args.step_impl     # the splat-only / CU-mask experiments exist in the mirror only
    ji = None
    if impl == "cpp":
        ji, cpp_pool, cpp_ray_sdf, cpp_cams, dec_dims = make_cpp_iteration(args, sc, params, dev, W, H, deg, views)
        cpp_up = [] if args.step_terms == "reference" else [ug6[k] for k in ("v_render_depths", "v_render_alphas", "v_render_normals", "v_render_median")]
        coll_events = {"splat": [], "sdf": []}
        if dist is not None:      # view-parallel: one collective per parameter family, on the stream its optimizer runs on
            def _hook(family):
                def mean_over_ranks(g):
                    # issued on the CURRENT stream = the leg that owns the family (JointIteration::set_grad_hooks); timed with events on that stream
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    if backend == "nccl":
                        dist.all_reduce(g, op=dist.ReduceOp.AVG)     # RCCL averages inside the collective: no second pass over the buffer
                    else:                                            # gloo (CPU-side test runs) has no AVG
                        dist.all_reduce(g)
                        g.mul_(1.0 / world)
                    b.record()
                    coll_events[family].append((a, b, g.numel() * 4))
                return mean_over_ranks
            ji.set_grad_hooks(_hook("splat"), _hook("sdf"))
    batcher = None
    if impl == "cpp" and args.ray_batch == "sampled" and not args.dump_grads:
        import gs_sdf_amd.hostlib as hostlib
        batcher = RayBatcher(hostlib.load(), sc, views, dev, seed=7 + 1009 * rank)      # every rank draws its own rays (seed xor rank)
    groups = []
    if not args.no_sdf and impl == "python":
        # hash-grid SDF (2^19 table, 16 levels x 2) + fused MFMA decoder; the reference's numerical-gradient
        # configuration (params.cpp:396-399 forces it for the tcnn decoder); ray batch 32768 (base.yaml:24)
        import gs_sdf_amd.sdf as sdfm
        lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0 if args.sdf_config == "default" else 1,
                           device=dev, seed=5)
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

    sizes, hist, all_hist = {}, {}, {}

    def note(k, v):
        hist.setdefault(k, []).append(v)
        all_hist.setdefault(k, []).append(v)

    # Two legs on two HIP streams.  The SDF leg (hash grid + MLP; its scatter is bound by the memory-side fp32 atomic units)
    # runs beside the splat leg (rasteriser; bound by VALU issue).  They are the reference's own loss groups
    # (neural_mapping.cpp:138-188 and :420-462 against :195-300) and touch in two places only: the visible splats' sample
    # points go splat -> SDF after compositing, and their gradient comes back SDF -> splat where it enters the projection
    # backward (trainer.join_grad).  Each leg owns its parameters, gradients and optimizer, so step i+1's splat leg does
    # not wait for step i's hash-grid scatter.  --no-overlap issues the identical work on one stream.
    # The hash-grid scatter kernel gets XCDs of its own (default 2 of 8) and both legs stay off them: beside it, any
    # kernel that shares an XCD with it finishes only when it does (gs_sdf_amd/streams.py has the measurements).
    overlap = not args.no_sdf and not args.no_overlap and impl == "python"
    main = torch.cuda.current_stream()
    side = scatter = aux = main
    if not args.no_sdf and impl == "python":
        lm.encoder.save_jacobian = True      # d/dx of the sample points from the forward's Jacobian (first order only)
    if overlap:
        from gs_sdf_amd.streams import xcd_partition_streams
        if args.scatter_xcds > 0:
            try:
                (main, side, aux), scatter = xcd_partition_streams(args.scatter_xcds, 3)
            except Exception as e:      # CU masks unavailable: same schedule on ordinary HIP streams
                print(f"[bench] XCD-partitioned streams unavailable ({e}); using unmasked streams", file=sys.stderr, flush=True)
                main, side, aux, scatter = (torch.cuda.Stream() for _ in range(4))
        else:
            # the splat leg's ~60 short kernels per step gate the start of the SDF leg (the visible set) and of the next step:
            # a high-priority queue keeps them from waiting behind the SDF leg's multi-millisecond kernels
            prio = int(os.environ.get("GSDF_SPLAT_STREAM_PRIORITY", "-1"))
            main = torch.cuda.Stream(priority=prio)
            sprio = int(os.environ.get("GSDF_SDF_STREAM_PRIORITY", "0"))
            side, aux, scatter = (torch.cuda.Stream(priority=sprio) for _ in range(3))
        lm.encoder.scatter_stream = scatter
        lm.decoder.aux_stream = aux          # decoder weight gradients: off the chain that leads back to the splat leg
        main.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main)
    elif impl == "cpp" and not args.no_overlap:
        # JointIteration issues the splat leg on the caller's stream and the SDF network's work on a pool stream of its own.  A
        # high-priority queue for the splat leg (what the mirror's four-stream schedule uses) was measured SLOWER here: 141.5 / 142.2
        # against 149.8 it/s, cfg3, 40 steps — the SDF leg is the longer one in this schedule and loses more than the splat leg gains
        prio = int(os.environ.get("GSDF_CPP_SPLAT_STREAM_PRIORITY", "0"))
        if prio != 0:
            main = torch.cuda.Stream(priority=prio)
            main.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(main)
    gate = GradGate()

    released = []

    def release_streams():
        if overlap and not released:
            released.append(1)
            from gs_sdf_amd.streams import destroy_all
            torch.cuda.synchronize()
            lm.encoder.scatter_stream = lm.decoder.aux_stream = None
            torch.cuda.set_stream(torch.cuda.default_stream())
            destroy_all()

    host = [] if os.environ.get("GSDF_BENCH_HOST_TIMES") else None   # debugging aid: host-side issue times per segment

    def stamp(tag):
        if host is not None:
            host.append((tag, time.perf_counter()))

    gs_state = {}
    analytic = args.sdf_config == "default"
    ref_terms = args.step_terms == "reference"
    Kh = [float(v) for v in (sc["K"][0, 0, 0], sc["K"][0, 1, 1], sc["K"][0, 0, 2], sc["K"][0, 1, 2])]
    c2w_host = [tuple(float(v) for v in torch.linalg.inv(vw.double())[:3, :4].reshape(-1)) for vw in views.cpu()]   # poses are known ahead
    nan_total = torch.zeros(1, dtype=torch.int32, device=dev)

    def step(i, update=True):
        stamp("begin")
        if impl == "cpp":
            vi = (i * world + rank) % views.shape[0]
            if batcher is not None:
                rp, rs = batcher.take(torch.cuda.current_stream())
            else:
                rp, rs = cpp_pool[i % 8], cpp_ray_sdf[i % 8]
            sz = ji.step(views[vi][None], K, target, rp, rs, cpp_up, update, cpp_cams[vi])
            if batcher is not None:
                batcher.issue()            # the NEXT step's batch, while this step runs (see RayBatcher)
            note("n_ray_pts", int(rp.shape[0]))
            for k in ("M", "I", "n_gs_sdf"):
                note(k, int(sz[k]))
            sizes.update({k: int(v) for k, v in sz.items()})
            return
        view = views[(i * world + rank) % views.shape[0]][None]
        if not args.no_sdf and not analytic:
            # per-ray SDF batch (neural_mapping.cpp:138-188): BCE on the SDF head + eikonal on the numerical gradient.
            # Independent of the render.  In overlap mode the WHOLE leg (encoder, decoder, loss, backward, scatter) runs on
            # the scatter stream: it would otherwise idle until the first scatter of the step.
            # (analytic configuration: the ray batch travels with the splat samples in ONE batch, below)
            ray_stream = scatter if args.ray_leg_on_scatter_xcds else side
            if ray_stream is not side:
                ray_stream.wait_stream(side)              # the SDF parameters of step i-1 (Adam ran on `side`)
            aux_saved, lm.decoder.aux_stream = lm.decoder.aux_stream, (None if (ray_stream is scatter and not args.ray_weights_aux) else lm.decoder.aux_stream)
            with torch.cuda.stream(ray_stream):
                pts, tgt = pool[i % 8], ray_sdf[i % 8]
                with sdfm.grad_sinks_armed():
                    lm.ray_loss(pts, tgt, 0.02, 0.1).backward()
            lm.decoder.aux_stream = aux_saved
        stamp("ray leg issued")
        xyz, quat, scales, opacity, sh = params.activated()
        colors, alphas, meta = ops.rasterization_2dgs_sdf(xyz, quat, scales, opacity, sh, view, K, W, H, near_plane=0.05,
                                                          far_plane=300.0, sh_degree=deg, center_reg=(args.sample_mode == "center"),
                                                          sample_seed=1 + i, samples_gate=gate)
        # colour: the reference's photometric loss 0.8 L1 + 0.2 D-SSIM (neural_mapping.cpp:237-240), fused HIP kernel;
        # depth / alpha / normal / median: op-level 1e-6 N(0,1) upstream gradients so that every backward path is live.
        # Issued BEFORE the coupling leg: its kernels only need the render, and the host spends ~1 ms issuing that leg.
        loss = ops.l1_dssim_loss(meta["color"][0], target, 0.8, 0.2)
        if ref_terms:
            # render_normal_weight x depth->normal consistency (neural_mapping.cpp:243-266, depth_type 0: expected depth; alpha
            # detached) + isotropic_weight x isotropic regulariser of the visible splats (:268-276): one fused launch each
            vi = (i * world + rank) % views.shape[0]
            loss = loss + 0.01 * ops.normal_consistency_loss(meta["depth"][0], alphas[0], meta["render_normal"][0], *Kh, c2w_host[vi]) \
                        + 0.05 * ops.isotropic_loss(scales, meta["gaussian_ids"])
        else:
            loss = loss + inject_grads([(meta["depth"], ug6["v_render_depths"]), (alphas, ug6["v_render_alphas"]),
                                        (meta["render_normal"], ug6["v_render_normals"]), (meta["render_median"], ug6["v_render_median"])])
        stamp("render + loss issued (2 syncs)")
        if not args.no_sdf:
            # GS <-> SDF coupling (neural_mapping.cpp:420-462): SDF at the visible splats' samples: gs_sdf_loss on the base
            # points + (full step) the eikonal regulariser on the numerical gradient at the same points, i.e. 6 more
            # encoder / decoder evaluations per sample (sdf_regularization(gs_samples.detach(), ...), :448-451 -> :106-136)
            vis = meta["visibilities"].detach()
            # samples_weights * visibilities, get_valid_mask(samples) & (visibilities > 0.1), nonzero (neural_mapping.cpp:423-437): 3 launches
            ids, w_all = lm.acc_struct_occ.visible_set(meta["samples"], vis, meta["samples_weights"], 0.1, lm._origin, lm.map_size_inv)
            w_all = w_all[:, None]
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
                if analytic:
                    # the iteration's whole SDF work as ONE batch: per-ray points (sdf_loss) + visible splats' samples (gs_sdf_loss),
                    # eikonal on the analytic gradient + align on both (neural_mapping.cpp:138-188, 420-462)
                    has = ids.numel() > 0
                    cl = lm.joint_sdf_loss_analytic(pool[i % 8], ray_sdf[i % 8], samples_cut if has else None, ids if has else None,
                                                    w_all if has else None, 0.02, 1.0, 1e-3, 0.1, 0.1)
                    with sdfm.grad_sinks_armed():
                        cl.backward()
                elif ids.numel() > 0:
                    cl = lm.gs_sdf_coupling(samples_cut, ids, w_all, 1e-3, 0.02, 0.1)
                    with sdfm.grad_sinks_armed():
                        cl.backward()
                gate.event = side.record_event() if side is not main else None     # d loss / d samples is complete
            stamp("samples leg issued")
        with (sdfm.grad_sinks_armed() if not args.no_sdf else contextlib.nullcontext()):
            if not args.no_sdf and samples_cut.grad is not None:
                if side is not main:
                    samples_cut.grad.record_stream(main)    # allocated on `side`, read by the projection backward on `main`
                torch.autograd.backward([loss, samples], [None, samples_cut.grad])
            else:
                loss.backward()
        # per-iteration train_callback -> NeuralGS::update_state (neural_mapping.cpp:486, neural_gaussian.cpp:626-680):
        # densification statistics from the compositing backward's `densify` gradient, one fused launch
        update_densify_state(gs_state, meta, N)
        stamp("backward issued")
        # view-parallel: the splat family's gradients are final here; the SDF leg's scatter keeps running on its own stream
        # beside the collective (no CU masks any more: the binned scatter is bandwidth-bound and shares the chip like any
        # other kernel; round 1 serialised the two because RCCL's workgroups queued behind the scatter's atomics)
        vp.all_reduce_group(params, args.collective)
        if update:
            adam.step(zero_grad=True)
            if ref_terms:   # train_callback -> prune_nan_gs's test (neural_gaussian.cpp:907-916), one launch, no host sync: the
                v_ = params.views   # count is read after the timed region (a real trainer reads it with the next step's sizes)
                nan_total.add_(ops.nan_rows(v_["offsets"], v_["scaling"], v_["quaternion"])[0])
        if not args.no_sdf:
            # the SDF network's gradients are final once the scatter stream has drained: all-reduce (61 MB table + MLP)
            # and optimizer step of this leg on its own stream, beside the splat leg
            with torch.cuda.stream(side):
                side.wait_stream(scatter)
                side.wait_stream(aux)
                vp.all_reduce_group(groups[0], args.collective)
                if update:
                    adam_sdf.step(zero_grad=True)
        stamp("optimizers issued")
        sizes.update(M=int(meta["gaussian_ids"].shape[0]), I=int(meta["flatten_ids"].shape[0]))
        for k in ("M", "I", "n_gs_sdf"):
            note(k, sizes.get(k, 0))

    vp.zero_grad()
    if args.dump_grads:
        step(0, update=False)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"splat": (ji.splat_flat_grad() if ji is not None else params.flat_grad).cpu(),
                        "sdf": [ji.sdf_flat_grad().cpu()] if ji is not None else [g.flat_grad.cpu() for g in groups], "sizes": dict(sizes)},
                       args.dump_grads)
        release_streams()
        if dist is not None:
            dist.destroy_process_group()
        return
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    # Steady-state warm-up, independent of --warmup: a fresh box starts at idle clocks with a cold caching allocator, and a
    # 5-step warm-up (50 ms) measures the ramp, not the step (round 2: the driver's 20-step run read 80 it/s where longer runs
    # read 100-110).  A FIXED number of untimed steps (GSDF_BENCH_WARM_STEPS, default 300: 1.2-1.5 s of GPU time; the compact line reports it
    # as internal_warmup_steps): the step time depends on the view (p10 / p90 of the step 3.6 / 4.7 ms over the 200 views), so the timed region
    # must be the SAME steps of the run every time — rounds 2-6 warmed up for 1.5 s of wall time, which ended after 290 or 300 steps by a hair and
    # moved a 20-step timed region by ten views (4.05 against 4.27 ms for the same code, round 6).
    extra_warm, warm_steps = 0, max(0, int(os.environ.get("GSDF_BENCH_WARM_STEPS", "300")))
    while extra_warm < warm_steps:
        for _ in range(min(10, warm_steps - extra_warm)):
            step(args.warmup + extra_warm)
            extra_warm += 1
        torch.cuda.synchronize()
    warm_total = args.warmup + extra_warm
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP-event timing of the roofline kernels over the timed region (the full per-operator table comes from a short
    # separate pass below: two events per launch on all ~25 operators cost ~2 % of the step in host time)
    step_marks = []

    import gs_sdf_amd.capi as capi
    roof_cabi = sorted(k for k, v in CABI_OPS.items() if v in ROOF)

    def timers_begin(only_roof):
        if impl == "cpp":
            capi.timing_begin(roof_cabi if only_roof else None)
        else:
            ops.TIMERS.enable(only=ROOF if only_roof else None)

    def timers_end():
        """-> (median, mean, calls) per operator"""
        if impl == "cpp":
            torch.cuda.synchronize()
            return cabi_timing_to_ops(capi.timing_end())
        r = ops.TIMERS.summary_ms("median"), ops.TIMERS.summary_ms("mean"), ops.TIMERS.calls()
        ops.TIMERS.disable()
        return r

    def timed(n_steps, first):
        hist.clear()
        timers_begin(True)
        if host is not None:
            host.clear()
        step_marks.clear()
        t0 = time.perf_counter()
        for i in range(n_steps):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(main)                      # the splat leg's stream: one mark per step where its first kernel is queued
            step_marks.append(ev)
            step(first + i)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(main)
        step_marks.append(ev)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        med_, mean_, calls_ = timers_end()
        return el, med_, mean_, calls_, {k: sum(v) / len(v) for k, v in hist.items()}

    elapsed, kern, kern_mean, calls, avg = timed(args.steps, warm_total)
    gaps = sorted(a.elapsed_time(b) for a, b in zip(step_marks[:-1], step_marks[1:]))
    step_dist = {"p10": gaps[len(gaps) // 10], "p50": gaps[len(gaps) // 2], "p90": gaps[(len(gaps) * 9) // 10], "max": gaps[-1],
                 "what": "ms between consecutive steps' first kernels on the splat leg's stream (HIP events), over the timed steps"}
    if host is not None and rank == 0:
        import collections
        acc, n = collections.OrderedDict(), 0
        for (ta, a), (tb, b) in zip(host[:-1], host[1:]):
            if tb != "begin":
                acc[tb] = acc.get(tb, 0.0) + (b - a)
            n += tb == "optimizers issued"
        print("host ms/step: " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items()), file=sys.stderr, flush=True)
    timers_begin(False)                       # every operator, outside the timed region
    nxt = warm_total + args.steps
    for i in range(min(10, args.steps)):
        step(nxt + i)
    nxt += min(10, args.steps)
    kern_all = timers_end()[0]
    if args.step_trace and impl == "cpp" and rank == 0:
        # a device timeline of the entry points without a profiler attached: begin / end of every C-ABI call over a few more steps
        torch.cuda.synchronize()
        capi.timing_begin(None)
        for i in range(6):
            step(nxt + i)
        nxt += 6
        torch.cuda.synchronize()
        with open(args.step_trace, "w") as f:
            for name, a, b in capi.timing_trace():
                f.write(f"{a * 1e3:10.1f} {b * 1e3:10.1f} {(b - a) * 1e3:8.1f} {name}\n")
    # The roofline kernels ALONE on the chip: the same step on ONE stream (a second JointIteration, same scene and views), HIP events on that
    # stream.  In the two-stream step a launch's duration includes the slowdown from the other leg's kernels running beside it (this round the
    # hash-grid forward runs beside the compositing backward: 2.4 ms there, 1.7 ms alone, the step equally fast) — `roofline.frac` keeps the
    # duration of the timed region, `roofline.alone` is the kernel's own rate and what the one-stream rocprofv3 summary in profiles/ must agree with.
    alone = None
    if impl == "cpp" and rank == 0 and world == 1 and not args.no_overlap and not args.no_sdf and not args.dump_grads:
        import copy
        a1 = copy.copy(args)
        a1.no_overlap = True
        ji1, pool1, rsdf1, cams1, _ = make_cpp_iteration(a1, sc, params, dev, W, H, deg, views)
        h1 = {}

        def step1(i):
            vi = i % views.shape[0]
            sz = ji1.step(views[vi][None], K, target, pool1[i % 8], rsdf1[i % 8], cpp_up, True, cpp_cams[vi])
            for k in ("M", "I", "n_gs_sdf"):
                h1.setdefault(k, []).append(int(sz[k]))
        for i in range(3):
            step1(warm_total + i)
        torch.cuda.synchronize()
        h1.clear()
        n1 = min(10, args.steps)
        capi.timing_begin(roof_cabi)
        for i in range(n1):
            step1(warm_total + i)
        torch.cuda.synchronize()
        med1, mean1, calls1 = cabi_timing_to_ops(capi.timing_end())
        alone = {"steps": n1, "mean_ms": mean1, "calls": calls1, "avg": {k: sum(v) / len(v) for k, v in h1.items()}}
        alone["avg"]["n_ray_pts"] = 32768.0
        del ji1
    replica_checksums = None
    if dist is not None:
        flat_s = ji.splat_flat() if ji is not None else params.flat
        flat_d = ji.sdf_flat() if ji is not None else (groups[0].flat if groups else flat_s[:0])
        cs = torch.stack([flat_s.double().sum(), flat_s.double().abs().sum(), flat_d.double().sum(), flat_d.double().abs().sum()]).to(dev)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)
        rows = [[float(v) for v in t.cpu()] for t in allcs]
        replica_checksums = {"what": "per rank: sum and sum |.| of the splat and of the SDF parameters (fp64) after the timed region",
                             "per_rank": rows, "all_equal": all(r == rows[0] for r in rows)}
    if rank == 0 and os.environ.get("GSDF_BENCH_DUMP_PARAMS"):
        # debugging / evidence hook (tools/compare_mlp_pipes.py): the parameters after warmup + steps optimizer steps
        torch.cuda.synchronize()
        torch.save({"splat": params.flat.detach().cpu(), "sdf": [g.flat.detach().cpu() for g in groups]}, os.environ["GSDF_BENCH_DUMP_PARAMS"])
    if rank == 0:
        from benchlib import report
        analytic_cfg = analytic and not args.no_sdf
        a = report.algorithmic(avg, N, W, H, deg, analytic, args.no_sdf, (dec_dims if impl == "cpp" else lm.decoder.dims) if not args.no_sdf else None)
        split_mlp = not args.no_sdf and os.environ.get("GSDF_MLP_MFMA", "bf16x3")[:1] not in "fF"
        roof = report.roofline(a, calls, kern_mean, kern, args.steps, args.workload, analytic, args.no_sdf, split_mlp, elapsed / args.steps)
        if alone is not None:
            # The roofline figure is the kernel's OWN rate: algorithmic bytes of a launch over its duration alone on the chip (the same step on one
            # stream, HIP events on that stream) — what the one-stream rocprofv3 summary in profiles/ reproduces.  `in_step` keeps the timed region's
            # figure: in the two-stream step a launch's duration includes the slowdown from the other leg's kernels beside it (the hash-grid forward
            # runs beside the compositing backward: 2.4 ms there, 1.7 alone, the step equally fast), which says where a leg waits, not how fast a kernel is.
            a_al = report.algorithmic(alone["avg"], N, W, H, deg, analytic, args.no_sdf, dec_dims)
            r_al = report.roofline(a_al, alone["calls"], alone["mean_ms"], alone["mean_ms"], alone["steps"], args.workload, analytic, args.no_sdf, split_mlp,
                                   elapsed / args.steps)
            pick = lambda r: {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "algorithmic_bytes", "algorithmic_flops") if k in r}
            dom = roof["kernel"]
            in_step = dict(pick(roof), sdf_points_per_step=round(a["sdf_pts"]), ms_per_step_by_kernel=roof["ms_per_step_by_kernel"],
                           others={k: pick(v) for k, v in roof["others"].items()}, valu=roof.get("valu"),
                           what="the timed region (two streams): HIP events on the launch stream, durations include the other leg's kernels running beside")
            src = r_al if r_al["kernel"] == dom else dict(r_al["others"][dom], kernel=dom)
            roof = dict(r_al, **pick(src), kernel=dom, kernel_selection=roof["kernel_selection"], launches_per_step=roof["launches_per_step"],
                        measured=("alone: the same step on ONE stream (a second JointIteration on the same scene and views, pool ray batches), HIP events on that "
                                  f"stream over {alone['steps']} steps after the timed region; profiles/ holds the one-stream rocprofv3 summary it agrees with"),
                        sdf_points_per_step=round(a_al["sdf_pts"]), in_step=in_step,
                        step_B_splat_bytes=int(a["b_splat"]), step_hbm_frac=a["b_splat"] / (elapsed / args.steps) / 8e12)
        direct = impl == "cpp" and not args.no_overlap and analytic and ref_terms
        detail = {
            "metric": "train iters/sec (splat raster + SDF fwd+bwd), 1M Gaussians @1080p" if not args.no_sdf
                      else "train iters/sec (splat raster fwd+bwd only), 1M Gaussians @1080p",
            "value": args.steps * world / elapsed, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "internal_warmup_steps": extra_warm, "step_ms_hip_events": step_dist,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {N} random Gaussians, {W}x{H}, sh_degree {deg}, 1 view/GPU/step",
                "M": round(a["M"]), "I": round(a["I"]), "L": round(a["I"] / a["T"]), "sdf_points_per_step": round(a["sdf_pts"]),
                "sdf_points": None if args.no_sdf else f"7 x ({a['n_ray']:.0f} ray + {a['n_gs']:.0f} splat samples), hash grid 2^19 x 16 x 2, fused 64-wide decoder",
                "sdf_config": None if args.no_sdf else (args.sdf_config + (" (deterministic mode)" if args.deterministic else "")),
                "ray_batch_short": None if args.no_sdf else ("sampled in the step (a16)" if batcher is not None else "pool of 8 pre-generated batches"),
                "ray_batch": None if batcher is None else {
                    "mode": "the reference's per-iteration ray-batch construction inside the timed step, one step ahead on its own stream (benchlib/raybatch.py; "
                            "neural_mapping.cpp:138-164, 73-104, 324-330; local_map.cpp:449-509)",
                    "rays_per_step": sum(x for x, _ in batcher.hist[-args.steps:]) / max(1, len(batcher.hist[-args.steps:])), "points_per_step": a["n_ray"]},
                "step_impl_short": ("C++ gsdf_extras::JointIteration, " + ("2 streams" if not args.no_overlap else "1 stream") + (", direct splat leg" if direct else "")
                                    if impl == "cpp" else "Python mirror, " + ("4 streams" if overlap else "1 stream")),
                "step_impl": ("C++/libtorch gsdf_extras::JointIteration (gs-sdf_amd/host/src/joint_step.cpp) over libgsdf_torch.so -> C ABI -> libgsdf_hip.so"
                              if impl == "cpp" else "Python mirror (gs_sdf_amd.ops / sdf over ctypes -> C ABI)"),
                "step": "reference joint iteration (neural_mapping.cpp:400-486), --sdf-config " + str(args.sdf_config) + ", --step-terms " + args.step_terms +
                        " (DESIGN.md section 6.1 lists the terms)",
                "parallelism": (f"view-parallel x{world}: backend {dist.get_backend()} ({'RCCL' if dist.get_backend() == 'nccl' else 'host-side'}), world_size "
                                f"{dist.get_world_size()}, one process per GPU, per-family gradient all-reduce on the owning leg's stream" if world > 1 else "single GPU"),
                "splat_order": args.splat_order,
                "sample_mode_short": args.sample_mode,
                "decoder_arithmetic": "bf16 MFMA, exact 3-term operand split, fp32 accumulate" if split_mlp else "fp32 MFMA"},
            "roofline": roof,
            "params_finite": (bool(torch.isfinite(ji.splat_flat()).all() and torch.isfinite(ji.sdf_flat()).all()) if ji is not None else
                              bool(torch.isfinite(params.flat).all()) and all(bool(torch.isfinite(g.flat).all()) for g in groups)),
            "nan_splats_seen_by_prune_test": int((ji.nan_splats_seen() if ji is not None else nan_total).item()),
            "hbm_gb": {"allocated_peak": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "reserved": round(torch.cuda.memory_reserved() / 2 ** 30, 2)},
            # every step this process ran (warm-up included): what one row of a rocprofv3 --stats summary of this command averages over
            "all_steps": {"steps": len(all_hist.get("M", [])), **{"mean_" + k: sum(v) / max(1, len(v)) for k, v in all_hist.items()}},
            # view-parallel replicas must stay identical: a checksum of every rank's parameters after the timed region (all equal <=> replicas identical)
            "replica_checksums": replica_checksums,
            "collectives": (None if dist is None or impl != "cpp" else {
                "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                **{fam: {"calls": len(ev), "bytes": ev[-1][2] if ev else 0,
                         "mean_ms": (sum(a.elapsed_time(b) for a, b, _ in ev[-args.steps:]) / max(1, len(ev[-args.steps:]))) if ev else None}
                   for fam, ev in coll_events.items()},
                "what": "per-family gradient all-reduce, HIP events on the stream it was issued on (splat: the caller's stream; sdf: JointIteration's second stream)"}),
            "kernel_ms": kern_all, "kernel_ms_note": "median launch duration per operator over 10 extra steps after the timed region",
        }
        detail["all_steps"]["mean_sdf_points"] = 7 * (detail["all_steps"].get("mean_n_ray_pts", 32768.0) + detail["all_steps"].get("mean_n_gs_sdf", 0.0))
        if world == 1 and not args.no_sdf and not args.no_secondary:
            from benchlib import secondary
            release_streams()
            detail["secondary"] = secondary.run_all(args, impl, analytic, sc, views, K, ug6, target, N, W, H, deg, dev)
        if world == 1 and not args.no_cpu_baseline:
            from benchlib.cpu_baseline import cpu_baseline
            detail["cpu_baseline"] = cpu_baseline(sc, views, params, N, W, H, deg, 0 if args.no_sdf else int(a["sdf_pts"]), dev)
        out_dir = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, os.environ.get("GSDF_BENCH_DETAIL", "bench_detail.json")), "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            print(f"[bench] could not write the detail file: {e}", file=sys.stderr, flush=True)
        print(report.compact_line(detail), flush=True)
    release_streams()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
