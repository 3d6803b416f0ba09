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
}


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
    ap.add_argument("--collective", default="all_reduce", choices=["all_reduce", "reduce_scatter_all_gather"],
                    help="N > 1: how a parameter family's flat gradient buffer is summed over the ranks")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked as `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU
        sys.exit(spawn_ranks(args.gpus))
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
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica)
    views = synth.make_views(200, seed=1).to(dev)
    K = sc["K"].to(dev)
    from gs_sdf_amd.trainer import morton_order
    order = morton_order(sc["means"]) if args.splat_order == "morton" else None
    params = SplatParams.from_scene(sc, dev, order)
    ug = {k: v.to(dev) for k, v in synth.upstream_grads(H, W, seed=2).items()}
    ug6 = {k: (1e-6 * v).contiguous() for k, v in ug.items()}       # the 1e-6 N(0,1) op-level upstream gradients, fixed
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)    # target image U(0,1) seed 3
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
        if dist is not None:      # view-parallel: one collective per parameter family, on the stream its optimizer runs on
            def _mean_over_ranks(g):
                if backend == "nccl":
                    dist.all_reduce(g, op=dist.ReduceOp.AVG)     # RCCL averages inside the collective: no second pass over the buffer
                else:                                            # gloo (CPU-side test runs) has no AVG
                    dist.all_reduce(g)
                    g.mul_(1.0 / world)
            ji.set_grad_hooks(_mean_over_ranks, _mean_over_ranks)
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

    sizes, hist = {}, {}

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
            hist.setdefault("n_ray_pts", []).append(int(rp.shape[0]))
            for k in ("M", "I", "n_gs_sdf"):
                hist.setdefault(k, []).append(int(sz[k]))
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
            hist.setdefault(k, []).append(sizes.get(k, 0))

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
    # read 100-110).  Untimed steps continue until the GPU has been busy for GSDF_BENCH_MIN_WARM_S seconds (default 1.5 s, at
    # most 300 steps); every rank takes the same number (the count is agreed through the slowest rank).
    min_warm_s, extra_warm = float(os.environ.get("GSDF_BENCH_MIN_WARM_S", "1.5")), 0
    t_w = time.perf_counter()
    while extra_warm < 300:
        go = torch.tensor([1.0 if time.perf_counter() - t_w < min_warm_s else 0.0], device=dev)
        if dist is not None:
            dist.all_reduce(go, op=dist.ReduceOp.MAX)
        if float(go.item()) == 0.0:
            break
        for _ in range(10):
            step(args.warmup + extra_warm)
            extra_warm += 1
        torch.cuda.synchronize()
    warm_total = args.warmup + extra_warm
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP-event timing of the roofline kernels over the timed region (the full per-operator table comes from a short
    # separate pass below: two events per launch on all ~25 operators cost ~2 % of the step in host time)
    ROOF = {"hashgrid_bwd", "hashgrid_fwd", "rasterize_2dgs_fwd", "rasterize_2dgs_bwd", "mlp_fwd", "mlp_bwd", "mlp_bwd_data", "mlp_bwd_weights",
            "mlp_bwd_bwd"}

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
    if rank == 0 and os.environ.get("GSDF_BENCH_DUMP_PARAMS"):
        # debugging / evidence hook (tools/compare_mlp_pipes.py): the parameters after warmup + steps optimizer steps
        torch.cuda.synchronize()
        torch.save({"splat": params.flat.detach().cpu(), "sdf": [g.flat.detach().cpu() for g in groups]}, os.environ["GSDF_BENCH_DUMP_PARAMS"])
    if rank == 0:
        M, I, n_gs = avg["M"], avg["I"], avg.get("n_gs_sdf", 0.0)        # means over the timed steps (views differ per step)
        P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        Kb = (deg + 1) ** 2
        n_ray = avg.get("n_ray_pts", 32768.0)                              # the per-ray batch of the step (sampled: ~32768 by the throttle)
        base_pts = 0 if args.no_sdf else n_ray + n_gs                      # points that carry gradients (ray batch + splat samples)
        sdf_pts = 7 * base_pts                                            # + their 6 central-difference points (forward-only when analytic)
        # algorithmic bytes / flops per STEP of each operator, divided by its launches per step
        # below.  The dominant kernel is the one with the largest total time per step.
        alg_step = {"rasterize_2dgs_bwd": 80 * I + 48 * P + 88 * M, "rasterize_2dgs_fwd": 80 * I + 48 * P}
        flops_step = {}
        if not args.no_sdf:
            dd = dec_dims if impl == "cpp" else lm.decoder.dims
            macs = sum(a_ * b_ for a_, b_ in zip(dd[:-1], dd[1:]))                             # multiply-adds per point and pass
            # S1 per query point: fwd 12 + 1024 (16 levels x 8 corners x 8 B) + 128 (+ 384 B of Jacobian per gradient-carrying
            # point); bwd 8 + 128 + 1024 scatter (+ 128 + 12 for the second-order operands of the analytic configuration)
            alg_step["hashgrid_fwd"] = 1164 * sdf_pts + 384 * base_pts
            alg_step["hashgrid_bwd"] = (1300 * base_pts) if analytic else (1160 * sdf_pts)
            bwd_pts = base_pts if analytic else sdf_pts
            flops_step = {"mlp_fwd": 2 * macs * sdf_pts, "mlp_bwd": 2 * 2 * macs * bwd_pts,         # one-pass: data + weights
                          "mlp_bwd_data": 2 * macs * bwd_pts, "mlp_bwd_weights": 2 * macs * bwd_pts,
                          "mlp_bwd_bwd": 2 * 2 * macs * base_pts}                                   # masked forward + weight GEMM
        launches = lambda k: max(1.0, calls.get(k, 0) / args.steps)
        alg = {k: v / launches(k) for k, v in alg_step.items() if calls.get(k)}
        flops = {k: v / launches(k) for k, v in flops_step.items() if calls.get(k)}
        # time per step of an operator = MEAN launch x launches per step (the launches of an SDF operator differ 10x in size: the
        # 32768-ray batch against ~0.45 M splat samples; `alg` is the per-launch mean to match); medians are reported too
        per_step = {k: kern_mean.get(k, 0.0) * calls.get(k, 0) / args.steps for k in list(alg) + list(flops)}
        dom = max(per_step, key=lambda k: per_step[k])
        dom_rule = "largest time per step by the in-bench HIP-event timers"
        # Dominant kernel.  The live timers bracket whole ENTRY POINTS on their stream: beside the other leg they also count the time a launch
        # waits for CUs, and an operator made of several kernels (the compositing backward = memsets + kernel + unpack) is their sum.  GPU time
        # proper is what rocprofv3 --kernel-trace --stats of this same command reports, and the contract asks for that summary to be committed:
        # when THIS ROUND's summary is in profiles/ (PROFILE_ROUND below — a file of an earlier round is never consulted, the kernels have
        # changed since), the operator it ranks first is the dominant one; the live ranking is printed next to it, and without the file the live
        # ranking decides.  Either way the kernel is timed live, here.
        PROFILE_ROUND = "r04"
        live_first = dom
        rocprof_first = None
        # Two summaries are committed: the default (two-stream) command's and the same work on ONE stream.  In the overlapped one a kernel's
        # duration includes the time its workgroups wait for CUs beside the other leg (its first row this round is l1_dssim_bwd_kernel, a
        # 0.11 ms kernel alone); the one-stream summary ranks by work, and that ranking is the one used.  Both first rows are reported.
        spath = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_bench_cfg3_serial_kernel_stats.csv")
        opath = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_bench_cfg3_kernel_stats.csv")
        if args.workload == "cfg3_1M_1080p" and analytic and not args.no_sdf and os.path.exists(spath):
            import csv
            kmap = (("hashgrid_fwd", "hashgrid_fwd"), ("raster_bwd_", "rasterize_2dgs_bwd"), ("raster_fwd_", "rasterize_2dgs_fwd"),
                    ("mlp_fwd_split_kernel<32, 512, false>", "mlp_fwd"), ("bin_apply", "hashgrid_bwd"), ("bin_emit", "hashgrid_bwd"))
            share = {}
            for row in list(csv.reader(open(spath)))[1:]:
                for sub, op in kmap:
                    if sub in row[0]:
                        share[op] = share.get(op, 0.0) + float(row[4])
                        break
            if share:
                top = max(share, key=lambda k: share[k])
                rocprof_first = {"operator": top, "percent_of_gpu_time": share[top], "file": os.path.relpath(spath, ROOT),
                                 "live_timers_rank_first": live_first, "agrees_with_live_timers": top == live_first}
                if os.path.exists(opath):
                    orow = list(csv.reader(open(opath)))[1]
                    rocprof_first["overlapped_summary_first_row"] = {"file": os.path.relpath(opath, ROOT), "kernel": orow[0][:60], "percent_of_gpu_time": float(orow[4]),
                                                                     "average_launch_us": float(orow[3]) / 1e3,
                                                                     "note": "duration beside the other leg = work + waiting for CUs"}
                if top in per_step:
                    dom, dom_rule = top, (f"first in rocprofv3 --kernel-trace --stats of this round's step on one stream ({os.path.relpath(spath, ROOT)}: "
                                          f"{share[top]:.1f} % of GPU time, each kernel alone on the chip); timed live here in the default two-stream step "
                                          f"(the live entry-point timers, which include CU waits, rank {live_first} first)")
        dur_ms = kern_mean.get(dom, float("nan"))

        split_mlp = os.environ.get("GSDF_MLP_MFMA", "bf16x3")[:1] not in "fF"

        def roof(k):
            if k in alg:
                a = alg[k] / (kern_mean[k] * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": a, "peak": 8000.0, "unit": "GB/s", "frac": a / 8000.0,
                        "algorithmic_bytes": int(alg[k])}
            a = flops[k] / (kern_mean[k] * 1e-3) / 1e12
            if split_mlp and k in ("mlp_fwd", "mlp_bwd"):
                # csrc/mlp_split.hip: fp32 operands as three exact bf16 terms, six partial products per multiply-add on
                # v_mfma_f32_32x32x16_bf16 -> the pipe executes 6x the algorithmic flops; priced against its dense bf16 peak
                return {"bound": "mfma", "achieved": 6 * a, "peak": 2500.0, "unit": "TFLOP/s", "frac": 6 * a / 2500.0,
                        "pipe": "bf16 MFMA, fp32-accurate 3-term operand split (6 products per multiply-add)",
                        "algorithmic_flops": int(flops[k]), "fp32_equivalent_tflops": a, "fp32_mfma_peak": 157.3}
            return {"bound": "mfma", "achieved": a, "peak": 157.3, "unit": "TFLOP/s", "frac": a / 157.3,
                    "algorithmic_flops": int(flops[k])}
        b_splat = (80 + 12 * Kb) * N + (364 + 12 * Kb) * M + 204 * I + 96 * P + 4 * T
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
        vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
        valu = json.load(open(vpath)).get(args.workload) if os.path.exists(vpath) else None
        terms = ("0.8 L1 + 0.2 D-SSIM on colour" +
                 ("; 0.01 x depth->normal consistency (neural_mapping.cpp:243-266); 0.05 x isotropic regulariser of the visible splats (:268-276)"
                  if ref_terms else "; 1e-6 N(0,1) op-level gradients on depth/alpha/normal/median (--step-terms round2)"))
        if not args.no_sdf and analytic:
            terms += ("; SDF configuration = the reference's DEFAULT (decoder_implementation 0: biased 5-layer decoder; numerical_grad 0): per-ray "
                      "batch (32768 points): sdf_loss + 0.1 eikonal on the ANALYTIC gradient (double backward through decoder and hash grid) "
                      "+ 0.1 align |analytic - numerical.detach()| (6 forward-only stencil evaluations per point); GS<->SDF: 1e-3 gs_sdf_loss "
                      "on the visible splats' samples (visibility > 0.1, occupancy-valid) + the same eikonal / align regularisers at "
                      "samples.detach() (neural_mapping.cpp:106-136, 420-462)")
        elif not args.no_sdf:
            terms += ("; SDF configuration = decoder_implementation 1 (bias-free FullyFusedMLP; the reference forces numerical_grad with it, "
                      "params.cpp:396-399): per-ray batch: sdf_loss + 0.1 eikonal on the numerical gradient (6-point stencil, all 7 rows "
                      "differentiated); GS<->SDF: 1e-3 gs_sdf_loss + 0.1 eikonal (numerical) at the visible splats' samples")
        terms += "; per-iteration update_state" + (" + prune_nan test" if ref_terms else "") + " (neural_gaussian.cpp:626-680, 907-916); fused Adam on all parameters"
        split_mlp_cfg = not args.no_sdf and os.environ.get("GSDF_MLP_MFMA", "bf16x3")[:1] not in "fF"
        out = {
            "metric": "train iters/sec (splat raster + SDF fwd+bwd), 1M Gaussians @1080p" if not args.no_sdf
                      else "train iters/sec (splat raster fwd+bwd only), 1M Gaussians @1080p",
            "value": args.steps * world / elapsed, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "internal_warmup_steps": extra_warm, "step_ms_hip_events": step_dist,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} random Gaussians, {W}x{H}, sh_degree {deg}, 1 view/GPU/step; means over the timed "
                                   f"steps: M={M:.0f} I={I:.0f} L={I / T:.0f}" + ("" if args.no_sdf else f"; hash-grid SDF (2^19 x16x2, fused "
                                   f"64-wide MLP) evaluated at {sdf_pts:.0f} points/step = 7 x ({n_ray:.0f} ray + {n_gs:.0f} splat samples)"),
                       "ray_batch": (None if args.no_sdf else
                                     ({"mode": "sampled inside the timed step, one step ahead on its own stream (bench.py: RayBatcher): random rays gathered from a "
                                               "host-side synthetic depth pack (2 M rays: camera centre -> splat centre), H2D copy, octree ray march (1 sample per "
                                               "occupied voxel) + 3 free + 3 near-surface samples + end point per ray, targets truncated at 3 leaves, in-range filter, "
                                               "throttled to ~32768 points per batch (neural_mapping.cpp:138-164, 73-104, 324-330; local_map.cpp:449-509)",
                                       "rays_per_step": sum(a for a, _ in batcher.hist[-args.steps:]) / max(1, len(batcher.hist[-args.steps:])),
                                       "points_per_step": n_ray} if batcher is not None else
                                      {"mode": "8 pre-generated batches of 32768 uniform points (--ray-batch pool)", "points_per_step": n_ray})),
                       "sdf_config": None if args.no_sdf else args.sdf_config,
                       "step_impl": ("C++/libtorch: gsdf_extras::JointIteration (gs-sdf_amd/host/src/joint_step.cpp) over libgsdf_torch.so -> C ABI -> "
                                     "libgsdf_hip.so; " + ("two HIP streams, the splat leg's operators called through the C ABI directly (step_direct: no autograd engine), the "
                                                           "SDF batch one autograd node" if (not args.no_overlap and analytic and ref_terms) else
                                                          ("two HIP streams" if not args.no_overlap else "one HIP stream") + ", autograd composition") + "; driven per step through "
                                     "the pybind harness" if impl == "cpp" else "Python mirror (gs_sdf_amd.ops / sdf over ctypes -> C ABI), "
                                     + ("four HIP streams" if overlap else "one HIP stream")),
                       "step": "reference joint iteration (neural_mapping.cpp:400-486): " + terms,
                       "parallelism": (f"view-parallel x{world}: torch.distributed backend {dist.get_backend()} ({'RCCL' if dist.get_backend() == 'nccl' else 'host-side'}), "
                                       f"world_size {dist.get_world_size()}, one process per GPU, gradient all-reduce (ReduceOp.{'AVG' if backend == 'nccl' else 'SUM, then 1/N'}) "
                                       "per parameter family on the stream its optimizer runs on" if world > 1 else "single GPU"),
                       "splat_order": "Morton order of the centres (trainer.morton_order)" if args.splat_order == "morton" else "as given (random)",
                       "sample_mode": ("center_reg = 1: SDF samples = splat centres, weight 1" if args.sample_mode == "center" else
                                       "stochastic (the reference's default, center_reg absent): one random point per visible splat's disc, "
                                       "weight exp(-|eps|^2/2)"),
                       "decoder_arithmetic": ("fp32 operands as 3 exact bf16 terms, 6 partial products per multiply-add on the bf16 MFMA pipe, "
                                              "fp32 accumulate: error against fp64 equal to the fp32 MFMA's (tools/ubench/mfma_split.hip)"
                                              if split_mlp_cfg else "fp32 MFMA")},
            "roofline": dict(roof(dom), kernel=dom, kernel_selection=dom_rule, committed_rocprof_ranking=rocprof_first, traffic=traffic, avg_launch_ms=dur_ms, median_launch_ms=kern.get(dom),
                             # what the kernel actually moves (PMC FETCH_SIZE + WRITE_SIZE of a single-stream run, profiles/) over its
                             # launch time measured here: how hard it drives HBM, next to `frac` (= algorithmic bytes only).  The binned
                             # scatter trades 2.6x more, fully coalesced, bytes for not using the 21 G/s fp32 atomic units; in the
                             # overlapped step its launches share HBM with the other leg's kernels
                             traffic_GBps=(None if not traffic or not dur_ms else traffic / (dur_ms * 1e-3) / 1e9),
                             traffic_frac_of_hbm_peak=(None if not traffic or not dur_ms else traffic / (dur_ms * 1e-3) / 8e12),
                             launches_per_step=calls.get(dom, 0) / args.steps,
                             timing=("HIP events on the launch stream over the timed steps (" + ("gsdf_timing_begin/_end inside the C ABI: one event "
                                     "pair around everything an entry point launches" if impl == "cpp" else "ops.TIMERS") + "); mean launch "
                                     "(operators with unequal launches); kernels of the two legs share the chip, so a launch's duration includes "
                                     "the slowdown from its neighbours — profiles/ holds the one-stream rocprofv3 stats"),
                             ms_per_step_by_kernel={k: round(v, 4) for k, v in per_step.items()},
                             # the same figure for the other large kernels
                             others={k: dict(roof(k), avg_launch_ms=kern_mean[k], median_launch_ms=kern[k]) for k in per_step if k != dom and kern.get(k)},
                             # compositing kernels: the VALU-issue side (they are bound by instruction issue, not by HBM): wave64
                             # VALU instructions per launch (profiles/valu_insts.json: rocprofv3 --pmc SQ_INSTS_VALU, single stream)
                             # over the launch time measured here, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
                             # (nominal), and against the 759 G/s independent v_fma_f32 chains reach on this chip
                             # (tools/ubench/pk_fma.hip; v_pk_fma_f32 gives only 1.07-1.14x more element-FMAs: no packed lever)
                             valu=(None if not valu else {k: {"insts_per_launch": v, "issue_peak_G_per_s": 614.4,
                                                              "frac_of_issue_peak": v / (kern_mean[k] * 1e-3) / 614.4e9,
                                                              "measured_fma_issue_G_per_s": 759.0,
                                                              "frac_of_measured_fma_issue": v / (kern_mean[k] * 1e-3) / 759.0e9}
                                                          for k, v in valu.items() if kern_mean.get(k)}),
                             step_B_splat_bytes=int(b_splat), step_hbm_frac=b_splat / (elapsed / args.steps) / 8e12),
            "params_finite": (bool(torch.isfinite(ji.splat_flat()).all() and torch.isfinite(ji.sdf_flat()).all()) if ji is not None else
                              bool(torch.isfinite(params.flat).all()) and all(bool(torch.isfinite(g.flat).all()) for g in groups)),
            "nan_splats_seen_by_prune_test": int((ji.nan_splats_seen() if ji is not None else nan_total).item()),
            "hbm_gb": {"allocated_peak": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                       "reserved": round(torch.cuda.memory_reserved() / 2 ** 30, 2)},
            "kernel_ms": kern_all, "kernel_ms_note": "median launch duration per operator over 10 extra steps after the timed region",
        }
        if world == 1 and not args.no_sdf and not args.no_secondary:
            # (1) the OTHER SDF configuration, same step otherwise: a clearly named line of its own (own roofline), run as a
            #     subprocess of this script so that nothing of this run's allocator / stream state leaks into it
            import subprocess
            other = "tcnn" if analytic else "default"
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(min(args.steps, 40)), "--warmup", str(args.warmup),
                       "--workload", args.workload, "--sdf-config", other, "--step-terms", args.step_terms, "--splat-order", args.splat_order,
                       "--sample-mode", args.sample_mode, "--no-secondary", "--no-cpu-baseline"] + (["--no-overlap"] if args.no_overlap else [])
                release_streams()
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["other_sdf_config"] = {"sdf_config": other, "value": j["value"], "unit": "iters/s", "ms_per_step": j["ms_per_step"],
                                           "steps": j["steps"], "step_ms_hip_events": j["step_ms_hip_events"], "step": j["config"]["step"],
                                           "workload": j["config"]["workload"], "roofline": j["roofline"], "kernel_ms": j["kernel_ms"]}
            except Exception as e:      # never let an extra line take the headline down
                out["other_sdf_config"] = {"sdf_config": other, "error": repr(e)[:300]}
            # (2) the joint iteration as neural_mapping_node would run it linked against the drop-in submodules with zero source
            #     edits (eager losses / Adam / numerical get_gradient around the drop-in operators; tcnn configuration)
            try:
                out["reference_loop_zero_edits"] = reference_loop(args, sc, views, K, ug6, target, N, W, H, deg, dev)
            except Exception as e:
                out["reference_loop_zero_edits"] = {"error": repr(e)[:300]}
            # (3) the other host implementation of the same step (Python mirror when the headline is the C++ step, and vice versa)
            try:
                other_impl = "python" if impl == "cpp" else "cpp"
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(min(args.steps, 40)), "--warmup", str(args.warmup),
                       "--workload", args.workload, "--sdf-config", args.sdf_config, "--step-terms", args.step_terms, "--splat-order", args.splat_order,
                       "--step-impl", other_impl, "--sample-mode", args.sample_mode, "--no-secondary", "--no-cpu-baseline"] + (["--no-overlap"] if args.no_overlap else [])
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["python_mirror_step" if other_impl == "python" else "cpp_joint_iteration"] = {
                    "value": j["value"], "unit": "iters/s", "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                    "step_ms_hip_events": j["step_ms_hip_events"], "step_impl": j["config"]["step_impl"]}
            except Exception as e:
                out["python_mirror_step" if impl == "cpp" else "cpp_joint_iteration"] = {"error": repr(e)[:300]}
            # (4) the other SDF-sample mode (the reference's default draws one stochastic point per visible splat; the headline uses the
            #     fully specified center_reg = 1 mode of the parity runs)
            try:
                other_mode = "stochastic" if args.sample_mode == "center" else "center"
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(min(args.steps, 40)), "--warmup", str(args.warmup),
                       "--workload", args.workload, "--sdf-config", args.sdf_config, "--step-terms", args.step_terms, "--splat-order", args.splat_order,
                       "--step-impl", impl, "--sample-mode", other_mode, "--no-secondary", "--no-cpu-baseline"] + (["--no-overlap"] if args.no_overlap else [])
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                out["other_sample_mode"] = {"sample_mode": j["config"]["sample_mode"], "value": j["value"], "unit": "iters/s",
                                            "ms_per_step": j["ms_per_step"], "steps": j["steps"], "step_ms_hip_events": j["step_ms_hip_events"]}
            except Exception as e:
                out["other_sample_mode"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, views, params, N, W, H, deg, 0 if args.no_sdf else int(sdf_pts), dev)
        print(json.dumps(out), flush=True)
    release_streams()
    if dist is not None:
        dist.destroy_process_group()


# C-ABI entry points -> operator names of the roofline / kernel_ms tables (gsdf_timing_begin / _end time whole entry points)
CABI_OPS = {"gsdf_hashgrid_fwd": "hashgrid_fwd", "gsdf_hashgrid_fwd_stencil": "hashgrid_fwd", "gsdf_hashgrid_fwd_jac_rows": "hashgrid_fwd",
            "gsdf_hashgrid_fwd_jac": "hashgrid_fwd", "gsdf_hashgrid_bwd_binned2": "hashgrid_bwd", "gsdf_hashgrid_bwd_binned_stencil": "hashgrid_bwd",
            "gsdf_hashgrid_bwd": "hashgrid_bwd", "gsdf_hashgrid_bwd_jac": "hashgrid_bwd_input", "gsdf_hashgrid_bwd_bwd": "hashgrid_bwd_bwd",
            "gsdf_mlp_fwd": "mlp_fwd", "gsdf_mlp_bwd": "mlp_bwd", "gsdf_mlp_bwd_data": "mlp_bwd_data", "gsdf_mlp_bwd_weights": "mlp_bwd_weights",
            "gsdf_mlp_bwd_bwd": "mlp_bwd_bwd", "gsdf_rasterize_2dgs_fwd": "rasterize_2dgs_fwd", "gsdf_rasterize_2dgs_bwd": "rasterize_2dgs_bwd"}


def cabi_timing_to_ops(rep):
    """gs_sdf_amd.capi.timing_end() report -> (median, mean, calls) per operator name (entry points of one operator merged)."""
    med, mean, calls, tot = {}, {}, {}, {}
    for name, r in rep.items():
        op = CABI_OPS.get(name, name[5:] if name.startswith("gsdf_") else name)
        calls[op] = calls.get(op, 0) + r["calls"]
        tot[op] = tot.get(op, 0.0) + r["total_ms"]
        med[op] = max(med.get(op, 0.0), r["median_ms"])        # merged entry points: the larger launch's median
    for op in calls:
        mean[op] = tot[op] / max(1, calls[op])
    return med, mean, calls


class RayBatcher:
    """The reference's per-iteration SDF ray batch, built INSIDE the timed step:
      NeuralSLAM::sdf_train_batch_iter (neural_mapping.cpp:138-164): k_batch_num random indices into the HOST-side depth pack
        (train_depth_pack_ lives on the CPU, :145-156), gather, copy to the device;
      NeuralSLAM::sample (:73-104) = gsdf_model::sample_rays (host/src/local_map.cpp): LocalMap::sample — octree ray march, one sample per
        occupied voxel crossed (gsdf occ_raymarch kernels), + free_sample_num stratified samples, those in front of the surface kept
        (local_map.cpp:449-509) — + surface_sample_num samples at depth - N(0, sample_std), targets truncated at +-truncated_dis, + the ray end
        points, in-range filter;
      the throttle of the training loop (:324-330): k_batch_num = min(batch_pt_num / EMA(points per ray), batch_pt_num), so that a batch
        holds ~batch_pt_num = 32768 points.
    The batch depends on the occupancy structure and the rays only, never on the parameters: it is issued ONE STEP AHEAD on a stream of its
    own (a data-loader prefetch), so its size read-backs (nonzero) wait for its own small kernels, not for the training step in flight.
    Synthetic depth pack: rays from the 200 camera centres to splat centres (every ray ends in an occupied leaf), 10000 per view."""

    def __init__(self, host, sc, views, dev, batch_pt_num=32768, rays_per_view=10000, leaf=0.0625, map_size=16.0, seed=7):
        self.host, self.dev, self.batch_pt_num = host, dev, batch_pt_num
        cfg = host.MapConfig()
        cfg.leaf_size, cfg.inner_map_size = leaf, map_size - 2 * leaf
        self.lm = host.LocalMap(torch.tensor([0.0, 0.0, 5.5]), cfg)
        self.lm.update_octree_as(sc["means"].to(dev), False)
        g = torch.Generator().manual_seed(7)            # the depth pack is the data set: the same on every rank; `seed` drives the draws
        c2w = torch.linalg.inv(views.cpu().double())
        centres = c2w[:, :3, 3].float()                                               # camera centres in the world
        V, N = centres.shape[0], sc["means"].shape[0]
        idx = torch.randint(0, N, (V, rays_per_view), generator=g)
        end = sc["means"][idx.reshape(-1)]
        org = centres[:, None, :].expand(V, rays_per_view, 3).reshape(-1, 3)
        d = end - org
        depth = d.norm(dim=1, keepdim=True)
        pin = lambda t: t.contiguous().pin_memory()
        self.pack = dict(origin=pin(org), direction=pin(d / depth), depth=pin(depth), xyz=pin(end))   # the host-side depth pack
        self.n_rays = org.shape[0]
        self.k_batch_num, self.pts_per_ray = batch_pt_num, 1.0                        # nsdf_train: k_batch_num = k_batch_ray_num (= batch_pt_num)
        self.stream = torch.cuda.Stream(device=dev)
        self.gen = torch.Generator().manual_seed(seed + 1)
        self.sample_std, self.truncated_dis = 0.02, 3 * leaf                          # base.yaml: sample_std; truncated at 3 leaves
        self.ready = None
        self.hist = []
        # the throttle starts from its steady state (the reference reaches it after ~50 iterations of :324-330; a batch of 32768 RAYS in this
        # scene would be 3.6 M points): a calibration batch of 256 rays measures the points per ray
        self.k_batch_num = 256
        self.issue()
        n0, p0 = self.hist[-1]
        self.pts_per_ray = max(p0 / max(n0, 1), 1e-3)
        self.k_batch_num = max(1, min(int(self.batch_pt_num / self.pts_per_ray), self.batch_pt_num))
        self.ready, self.hist = None, []

    def issue(self):
        """queues the next batch on the prefetch stream -> nothing; `take()` hands it to the step"""
        n = int(self.k_batch_num)
        indices = (torch.rand(n, generator=self.gen) * self.n_rays).long().clamp_(0, self.n_rays - 1)      # :141-149
        with torch.cuda.stream(self.stream):
            rays = {k: v.index_select(0, indices).to(self.dev, non_blocking=True) for k, v in self.pack.items()}   # :151-156
            b = self.host.sample_rays(self.lm, rays, self.sample_std, self.truncated_dis, 3, True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        pt_n = int(b["xyz"].shape[0])
        self.pts_per_ray = self.pts_per_ray * 0.9 + (pt_n / max(n, 1)) * 0.1           # :324-327
        self.k_batch_num = max(1, min(int(self.batch_pt_num / self.pts_per_ray), self.batch_pt_num))
        self.hist.append((n, pt_n))
        self.ready = (b["xyz"].contiguous(), b["ray_sdf"].contiguous(), ev)

    def take(self, consumer_stream):
        if self.ready is None:
            self.issue()
        xyz, rsdf, ev = self.ready
        consumer_stream.wait_event(ev)
        xyz.record_stream(consumer_stream); rsdf.record_stream(consumer_stream)
        self.ready = None
        return xyz, rsdf


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
                            args.sample_mode == "center")   # level 8: 1/16 m leaves in 16 m
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
    t_w, i = time.perf_counter(), args.warmup
    while time.perf_counter() - t_w < float(os.environ.get("GSDF_BENCH_MIN_WARM_S", "1.5")) and i < args.warmup + 300:     # steady state, as the Python step
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
    cores = os.cpu_count() or 1
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
    # The compositing gradients' first-order error bounds (g64["cond"], eps32 units) pushed through the LINEAR projection / SH backward, one
    # upstream component at a time, so that every term enters with its absolute value: bound(leaf) = sum_k |J^T (e_k . bound_k)|.
    # The end-to-end parameter gradients below are then gated like the compositing's own: 1e-4 max(|ref|, mean|ref|) + COND_C eps32 bound.
    cnd = g64["cond"]
    zM = lambda *sh: np.zeros((M,) + sh, np.float64)
    b_means, b_quats, b_scales = np.zeros((N, 3)), np.zeros((N, 4)), np.zeros((N, 3))
    for k in range(14):
        v2d, vrt, vnr = zM(2), zM(3, 3), zM(3)
        if k < 2:
            v2d[:, k] = cnd[:, k]
        elif k < 11:
            vrt.reshape(M, 9)[:, k - 2] = cnd[:, k]
        else:
            vnr[:, k - 11] = cnd[:, 15 + k - 11]
        bm_, bq_, bs_ = orc.projection_2dgs_bwd(means, quats, scales, n(view), n(sc["K"]), W, H, p["camera_ids"], p["gaussian_ids"], v2d, np.zeros(M, np.float64),
                                                vrt, vnr, prec="f64")
        b_means += np.abs(bm_); b_quats += np.abs(bq_); b_scales += np.abs(bs_)
    b_sh = np.zeros(n(sc["sh"]).shape)
    for k in range(3):
        vc = zM(3)
        vc[:, k] = cnd[:, 11 + k]
        bsh_, bms_ = orc.view_colors_bwd(n(view), means, n(sc["sh"]), p["camera_ids"], p["gaussian_ids"], deg, vc, prec="f64")
        b_sh += np.abs(bsh_); b_means += np.abs(bms_)
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
            "sample": f"splat half of 1 iteration in full (oracle/splat_oracle.c f32 build, OpenMP over tiles on {cores} threads for "
                      f"compositing, projection/sort single-threaded): {t_splat:.1f} s" +
                      (f"; SDF half (oracle/sdf_oracle.c fwd+bwd) on {n_s} of the step's {n_sdf_points} query points, scaled linearly: "
                       f"{t_sdf:.1f} s" if n_sdf_points else ""),
            "parity": par}


if __name__ == "__main__":
    main()
