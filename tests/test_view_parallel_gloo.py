"""N>1 path on CPU: world_size-2 gloo run of the view-parallel gradient synchronisation
(gs_sdf_amd.trainer.ViewParallel).  The HIP operators need a GPU, so each rank's per-view loss is a
stand-in differentiable function of the same flat parameter buffer; the property under test is the
distributed one: after all_reduce_grads() every rank holds the MEAN over views of the per-view grads,
identical to a single process that evaluates both views."""
import os
import socket

import torch
import torch.multiprocessing as mp

import gs_sdf_amd.synth as synth
from gs_sdf_amd.trainer import SplatParams, ViewParallel


def _loss(params, view):
    xyz, quat, scales, opacity, sh = params.activated()
    cam = xyz @ view[:3, :3].T + view[:3, 3]
    return ((cam[:, :2] / cam[:, 2:3]).square().sum() + (scales * opacity[:, None]).sum() + (sh.sum((1, 2)) * cam[:, 2]).sum()
            + (quat / quat.norm(dim=-1, keepdim=True)).sum())


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(500, 64, 48, sh_degree=1, seed=0)
    views = synth.make_views(4, seed=1)
    params = SplatParams.from_scene(sc, torch.device("cpu"))
    vp = ViewParallel(params, dist)
    for step in range(2):
        vp.zero_grad()
        _loss(params, views[(step * world + rank) % 4]).backward()
        if step == 0:
            vp.all_reduce_grads()                       # blocking form
        else:
            vp.all_reduce_group_async(params)           # overlapped form used by bench.py
            vp.finish()
    out[rank] = params.flat_grad.clone()
    dist.destroy_process_group()


def test_all_reduce_matches_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    sc = synth.make_scene(500, 64, 48, sh_degree=1, seed=0)
    views = synth.make_views(4, seed=1)
    params = SplatParams.from_scene(sc, torch.device("cpu"))
    vp = ViewParallel(params, None)
    vp.zero_grad()
    for r in range(2):                      # step 1 of the 2-rank job used views 2 and 3
        (_loss(params, views[2 + r]) / 2).backward()
    assert torch.allclose(out[0], out[1])
    assert torch.allclose(out[0], params.flat_grad, rtol=1e-5, atol=1e-6)
    # grads are views of ONE flat buffer: a single collective covers every parameter
    assert sum(p.numel() for p in params.parameters()) == params.flat_grad.numel()
    for p in params.parameters():
        assert p.grad.data_ptr() >= params.flat_grad.data_ptr()


def _refine_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)                 # every rank saw different views
    n = 1000
    state = dict(grad2d=torch.rand(n, generator=g), count=torch.randint(0, 5, (n,), generator=g).float(),
                 vis=torch.rand(n, generator=g), radii=torch.rand(n, generator=g))
    before = {k: v.clone() for k, v in state.items()}
    ViewParallel(None, dist).sync_refine_state(state)
    out[rank] = (before, {k: v.clone() for k, v in state.items()})
    dist.destroy_process_group()


def test_refine_statistics_are_merged_identically_on_every_rank():
    """SURVEY 8e, second collective: grad2d / count summed, vis / radii max-merged, so that the densification masks
    (grad2d / count > threshold, vis < threshold) come out identical on all ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_refine_worker, args=(2, port, out), nprocs=2, join=True)
    (b0, a0), (b1, a1) = out[0], out[1]
    for k in ("grad2d", "count"):
        assert torch.equal(a0[k], a1[k]) and torch.allclose(a0[k], b0[k] + b1[k])
    for k in ("vis", "radii"):
        assert torch.equal(a0[k], a1[k]) and torch.equal(a0[k], torch.maximum(b0[k], b1[k]))
    grow0 = (a0["grad2d"] / a0["count"].clamp_min(1)) > 0.2
    grow1 = (a1["grad2d"] / a1["count"].clamp_min(1)) > 0.2
    assert torch.equal(grow0, grow1) and 0 < int(grow0.sum()) < grow0.numel()


def _collective_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(501, 64, 48, sh_degree=1, seed=0)            # 501 rows: the flat buffer is not a multiple of 2
    views = synth.make_views(4, seed=1)
    res = {}
    for mode in ("all_reduce", "reduce_scatter_all_gather"):
        params = SplatParams.from_scene(sc, torch.device("cpu"))
        vp = ViewParallel(params, dist)
        vp.zero_grad()
        _loss(params, views[rank]).backward()
        g = torch.Generator().manual_seed(50 + rank)
        stats = [torch.rand(501, generator=g), torch.randint(0, 4, (501,), generator=g).float()]
        before = [t.clone() for t in stats]
        vp.all_reduce_group(params, mode, extra_sum=stats)              # the refine statistics ride on the gradient message
        res[mode] = (params.flat_grad.clone(), [t.clone() for t in stats], before)
    out[rank] = res
    dist.destroy_process_group()


def test_reduce_scatter_all_gather_equals_all_reduce_and_carries_the_refine_sums():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_collective_worker, args=(2, port, out), nprocs=2, join=True)
    for mode in ("all_reduce", "reduce_scatter_all_gather"):
        g0, st0, b0 = out[0][mode]
        g1, st1, b1 = out[1][mode]
        assert torch.equal(g0, g1)
        for a, b, x, y in zip(st0, st1, b0, b1):
            assert torch.equal(a, b) and torch.allclose(a, x + y)          # summed, NOT scaled by 1/G
    assert torch.allclose(out[0]["all_reduce"][0], out[0]["reduce_scatter_all_gather"][0], rtol=1e-6, atol=1e-7)


def _split_worker(rank, world, port, out):
    import torch.distributed as dist
    from gs_sdf_amd.neural_gs import FlatNeuralGS, GSConfig
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(400, 64, 48, sh_degree=0, seed=0)
    cfg = GSConfig(refine_start_iter=1, refine_every=2, reset_every=1000, grow_grad2d=0.3, grow_scale3d=1e-9)   # every grower splits
    gs = FlatNeuralGS(sc["means"], sc["log_scales"], sc["quats"], sc["logit_opacities"], sc["sh"][:, :1], sc["sh"][:, 1:], cfg)
    opt = gs.make_optimizer()
    vp = ViewParallel(gs.params, dist)
    torch.manual_seed(1234 + rank)
    torch.rand(100 * (rank + 1))                         # the ranks have consumed DIFFERENT amounts of global randomness
    g = torch.Generator().manual_seed(7 + rank)          # ... and saw different views: different local statistics
    n = 400
    ids = torch.arange(n)
    info = dict(gradient_2dgs=torch.zeros(n, 2, requires_grad=True), n_cameras=torch.tensor([1]), width=torch.tensor([64]),
                height=torch.tensor([48]), gaussian_ids=ids, visibilities=torch.rand(n, 1, generator=g), radii=torch.ones(n, dtype=torch.int32))
    info["gradient_2dgs"].grad = torch.rand(n, 2, generator=g) * 0.02
    log = gs.train_callback(2, 100, opt, info, view_parallel=vp)
    out[rank] = (log, gs.params.flat.clone(), gs.params.anchors.clone())
    dist.destroy_process_group()


def test_replicas_stay_bit_identical_through_a_split():
    """split() draws random offsets; with per-rank RNG states the replicated splat sets silently diverge.  The
    refinement step seeds its generator from the iteration, and the statistics are merged first, so both ranks must hold
    bit-identical parameters afterwards although their global RNG states and local statistics differ."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_split_worker, args=(2, port, out), nprocs=2, join=True)
    (log0, flat0, anc0), (log1, flat1, anc1) = out[0], out[1]
    assert log0["split"] > 0 and log0 == log1
    assert flat0.shape == flat1.shape and torch.equal(flat0, flat1) and torch.equal(anc0, anc1)
