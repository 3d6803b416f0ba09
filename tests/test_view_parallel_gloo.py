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
