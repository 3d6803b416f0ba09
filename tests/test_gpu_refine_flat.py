"""Row a18 on the flat buffers: gsdf_extras::JointIteration::refine / prune_rows / reset_opacity (csrc/refine.hip: ONE row map, two kernels over
the rows, the counts in host-visible words) against gsdf_model::NeuralGS's grow_gs -> prune_gs -> zero_state on a torch::optim::Adam (the
reference's policy and Adam-state surgery, neural_gaussian.cpp:690-926, optimizer_utils.cpp:5-165; itself held to the reference's compiled
code in tests/test_gpu_reference_classes.py): the same splat set in the same order, the same parameters, Adam moments and statistics."""
import os

import pytest
import torch

import gs_sdf_amd.synth as synth

pytestmark = pytest.mark.gpu
FIELDS = ("offsets_", "scaling_", "quaternion_", "opacity_", "features_dc_", "features_rest_")


@pytest.fixture(scope="module")
def host():
    import gs_sdf_amd.hostlib as hostlib
    return hostlib.load()


def _pair(host, N, deg, seed, iter_, radii=False):
    """-> (NeuralGS, its Adam after one step, JointIteration with the same parameters / moments / statistics, RefineConfig)"""
    import gs_sdf_amd.sdf as sdfm
    dev = torch.device("cuda:0")
    W, H = 256, 192
    sc = synth.make_scene(N, W, H, sh_degree=deg, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    ccfg = host.GSConfig()
    ccfg.sh_degree = deg
    if radii:
        ccfg.refine_scale2d_stop_iter = 10_000
    # scales straddling the grow / prune thresholds, opacities straddling prune_opa
    log_scales = torch.log(torch.exp(torch.rand(N, 3, generator=g) * 6.0 - 6.5)).to(dev)       # e^-6.5 .. e^-0.5
    logit_opa = (torch.rand(N, generator=g) * 8.0 - 4.0).to(dev)
    args = [sc["means"].to(dev), log_scales, sc["quats"].to(dev) * 1.7, logit_opa, sc["sh"][:, :1].to(dev), sc["sh"][:, 1:].to(dev)]
    ngs = host.NeuralGS(None, *args, 4, 1.0, ccfg)
    opt = ngs.make_optimizer()
    for k in range(6):
        p = opt.param(k)
        p.grad = torch.randn(p.shape, generator=g).to(dev) * 1e-3
    opt.step()
    fields = [getattr(ngs, f).detach().clone() for f in FIELDS]
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0, device=dev, seed=5)
    enc, dec = host.TCNNEncoding(16, 2, 19, 32, 2.0), host.TCNNNetwork(32, 2, 64, 4, True)
    enc.params_, dec.params_, dec.biases_ = lm.encoder.params_.detach().clone(), lm.decoder.params_.detach().clone(), lm.decoder.biases_.detach().clone()
    ji = host.JointIteration(ngs.anchors_.detach().clone(), fields, enc, dec, [0.0, 0.0, 5.5], 16.0, 0.02, 8, W, H, deg, False, True, True, True)
    m = torch.cat([opt.moments(k)[0].reshape(-1) for k in range(6)])
    v = torch.cat([opt.moments(k)[1].reshape(-1) for k in range(6)])
    ji.set_splat_adam_moments(m, v)
    state = dict(grad2d=(torch.rand(N, generator=g) * 6e-3).to(dev), count=torch.randint(0, 12, (N,), generator=g).float().to(dev),
                 vis=torch.rand(N, generator=g).to(dev))
    if radii:
        state["radii"] = (torch.rand(N, generator=g) * 0.1).to(dev)
    ngs.state = {k: t.clone() for k, t in state.items()}
    ji.set_state({k: t.clone() for k, t in state.items()})
    rc = host.RefineConfig()
    rc.refine_scale2d_stop_iter = ccfg.refine_scale2d_stop_iter
    rc.num_train_data = 4
    return ngs, opt, ji, rc


def _flat_of(ngs):
    return torch.cat([getattr(ngs, f).detach().reshape(-1) for f in FIELDS])


def _assert_same(ngs, opt, ji, what, offsets_tol=2e-6):
    n = ngs.anchors_.shape[0]
    assert ji.n_splats() == n, (what, ji.n_splats(), n)
    assert torch.equal(ji.anchors(), ngs.anchors_), what + ": anchors"
    a, b = ji.splat_flat(), _flat_of(ngs)
    assert a.shape == b.shape
    # offsets of split children go through a batched matmul in the reference's composition (different accumulation order): within rounding
    off = slice(0, 3 * n)
    assert float((a[off] - b[off]).abs().max()) <= offsets_tol * float(b[off].abs().max() + 1.0), what + ": offsets"
    assert torch.equal(a[3 * n:], b[3 * n:]) or bool(((a[3 * n:] == b[3 * n:]) | (a[3 * n:].isinf() & b[3 * n:].isinf())).all()), what + ": parameters"
    jm, jv = ji.splat_adam_moments()
    m = torch.cat([opt.moments(k)[0].reshape(-1) for k in range(6)])
    v = torch.cat([opt.moments(k)[1].reshape(-1) for k in range(6)])
    assert torch.equal(jm, m) and torch.equal(jv, v), what + ": Adam moments"
    st, js = ngs.state, ji.get_state()
    assert set(st) == set(js)
    for k in st:
        assert torch.equal(st[k], js[k]), what + ": state " + k


@pytest.mark.parametrize("N,deg,iter_,radii", [(20_000, 1, 600, False), (50_000, 0, 3100, False), (20_000, 2, 600, True), (1_000_000, 0, 700, False)])
def test_refine_is_grow_then_prune_then_zero_state(host, N, deg, iter_, radii):
    ngs, opt, ji, rc = _pair(host, N, deg, 3, iter_, radii)
    _assert_same(ngs, opt, ji, "before")
    torch.manual_seed(11)
    n_dupli, n_split = ngs.grow_gs(iter_, opt)
    n_prune = ngs.prune_gs(iter_, opt)
    ngs.zero_state()
    torch.manual_seed(11)
    out = ji.refine(iter_, rc)
    assert n_dupli > 100 and n_split > 100 and n_prune > 100, (n_dupli, n_split, n_prune)     # every stage did something
    assert out["n_split"] == n_split and out["N"] == ngs.anchors_.shape[0] == N + n_dupli + n_split - n_prune
    _assert_same(ngs, opt, ji, "after refine")
    # the refined set trains on: a second refinement on top of it (fresh statistics)
    g = torch.Generator().manual_seed(5)
    n2 = ngs.anchors_.shape[0]
    dev = ngs.anchors_.device
    st = dict(grad2d=(torch.rand(n2, generator=g) * 6e-3).to(dev), count=torch.randint(0, 12, (n2,), generator=g).float().to(dev), vis=ngs.state["vis"].clone())
    if radii:
        st["radii"] = (torch.rand(n2, generator=g) * 0.1).to(dev)
    ngs.state = {k: t.clone() for k, t in st.items()}
    ji.set_state({k: t.clone() for k, t in st.items()})
    torch.manual_seed(12)
    ngs.grow_gs(iter_ + 100, opt); ngs.prune_gs(iter_ + 100, opt); ngs.zero_state()
    torch.manual_seed(12)
    ji.refine(iter_ + 100, rc)
    _assert_same(ngs, opt, ji, "after the second refine")


def test_prune_rows_and_reset_opacity(host):
    ngs2, opt2, ji2, rc2 = _pair(host, 10_000, 1, 7, 400)
    st = ngs2.state
    st["vis"][::7] = 0.0
    ngs2.state = st
    js = ji2.get_state(); js["vis"][::7] = 0.0; ji2.set_state(js)
    expect = int((st["vis"] < 1e-4).sum())
    removed = ngs2.prune_invisible_gs(400, opt2)
    out = ji2.train_callback(400, 10_000, rc2)             # 400 % 4 == 0 -> prune_invisible; 400 <= refine_start_iter -> no growing
    assert removed == out["n_invisible"] == expect >= len(range(0, 10_000, 7))
    _assert_same(ngs2, opt2, ji2, "after prune_invisible")
    ngs2.reset_opacity(opt2)
    ji2.reset_opacity(rc2)
    _assert_same(ngs2, opt2, ji2, "after reset_opacity")


def test_refine_then_step_runs(host):
    """the joint iteration keeps training on the refined set: step -> train_callback(refine) -> step, parameters finite, sizes follow"""
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.trainer import SplatParams
    dev = torch.device("cuda:0")
    N, W, H = 30_000, 320, 192
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=0)
    params = SplatParams.from_scene(sc, dev, None)
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0, device=dev, seed=5)
    enc, dec = host.TCNNEncoding(16, 2, 19, 32, 2.0), host.TCNNNetwork(32, 2, 64, 4, True)
    enc.params_, dec.params_, dec.biases_ = lm.encoder.params_.detach().clone(), lm.decoder.params_.detach().clone(), lm.decoder.biases_.detach().clone()
    fields = [params.views[k].detach().clone() for k in ("offsets", "scaling", "quaternion", "opacity", "features_dc", "features_rest")]
    for two in (False, True):
        ji = host.JointIteration(params.anchors, fields, enc, dec, [0.0, 0.0, 5.5], 16.0, 0.02, 8, W, H, 0, two, True, True, True)
        views = synth.make_views(8, seed=1).to(dev)
        K = sc["K"].to(dev)
        target = torch.rand(H, W, 3, device=dev)
        pts = ((torch.rand(4096, 3) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
        sdf = (torch.randn(4096, 1) * 0.02).to(dev)
        rc = host.RefineConfig()
        rc.refine_start_iter, rc.refine_every, rc.grow_grad2d = 2, 3, 1e-7        # grow early and often
        ns = [ji.n_splats()]
        for it in range(1, 10):
            sz = ji.step(views[it % 8][None], K, target, pts, sdf, [], True, [])
            out = ji.train_callback(it, 1000, rc)
            ns.append(out["N"])
            assert out["N"] == ji.n_splats() and int(sz["M"]) <= ns[-2]
        torch.cuda.synchronize()
        assert ns[-1] != ns[0] and len(set(ns)) >= 3, ns
        assert bool(torch.isfinite(ji.splat_flat()[torch.isfinite(ji.splat_flat()) | ~ji.splat_flat().isinf()]).all())
        m, v = ji.splat_adam_moments()
        assert m.numel() == ji.splat_flat().numel() and bool(torch.isfinite(m).all()) and bool(torch.isfinite(v).all())


def test_view_parallel_hook_reduces_every_statistic_exactly_once(host):
    """The view-parallel hook (JointIteration::set_refine_hook) sums grad2d / count and maxes vis / radii over the ranks IN PLACE.  A fake two-rank
    hook (identical ranks: the sums double, the maxima stay) must give the decisions of a single process whose statistics were doubled by hand —
    also in an iteration where the invisible prune AND the refinement fire, and a prune-only iteration must not leave reduced sums behind."""
    seen = []

    def hook(state):
        seen.append(sorted(state.keys()))
        for k in ("grad2d", "count"):
            if k in state:
                state[k].mul_(2.0)

    def run(it, with_hook):
        ngs, opt, ji, rc = _pair(host, 8_000, 0, 11, it, radii=True)
        rc.refine_every, rc.refine_start_iter = 100, 500
        st = ji.get_state()
        if with_hook:
            ji.set_refine_hook(hook)
        else:
            st["grad2d"] = st["grad2d"] * 2.0; st["count"] = st["count"] * 2.0
            ji.set_state(st)
        before = {k: v.clone() for k, v in ji.get_state().items()}
        torch.manual_seed(5)
        out = ji.train_callback(it, 10_000, rc)
        return out, ji, before

    # 600 % 4 == 0 (invisible prune) and 600 % 100 == 0 (refinement): both consumers in one iteration
    seen.clear()
    out_h, ji_h, _ = run(600, True)
    out_r, ji_r, _ = run(600, False)
    assert seen == [["vis"], ["count", "grad2d", "radii"]], seen          # each statistic handed to the hook once, by its consumer
    assert out_h == out_r and out_h.get("n_dupli", 0) + out_h.get("n_split", 0) > 0, (out_h, out_r)
    assert torch.equal(ji_h.splat_flat(), ji_r.splat_flat())
    # 604 % 4 == 0, 604 % 100 != 0: the prune alone — grad2d / count must stay the local accumulators (not reduced, not scaled)
    seen.clear()
    out_p, ji_p, before = run(604, True)
    assert seen == [["vis"]] and "n_invisible" in out_p
    keep = before["vis"] >= 1e-4
    after = ji_p.get_state()
    assert torch.equal(after["grad2d"], before["grad2d"][keep]) and torch.equal(after["count"], before["count"][keep])
