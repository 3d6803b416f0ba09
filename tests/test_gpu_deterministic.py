"""Deterministic mode (include/gsdf_hip.h: gsdf_deterministic, round 6): two runs from the same state produce the same BITS — the compositing
backward's gradients (64-bit fixed point instead of float atomics), the loss values (ordered reductions), the SDF leg's gradients (per-wave partial
buffers, the table scatter's integer sums) and therefore the parameters after several steps of the joint iteration."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    import gs_sdf_amd.hostlib as hostlib
    return hostlib.load()


def _scene(dev, n=60000, W=640, H=400, seed=0):
    import gs_sdf_amd.synth as synth
    sc = synth.make_scene(n, W, H, sh_degree=0, seed=seed)
    vm = synth.make_views(3, seed=1).to(dev)
    return sc, vm


def _raster_grads(dev, sc, vm, absgrad):
    import gs_sdf_amd.ops as ops
    import gs_sdf_amd.synth as synth
    W, H = sc["W"], sc["H"]
    leaves = [sc["means"].to(dev).requires_grad_(True), sc["quats"].to(dev).requires_grad_(True), sc["log_scales"].exp().to(dev).requires_grad_(True),
              torch.sigmoid(sc["logit_opacities"]).to(dev).requires_grad_(True), sc["sh"].to(dev).requires_grad_(True)]
    colors, alphas, meta = ops.rasterization_2dgs_sdf(*leaves, vm[:1], sc["K"].to(dev), W, H, near_plane=0.05, far_plane=300.0, sh_degree=0,
                                                      absgrad=absgrad)
    ug = synth.upstream_grads(H, W, seed=2)
    loss = (colors[..., :3] * ug["v_render_colors"].to(dev)).sum() + (alphas * ug["v_render_alphas"].to(dev)).sum() \
        + (meta["render_normal"] * ug["v_render_normals"].to(dev)).sum() + (meta["render_median"] * ug["v_render_median"].to(dev)).sum() + (colors[..., 3:4] * ug["v_render_depths"].to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    out = [t.grad.clone() for t in leaves]
    if absgrad:
        out.append(meta["absgrad"].grad.clone())
    return out


@pytest.mark.parametrize("absgrad", [False, True])
def test_compositing_backward_is_bit_reproducible_and_agrees_with_the_float_path(absgrad):
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    sc, vm = _scene(dev)
    ref = _raster_grads(dev, sc, vm, absgrad)                       # float atomics
    with capi.deterministic():
        assert capi.lib().gsdf_deterministic(-1) == 1
        a = _raster_grads(dev, sc, vm, absgrad)
        b = _raster_grads(dev, sc, vm, absgrad)
    assert capi.lib().gsdf_deterministic(-1) == 0
    for x, y in zip(a, b):
        assert torch.isfinite(x).all() and float(x.abs().sum()) > 0
        assert torch.equal(x, y), "two deterministic runs differ"
    # against the float path: the fixed point's unit is 2^-34 of the largest upstream gradient, each record a sum of <= 2.6e5 such roundings
    for x, r in zip(a, ref):
        # (measured 2.5e-6 for d/d means: the float path's own run-to-run spread — its fp32 atomics round every partial sum of moments of 1e4-1e5,
        #  the fixed-point sums are exact to 5e-10)
        assert float((x.double() - r.double()).norm()) <= 1e-5 * float(r.double().norm())


def test_out_of_range_tile_sum_poisons_the_launch():
    """a tile sum of 2^13 times the launch's largest upstream gradient cannot be told from an overflow of the total: NaN, not a wrapped integer"""
    import gs_sdf_amd.capi as capi
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    sc, vm = _scene(dev, n=3000, W=256, H=192)
    W, H = sc["W"], sc["H"]
    means = sc["means"].to(dev).requires_grad_(True)
    args = (means, sc["quats"].to(dev), sc["log_scales"].exp().to(dev), torch.sigmoid(sc["logit_opacities"]).to(dev), sc["sh"].to(dev) * 3e5)
    with capi.deterministic():
        colors, alphas, meta = ops.rasterization_2dgs_sdf(*args, vm[:1], sc["K"].to(dev), W, H, near_plane=0.05, far_plane=300.0, sh_degree=0)
        # colours of 1e5: d loss / d opacity of a splat sums colour x T over a tile, far beyond 2^13 x the unit upstream gradient
        (colors[..., :3].sum()).backward()
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(means.grad).all())


def test_joint_iteration_is_bit_reproducible(host):
    """six steps of gsdf_extras::JointIteration (two streams, direct splat leg, analytic SDF configuration) twice from the same state"""
    import gs_sdf_amd.capi as capi
    from benchlib.steps import make_cpp_iteration
    import argparse
    import gs_sdf_amd.synth as synth
    from gs_sdf_amd.trainer import SplatParams
    dev = torch.device("cuda:0")
    N, W, H = 120000, 960, 544
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=0)
    views = synth.make_views(4, seed=1).to(dev)
    K = sc["K"].to(dev)
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    args = argparse.Namespace(no_overlap=False, sdf_config="default", step_terms="reference", sample_mode="center", hashgrid_resident=-1, no_sdf=False)

    def run():
        torch.manual_seed(0)
        params = SplatParams.from_scene(sc, dev, None)
        ji, pool, rsdf, cams, _ = make_cpp_iteration(args, sc, params, dev, W, H, 0, views)
        losses = []
        for i in range(6):
            vi = i % views.shape[0]
            ji.step(views[vi][None], K, target, pool[i % 8], rsdf[i % 8], [], True, cams[vi])
            torch.cuda.synchronize()
            losses.append([[float(v) for v in t.reshape(-1)] for t in ji.last_losses()])
        ji.sync()
        torch.cuda.synchronize()
        return ji.splat_flat().clone(), ji.sdf_flat().clone(), losses

    with capi.deterministic():
        a = run()
        b = run()
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    assert torch.equal(a[0], b[0]), "splat parameters differ between two deterministic runs"
    assert torch.equal(a[1], b[1]), "SDF parameters differ between two deterministic runs"
    assert a[2] == b[2], "loss values differ between two deterministic runs"


def test_timing_trace_is_a_timeline_of_the_entry_points():
    """gsdf_timing_trace (include/gsdf_hip.h): the HIP-event pairs of the C-ABI timers in call order, begin <= end, relative to the first call"""
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    L = capi.lib()
    x = torch.rand(5000, 3, device=dev)
    table = torch.rand(7634944, 2, device=dev)
    feat = torch.empty(5000, 32, device=dev)
    capi.timing_begin(None)
    for _ in range(3):
        capi.check(L.gsdf_hashgrid_fwd(5000, 16, 2, 19, 32, 2.0, capi.f32(x), capi.f32(table), capi.f32(feat), capi.stream()), "fwd")
    torch.cuda.synchronize()
    tr = capi.timing_trace()
    assert [n for n, _, _ in tr] == ["gsdf_hashgrid_fwd"] * 3
    assert tr[0][1] == 0.0 and all(b >= a >= 0.0 for _, a, b in tr) and tr[1][1] >= tr[0][2] - 1e-3 and tr[2][1] >= tr[1][2] - 1e-3
    assert capi.timing_trace() == []          # the collection is handed over once
