"""Round-6 entry points of the C ABI (include/gsdf_hip.h, ABI 8), called directly:
  * gsdf_rasterize_2dgs_bwd with and without the forward's workspace (`fwd_ws`: packed splat records + 2x2 reach masks of the (tile, splat) pairs;
    NULL -> the backward runs the pack and mask passes itself): the same gradients either way; a forward without a workspace is refused;
  * gsdf_visible_gather / gsdf_rows_scatter_add / gsdf_nan_rows_accumulate (the joint step's row gathers, gradient row scatters and NaN-row count,
    libtorch index_select / index_add_ / add in the reference: neural_gaussian.cpp:259-262, 907-916) against their torch expressions."""
import pytest
import torch

import gs_sdf_amd.synth as synth

pytestmark = pytest.mark.gpu


def _scene(dev, N=20000, W=320, H=208):
    import gs_sdf_amd.ops as ops
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=3)
    vm = synth.make_views(2, seed=1)[1:2].to(dev)
    d = lambda t: t.to(dev)
    with torch.no_grad():
        cam, gid, radii, m2d, dep, rt, nrm, smp, sw = ops.fully_fused_projection_2dgs(d(sc["means"]), d(sc["quats"]), d(sc["log_scales"].exp()), vm,
                                                                                      d(sc["K"]), W, H, 0.05, 300.0, 0.0)
        col = ops.get_view_colors(vm, d(sc["means"]), radii, d(sc["sh"]), cam, gid, 0)
        opa = torch.sigmoid(d(sc["logit_opacities"]))[gid].contiguous()
        tpg, flat, offs = ops.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid)
    ug = {k: v.to(dev) for k, v in synth.upstream_grads(H, W, seed=2).items()}
    return ops, W, H, m2d, rt, col, opa, nrm, offs, flat, ug


def test_backward_with_and_without_the_forward_workspace():
    dev = torch.device("cuda:0")
    ops, W, H, m2d, rt, col, opa, nrm, offs, flat, ug = _scene(dev)
    with torch.no_grad():
        fwd = ops.rasterize_fwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat)
        g_ws = ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd, ug, None, absgrad=True)
        fwd_no = dict(fwd)
        fwd_no["fwd_ws"] = None                      # the backward packs the records and masks into its own workspace
        g_no = ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd_no, ug, None, absgrad=True)
    torch.cuda.synchronize()
    assert int(flat.numel()) > 0 and float(g_ws["v_colors"].abs().sum()) > 0
    for k in g_ws:
        a, b = g_ws[k], g_no[k]
        # the same kernels on the same records and masks; only the order of the float atomics of the flush differs run to run
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(a.abs().max())), k


def test_forward_refuses_a_missing_workspace():
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    ops, W, H, m2d, rt, col, opa, nrm, offs, flat, ug = _scene(dev, N=3000, W=128, H=96)
    L = capi.lib()
    C, M, I = 1, opa.shape[0], flat.shape[0]
    e = lambda *s: torch.empty(s, device=dev)
    rc, rd, ra, rn, rm = e(C, H, W, 3), e(C, H, W, 1), e(C, H, W, 1), e(C, H, W, 3), e(C, H, W, 1)
    last = torch.empty((C, H, W), dtype=torch.int32, device=dev); med = torch.empty_like(last)
    vis, fT = e(M, 1), e(C, H, W)
    f32, ptr = capi.f32, capi.ptr
    status = L.gsdf_rasterize_2dgs_fwd(C, M, I, W, H, 16, f32(m2d), f32(rt), f32(col), f32(opa), f32(nrm), None, None, ptr(offs, torch.int32),
                                       ptr(flat, torch.int32), f32(rc), f32(rd), f32(ra), f32(rn), f32(rm), ptr(last), ptr(med), f32(vis), f32(fT), None,
                                       capi.stream())
    with pytest.raises(RuntimeError, match="workspace"):
        capi.check(status, "rasterize_2dgs_fwd")
    assert L.gsdf_rasterize_2dgs_fwd_ws_bytes(M, I) >= 128 * M + 8 * I


def test_row_gather_scatter_and_nan_count():
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    L = capi.lib()
    g = torch.Generator().manual_seed(5)
    N, M = 50_000, 17_001
    xyz, opac = torch.randn(N, 3, generator=g).to(dev), torch.rand(N, generator=g).to(dev)
    ids = torch.randperm(N, generator=g)[:M].sort().values.to(dev)
    rows, orow, ones = torch.empty(M, 3, device=dev), torch.empty(M, device=dev), torch.zeros(M, 1, device=dev)
    f32, ptr = capi.f32, capi.ptr
    capi.check(L.gsdf_visible_gather(M, ptr(ids, torch.int64), f32(xyz), f32(opac), f32(rows), f32(orow), f32(ones), capi.stream()), "visible_gather")
    assert torch.equal(rows, xyz.index_select(0, ids)) and torch.equal(orow, opac.index_select(0, ids)) and bool((ones == 1).all())
    capi.check(L.gsdf_visible_gather(M, ptr(ids, torch.int64), f32(xyz), f32(opac), None, f32(orow), None, capi.stream()), "visible_gather (opacity only)")
    # scatter-add: unique ids -> plain read-modify-write; repeated ids -> atomics
    for cols in (1, 3):
        src = torch.randn(M, cols, generator=g).to(dev).contiguous()
        dst = torch.randn(N, cols, generator=g).to(dev).contiguous()
        want = dst.clone().index_add_(0, ids, src)
        capi.check(L.gsdf_rows_scatter_add(M, cols, ptr(ids, torch.int64), 1, f32(src), f32(dst), capi.stream()), "rows_scatter_add")
        assert torch.equal(dst, want)
        rep = torch.randint(0, 64, (M,), generator=g).to(dev)
        dst2 = torch.zeros(64, cols, device=dev)
        capi.check(L.gsdf_rows_scatter_add(M, cols, ptr(rep, torch.int64), 0, f32(src), f32(dst2), capi.stream()), "rows_scatter_add (atomics)")
        assert torch.allclose(dst2, torch.zeros(64, cols, device=dev).index_add_(0, rep, src), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        capi.check(L.gsdf_rows_scatter_add(M, 2, ptr(ids, torch.int64), 1, f32(rows), f32(xyz), capi.stream()), "bad cols")
    # NaN rows: a running total over two calls
    off, sc, q = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev), torch.randn(N, 4, device=dev)
    off[5, 1] = float("nan"); sc[77, 0] = float("nan"); q[77, 3] = float("nan"); q[N - 1, 0] = float("nan")
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(2):
        capi.check(L.gsdf_nan_rows_accumulate(N, f32(off), f32(sc), f32(q), ptr(total), capi.stream()), "nan_rows_accumulate")
    assert int(total.item()) == 6
