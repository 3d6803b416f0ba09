"""GPU parity tests of the SDF hot path (hash-grid encoding S1 with first/second order backward, fused
MFMA decoder S2, LocalMap.get_sdf/get_gradient): every operator goes through the C ABI
(gs_sdf_amd.sdf -> libgsdf_hip.so) and is compared with the CPU oracle on the same seeded inputs.
Bar: within 1e-4 relative (fp32); see tests/util.py:assert_close for the precise statement."""
import numpy as np
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu
REL = 1e-4
CFG = dict(n_levels=16, n_feat=2, log2_hashmap=19, base_res=32, per_level_scale=2.0)   # base.yaml:8-10
CFG_SMALL = dict(n_levels=7, n_feat=2, log2_hashmap=12, base_res=8, per_level_scale=1.5)


def tcfg(c):
    return dict(otype="Grid", type="Hash", n_levels=c["n_levels"], n_features_per_level=c["n_feat"],
                log2_hashmap_size=c["log2_hashmap"], base_resolution=c["base_res"], per_level_scale=c["per_level_scale"],
                interpolation="Linear")


@pytest.fixture(scope="module")
def sdf():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    import gs_sdf_amd.capi as capi
    capi.lib()
    import gs_sdf_amd.sdf as s
    return s


def n(t):
    return t.detach().cpu().numpy()


def _points(B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, generator=g)
    if B >= 6:
        x[:6] = torch.tensor([0.0, 1.0, 0.5, 0.25, 1.0 - 2 ** -20, 2 ** -21])[:, None]  # edges and cell boundaries
    return x, g


@pytest.mark.parametrize("cfg,B", [(CFG, 32768), (CFG, 20000), (CFG_SMALL, 1000), (CFG, 1)])     # 32768: binned scatter, 20000: atomic
def test_hashgrid_fwd_bwd_bwdbwd(sdf, oracle, cfg, B):
    dev = torch.device("cuda:0")
    enc = sdf.TCNNEncoding(3, tcfg(cfg), "enc", dev, seed=1)
    offs, total = oracle.grid_offsets(cfg)
    assert enc.offsets == list(offs) and enc.params_.numel() == total * 2
    assert float(enc.params_.detach().abs().max()) <= 1e-4
    x, g = _points(B, 0)
    table = (torch.rand(total * 2, generator=g) * 2 - 1)          # O(1) values make the check meaningful
    enc.params_ = table.to(dev).requires_grad_(True)
    xd = x.to(dev).requires_grad_(True)
    feat = enc.forward(xd)
    # The oracle's f32 build is the reference here: pos = fmaf(scale, x, 0.5) is evaluated in fp32 as in
    # tiny-cuda-nn, and at the finest levels (scale = 2^20) the fp32 fraction is quantised to 1/16, so an
    # fp64 evaluation is a DIFFERENT function (features differ by O(0.1)), not a more accurate one.
    PREC = "f32"
    f32 = oracle.grid_fwd(n(x), n(table).reshape(-1, 2), cfg, prec=PREC)
    assert_close(feat, f32, 1e-5, "features (same cell indices, same weights)")
    v = torch.randn(B, feat.shape[1], generator=g)
    vd = v.to(dev).requires_grad_(True)
    v_x, v_t = torch.autograd.grad(feat, (xd, enc.params_), vd, create_graph=True)
    vt_o, vx_o = oracle.grid_bwd(n(x), n(table).reshape(-1, 2), n(v), cfg, prec=PREC)
    assert_close(v_x, vx_o, REL, "v_x")
    assert_close(v_t.view(-1, 2), vt_o, REL, "v_table")
    vv = torch.randn(B, 3, generator=g)
    vvd = vv.to(dev).requires_grad_(True)
    g_v, g_t, g_x = torch.autograd.grad((v_x * vvd).sum(), (vd, enc.params_, xd), create_graph=True)
    gv_o, gt_o, gx_o = oracle.grid_bwd_bwd(n(x), n(table).reshape(-1, 2), n(v), n(vv), cfg, prec=PREC)
    assert_close(g_v, gv_o, REL, "double backward: d/d v_feat")
    assert_close(g_t.view(-1, 2), gt_o, REL, "double backward: d/d table")
    assert_close(g_x, gx_o, REL, "double backward: d/d x")
    # third order (a loss on the analytic Hessian, local_map.cpp:163-168): loss3 = <lam, g_x> (+ <mu, g_v>); the oracle's orc_grid_bwd3 is pinned
    # to a torch-fp64 autograd restatement on the CPU (tests/test_oracle_sdf_selfcheck.py)
    lam = torch.randn(B, 3, generator=g)
    mu = torch.randn(B, feat.shape[1], generator=g)
    for use_mu in (False, True):
        loss3 = (g_x * lam.to(dev)).sum() + ((g_v * mu.to(dev)).sum() if use_mu else 0.0)
        t_v, t_t, t_vv, t_x = torch.autograd.grad(loss3, (vd, enc.params_, vvd, xd), retain_graph=True)
        o_v, o_t, o_vv, o_x = oracle.grid_bwd3(n(x), n(table).reshape(-1, 2), n(v), n(vv), n(lam), n(mu) if use_mu else None, cfg, prec=PREC)
        if B >= 100:
            assert_close(t_v, o_v, REL, "third order: d/d v_feat")
            assert_close(t_t.view(-1, 2), o_t, REL, "third order: d/d table")
            assert_close(t_vv, o_vv, REL, "third order: d/d vv_x")
            assert_close(t_x, o_x, REL, "third order: d/d x")
        else:
            # a single point: an entry's value is a sum of three mixed-derivative products that may cancel (fp32 in the kernel and in the f32
            # oracle, differently associated), and there is no tensor mean to judge it against: the bar is relative to the tensor's maximum
            for got, ref, what in ((t_v, o_v, "v_feat"), (t_t.view(-1, 2), o_t, "table"), (t_vv, o_vv, "vv_x"), (t_x, o_x, "x")):
                e = np.abs(n(got).astype(np.float64) - ref).max()
                assert e <= REL * np.abs(ref).max() + 1e-30, f"third order: d/d {what}: {e:.3e} against max {np.abs(ref).max():.3e}"


def _near_relu_kink(x, dims, W, b, eps=1e-5):
    h, off, boff, risky = x.astype(np.float64), 0, 0, np.zeros(len(x), bool)
    for l in range(len(dims) - 2):
        z = h @ W[off:off + dims[l] * dims[l + 1]].astype(np.float64).reshape(dims[l + 1], dims[l]).T
        if b is not None:
            z = z + b[boff:boff + dims[l + 1]]
        off, boff = off + dims[l] * dims[l + 1], boff + dims[l + 1]
        risky |= (np.abs(z) < eps).any(axis=1)
        h = np.maximum(z, 0.0)
    return risky


@pytest.mark.parametrize("dims,bias,B", [([32, 64, 64, 64, 64, 2], True, 32768),     # torch decoder topology
                                         ([32, 64, 64, 64, 2], False, 5000),         # tcnn FullyFusedMLP topology
                                         ([64, 64, 64, 16], True, 77),               # fp32-pipe kernels (input width 64)
                                         ([32, 64, 64, 64, 3], True, 4099),          # one-pass backward, 4 layers with biases
                                         ([32, 64, 64, 64, 64, 20], False, 1000),    # ... 5 layers, outputs in both k-steps of the last layer
                                         ([32, 64, 64, 64, 16], False, 300),         # widest output the one-pass backward takes
                                         ([32, 64, 64, 64, 64, 17], True, 300),      # one more: fp32-pipe backward, split forward
                                         ([32, 64, 64, 64, 2], False, 1),
                                         ([32, 64, 64, 64, 64, 2], True, 33)])
def test_fused_mlp_fwd_bwd(sdf, oracle, dims, bias, B):
    dev = torch.device("cuda:0")
    net = sdf.TCNNNetwork(dims[0], dims[-1], dict(n_neurons=64, n_hidden_layers=len(dims) - 2), "dec", dev, bias=bias, seed=3)
    assert net.dims == dims
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, dims[0], generator=g)
    W, b = n(net.params_), (n(net.biases_) if bias else None)
    # a ReLU network's gradients are discontinuous where a pre-activation crosses zero: points with one within 1e-5 of it
    # (where evaluations that round differently may disagree on the sign) are taken out of the batch
    x = x[~torch.from_numpy(_near_relu_kink(n(x), dims, W, b))]
    B = x.shape[0]
    xd = x.to(dev).requires_grad_(True)
    out = net.forward(xd)
    ref, acts = oracle.mlp_fwd(n(x), dims, W, b, want_acts=True, prec="f64")
    assert_close(out, ref, REL, "mlp out")
    v = torch.randn(B, dims[-1], generator=g)
    out.backward(v.to(dev))
    v_in, v_w, v_b = oracle.mlp_bwd(n(x), dims, W, b, n(v), prec="f64")
    assert_close(xd.grad, v_in, REL, "mlp v_in")
    assert_close(net.params_.grad, v_w, REL, "mlp v_weights")
    if bias:
        assert_close(net.biases_.grad, v_b, REL, "mlp v_biases")
    with torch.no_grad():          # inference path (no saved activations)
        assert_close(net.forward(xd.detach()), ref, REL, "mlp out (no grad)")


@pytest.mark.parametrize("impl", [1, 2])
def test_local_map_get_sdf_and_numerical_gradient(sdf, oracle, impl):
    dev = torch.device("cuda:0")
    lm = sdf.LocalMap([0.5, -1.0, 0.25], 8.0, bce_sigma=0.02, decoder_implementation=impl, device=dev, seed=2)
    g = torch.Generator().manual_seed(9)
    lm.encoder.params_ = ((torch.rand(lm.encoder.params_.numel(), generator=g) * 2 - 1) * 0.5).to(dev).requires_grad_(True)
    B = 4096
    xyz = (torch.rand(B, 3, generator=g) - 0.5) * 7.0 + torch.tensor([0.5, -1.0, 0.25])
    xd = xyz.to(dev)
    sdf_v, isig = lm.get_sdf(xd)
    # oracle composition (sub_map.cpp:82-97 normalisation, local_map.cpp:87-103 head)
    x01 = 0.5 * ((n(xyz) - np.array([0.5, -1.0, 0.25], np.float32)) * np.float32(2.0 / 8.0)) + 0.5
    table = n(lm.encoder.params_).reshape(-1, 2)
    W = n(lm.decoder.params_)
    b = n(lm.decoder.biases_) if lm.decoder.biases_ is not None else None

    def o_sdf(p01):
        feat = oracle.grid_fwd(p01, table, CFG, prec="f32")     # fp32 cell addressing, see test_hashgrid_*
        return oracle.mlp_fwd(feat, lm.decoder.dims, W, b, prec="f64")
    out = o_sdf(x01)
    s_ref, i_ref = oracle.sdf_head(out, 1.0 / 0.02, prec="f64")
    assert_close(sdf_v[:, 0], s_ref, REL, "sdf")
    assert_close(isig[:, 0], i_ref, REL, "isigma")
    delta = 0.02
    grad, hess = lm.get_gradient(xd, delta, sdf_v, hessian=True, numerical_grad=True)
    gref = np.zeros((B, 3)); href = np.zeros((B, 3))
    for d in range(3):
        e = np.zeros(3, np.float32); e[d] = delta
        xp = 0.5 * ((n(xyz) + e - np.array([0.5, -1.0, 0.25], np.float32)) * np.float32(2.0 / 8.0)) + 0.5
        xm = 0.5 * ((n(xyz) - e - np.array([0.5, -1.0, 0.25], np.float32)) * np.float32(2.0 / 8.0)) + 0.5
        sp, sm = o_sdf(xp)[:, 0], o_sdf(xm)[:, 0]
        gref[:, d] = 0.5 / delta * (sp - sm)
        href[:, d] = (sp + sm - 2 * out[:, 0]) / delta ** 2
    assert_close(grad, gref, 5e-4, "numerical gradient (central differences amplify fp32 round-off by 1/delta)")
    # loss backward through the fused path reaches every parameter
    loss = sdf.sdf_loss(sdf_v, torch.zeros_like(sdf_v), isig) + 0.1 * sdf.eikonal_loss(grad)
    loss.backward()
    for p in lm.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0


def test_analytic_eikonal_double_backward_matches_oracle(sdf, oracle):
    """Reference default (numerical_grad: 0, decoder_implementation: 0): SDF gradient by autograd through the
    HIP encoder + fused decoder, eikonal loss on it, backward again (double backward through encoder AND decoder kernels).
    ReLU masks are piecewise constant, so d(eikonal)/d(table) is exactly the oracle's grid double backward."""
    dev = torch.device("cuda:0")
    lm = sdf.LocalMap([0.0, 0.0, 0.0], 2.0, decoder_implementation=0, device=dev, seed=5)
    g = torch.Generator().manual_seed(11)
    lm.encoder.params_ = ((torch.rand(lm.encoder.params_.numel(), generator=g) * 2 - 1) * 0.3).to(dev).requires_grad_(True)
    B = 2048
    xyz = (torch.rand(B, 3, generator=g) - 0.5) * 1.8
    xd = xyz.to(dev).requires_grad_(True)
    sdf_v, _ = lm.get_sdf(xd)
    grad = lm.get_gradient(xd, 0.02, sdf_v, hessian=False, numerical_grad=False)[0]
    loss = sdf.eikonal_loss(grad)
    loss.backward()
    # oracle: J = d sdf / d feat from the decoder, grad_x = J . dfeat/dx, vv = dL/dgrad_x
    x01 = (0.5 * (n(xyz) * np.float32(2.0 / 2.0)) + 0.5).astype(np.float32)
    table = n(lm.encoder.params_).reshape(-1, 2)
    W, b = n(lm.decoder.params_), n(lm.decoder.biases_)            # decoder_implementation 0 on the fused kernels (biases)
    dims = [32, 64, 64, 64, 64, 2]
    assert lm.decoder.dims == dims
    feat = oracle.grid_fwd(x01, table, CFG, prec="f32")
    v_out = np.zeros((B, 2)); v_out[:, 0] = 1.0
    J, _, _ = oracle.mlp_bwd(feat, dims, W, b, v_out, prec="f64")
    _, gx01 = oracle.grid_bwd(x01, table, J, CFG, prec="f32")
    gx = gx01 * (0.5 * 2.0 / 2.0)                                   # chain rule of xyz -> [0,1] normalisation
    assert_close(grad, gx, REL, "analytic SDF gradient")
    nrm = np.linalg.norm(gx, axis=1, keepdims=True)
    vv = (2.0 * (nrm - 1.0) / B) * gx / nrm * (0.5 * 2.0 / 2.0)
    _, gt, _ = oracle.grid_bwd_bwd(x01, table, J, vv, CFG, prec="f32")
    # vv contains (|g|-1): a cancellation that turns the 1e-6 fp32 error of g into up to ~1e-3 of vv for the
    # points with |g| ~ 1, and fine-level table entries are touched by single points -> 2e-3 here; the
    # operator-level double backward is held to 1e-4 in test_hashgrid_fwd_bwd_bwdbwd.
    assert_close(lm.encoder.params_.grad.view(-1, 2), gt, 2e-3, "d eikonal / d table (double backward)")


@pytest.mark.parametrize("sizes,lrs", [([1003, 4096, 7, 250_001], [1.6e-4, 5e-3, 5e-2, 1e-3]),
                                       # segments shorter than a float4 and EMPTY ones with unaligned boundaries (features_rest
                                       # at sh_degree 0 is an empty group in the middle of the splat buffer)
                                       ([5, 1, 0, 2, 1, 0, 0, 3, 4099, 1], [1e-1, 2e-2, 9.0, 3e-3, 4e-2, 9.0, 9.0, 5e-3, 6e-4, 7e-2])])
def test_fused_adam_matches_torch_adam(sizes, lrs):
    """gsdf_adam_step over a flat buffer with per-segment learning rates == torch.optim.Adam with the same groups."""
    from gs_sdf_amd.trainer import FusedAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = sum(sizes)
    flat = torch.randn(n, generator=g).to(dev)
    flat_grad = torch.zeros(n, device=dev)
    ref_params = [torch.nn.Parameter(t.clone()) for t in flat.split(sizes)]
    ref = torch.optim.Adam([dict(params=[p], lr=lr) for p, lr in zip(ref_params, lrs)], eps=1e-15)
    opt = FusedAdam(eps=1e-15)
    opt.add_group(flat, flat_grad, list(zip(sizes, lrs)))
    for it in range(6):
        grad = torch.randn(n, generator=g).to(dev) * (10.0 ** (it - 3))
        flat_grad.copy_(grad)
        for p, gg in zip(ref_params, grad.split(sizes)):
            p.grad = gg.clone()
        ref.step(); opt.step(zero_grad=bool(it % 2))
        assert_close(flat, torch.cat([p.detach() for p in ref_params]), 1e-6, f"params after step {it + 1}")
        # gsdf_adam_step_zero_grad leaves the gradient buffer zeroed (every element, the ragged tail too); gsdf_adam_step leaves it alone
        assert torch.equal(flat_grad, torch.zeros_like(grad) if it % 2 else grad), f"gradient buffer after step {it + 1}"


def test_in_place_table_gradient_accumulation_equals_autograd(sdf):
    """LocalMap.flatten(accumulate_table_grad_in_place=True) (trainer fast path) gives the same parameter gradients as
    plain autograd accumulation over several get_sdf calls."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    xs = [((torch.rand(n, 3, generator=g) - 0.5) * 1.5).to(dev) for n in (3000, 5000)]
    grads = []
    for inplace in (False, True):
        lm = sdf.LocalMap([0.0, 0.0, 0.0], 2.0, decoder_implementation=1, device=dev, seed=7)
        grp = lm.flatten(accumulate_table_grad_in_place=inplace)
        loss = sum((lm.get_sdf(x)[0] ** 2).sum() for x in xs)
        if inplace:
            with sdf.grad_sinks_armed():
                loss.backward()
        else:
            loss.backward()
        grads.append(grp.flat_grad.clone())
    assert float(grads[0].abs().sum()) > 0
    assert_close(grads[1], grads[0], 1e-5, "flat gradient (in-place sink vs autograd)")


def test_query_points_bit_identical_to_the_reference_expression(sdf):
    """gsdf_sdf_query_points == 0.5 * ((xyz - pos) * 2 * map_size_inv) + 0.5 (sub_map.cpp:82-97) bit for bit, and
    its stencil rows == that expression on xyz[None] + offsets (local_map.cpp:112-124); gradient = torch's."""
    dev = torch.device("cuda:0")
    lm = sdf.LocalMap([0.5, -1.0, 0.25], 12.0, decoder_implementation=1, device=dev, seed=2)
    xyz = ((torch.rand(4099, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 11.0).to(dev).requires_grad_(True)
    ref = lambda p: 0.5 * ((p - lm.pos_W_M) * 2 * lm.map_size_inv) + 0.5
    assert torch.equal(lm.query_points(xyz), ref(xyz)) and torch.equal(lm.query_points(xyz), lm.xyz_to_zp1_pts(xyz))
    d = 0.013
    offs = torch.tensor([[d, 0, 0], [-d, 0, 0], [0, d, 0], [0, -d, 0], [0, 0, d], [0, 0, -d]], device=dev)[:, None, :]
    want = torch.cat([ref(xyz), ref((xyz[None] + offs).view(-1, 3))], 0)
    got = lm.query_points(xyz, d)
    assert torch.equal(got, want)
    v = torch.randn_like(want)
    assert_close(torch.autograd.grad((got * v).sum(), xyz)[0], torch.autograd.grad((want * v).sum(), xyz)[0], 1e-6, "v_xyz")


@pytest.mark.parametrize("n", [1, 777, 32768])
def test_fused_ray_loss_matches_the_reference_composition(sdf, n):
    """LocalMap.ray_loss (one encoder / decoder / loss launch over the 7n points) == loss::sdf_loss(get_sdf) +
    w * loss::eikonal_loss(get_gradient numerical) composed from the mirrored torch expressions (loss.cpp:49-83,
    local_map.cpp:87-131): value and every parameter gradient."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    xyz = ((torch.rand(n, 3, generator=g) - 0.5) * 6.0).to(dev)
    gt = (torch.randn(n, 1, generator=g) * 0.05).to(dev)
    gt[::5] *= 20.0                                            # some targets saturate the 1e-7 clamp
    out = []
    for fused in (False, True):
        lm = sdf.LocalMap([0.1, 0.2, -0.3], 8.0, bce_sigma=0.02, decoder_implementation=1, device=dev, seed=9)
        with torch.no_grad():                                  # make the isigma head / its 5e2 clamp live
            lm.decoder.params_.mul_(4.0)
        if fused:
            loss = lm.ray_loss(xyz, gt, 0.02, 0.1)
        else:
            s, isig = lm.get_sdf(xyz)
            loss = sdf.sdf_loss(s, gt, isig) + 0.1 * sdf.eikonal_loss(lm.get_gradient(xyz, 0.02, s, False, True)[0])
        grads = torch.autograd.grad(loss, lm.parameters())
        out.append((loss.detach(), grads))
    assert_close(out[1][0], out[0][0], 1e-5, "ray loss value")
    for a, b, name in zip(out[1][1], out[0][1], ("table", "mlp")):
        assert float(b.abs().max()) > 0
        assert_close(a, b, 2e-4, f"ray loss gradient wrt {name}")


def test_fused_ray_loss_elementwise_against_fp64_torch(sdf):
    """gsdf_sdf_ray_loss alone on synthetic decoder outputs, against the fp64 torch expressions: covers softplus beyond
    its threshold, the isigma clamp, saturated targets and a zero numerical gradient (norm backward = 0)."""
    dev = torch.device("cuda:0")
    n = 5000
    g = torch.Generator().manual_seed(5)
    attr = torch.randn(7 * n, 2, generator=g) * torch.tensor([0.2, 0.3])
    attr[:50, 1] = 0.5                                         # softplus threshold branch, isigma > 5e2 -> clamped
    attr[n:, 1] = 7.0                                          # stencil rows' raw column must not matter
    for k in range(6):
        attr[n + k * n + 100: n + k * n + 110, 0] = 0.25       # zero gradient points
    gt = torch.randn(n, 1, generator=g) * 0.05
    gt[:20] = 1.0                                              # sigmoid(-gt*isigma) underflows the 1e-7 clamp
    a64 = attr.double().requires_grad_(True)
    s, raw = a64[:n, 0:1], a64[:n, 1:2]
    isig = (1 + torch.nn.functional.softplus(raw, beta=100) * 50.0)
    ps = a64[n:, 0:1].view(6, n, 1)
    grad = 0.5 / 0.02 * torch.cat([ps[0] - ps[1], ps[2] - ps[3], ps[4] - ps[5]], 1)
    want = sdf.sdf_loss(s, gt.double(), isig) + 0.1 * sdf.eikonal_loss(grad)
    v_want = torch.autograd.grad(want, a64)[0]
    a32 = attr.to(dev).requires_grad_(True)
    got = sdf._SdfRayLoss.apply(a32, gt.to(dev), 50.0, 0.02, 0.1, n)
    v_got = torch.autograd.grad(got * 3.0, a32)[0]
    assert_close(got, want.float().to(dev), 1e-5, "loss")
    assert_close(v_got, 3.0 * v_want.float().to(dev), 1e-4, "d loss / d attr")
    assert float(v_got[n:, 1].abs().max()) == 0.0


@pytest.mark.parametrize("B", [1, 7, 4099])
def test_saved_jacobian_input_gradient_equals_table_walk(sdf, B):
    """TCNNEncoding.save_jacobian: d/dx from the Jacobian stored by the forward == the backward kernel's d/dx, and the
    second-order path (create_graph) is untouched by the option."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B)
    x0 = torch.rand(B, 3, generator=g).to(dev)
    v = torch.randn(B, 32, generator=g).to(dev)
    out = {}
    for jac in (False, True):
        enc = sdf.TCNNEncoding(3, None, "enc", dev, seed=3)
        with torch.no_grad():
            enc.params_.mul_(1e3)
        enc.save_jacobian = jac
        x = x0.clone().requires_grad_(True)
        f = enc.forward(x)
        gx, gt = torch.autograd.grad((f * v).sum(), (x, enc.params_))
        x2 = x0.clone().requires_grad_(True)
        f2 = enc.forward(x2)
        g1 = torch.autograd.grad((f2 * v).sum(), x2, create_graph=True)[0]
        gg = torch.autograd.grad((g1 ** 2).sum(), enc.params_)[0]
        out[jac] = (f.detach(), gx, gt, gg)
    assert torch.equal(out[True][0], out[False][0])
    assert_close(out[True][1], out[False][1], 1e-5, "d/dx (jacobian vs table walk)")
    assert_close(out[True][2], out[False][2], 1e-5, "table gradient")
    assert_close(out[True][3], out[False][3], 1e-5, "second-order table gradient")


def test_fused_gs_sdf_loss_matches_the_reference_composition(sdf):
    """LocalMap.gs_sdf_loss == scale * loss::gs_sdf_loss(get_sdf(x)[0], w[ids]) (loss.cpp:7-11, neural_mapping.cpp:436-462):
    value, parameter gradients and the gradient w.r.t. the sample points."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    M, n = 9000, 4000
    pts = ((torch.rand(M, 3, generator=g) - 0.5) * 6.0).to(dev)
    w_all = torch.rand(M, 1, generator=g).to(dev)
    ids = torch.randperm(M, generator=g)[:n].sort().values.to(dev)
    out = []
    for fused in (False, True):
        lm = sdf.LocalMap([0.1, 0.2, -0.3], 8.0, decoder_implementation=1, device=dev, seed=9)
        with torch.no_grad():
            lm.encoder.params_.mul_(1e3)
        x = pts.clone().requires_grad_(True)
        xs = x.index_select(0, ids)
        loss = lm.gs_sdf_loss(xs, w_all, ids, 1e-3) if fused else 1e-3 * sdf.gs_sdf_loss(lm.get_sdf(xs)[0], w_all.index_select(0, ids))
        out.append((loss.detach(), torch.autograd.grad(loss, [x] + lm.parameters())))
    assert_close(out[1][0], out[0][0], 1e-5, "gs_sdf loss value")
    for a, b, name in zip(out[1][1], out[0][1], ("points", "table", "mlp")):
        assert float(b.abs().max()) > 0
        assert_close(a, b, 2e-4, f"gs_sdf loss gradient wrt {name}")


@pytest.mark.parametrize("eik", [False, True])
def test_single_node_coupling_leg_equals_composed_operators(sdf, eik):
    """LocalMap.gs_sdf_coupling (one autograd node, in-place gradient sinks) == the composed operators of the joint
    iteration (neural_mapping.cpp:436-457): gs_sdf_loss(get_sdf(samples)) [+ w_eik * eikonal_loss(get_gradient(
    samples.detach(), delta, numerical))]: loss, d/d samples and the flat parameter gradients."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(6)
    M, n = 7000, 3100
    pts = ((torch.rand(M, 3, generator=g) - 0.5) * 6.0).to(dev)
    w_all = torch.rand(M, 1, generator=g).to(dev)
    ids = torch.randperm(M, generator=g)[:n].sort().values.to(dev)
    delta, w_eik = (0.02, 0.1) if eik else (None, 0.0)
    out = []
    for fused in (False, True):
        lm = sdf.LocalMap([0.1, 0.2, -0.3], 8.0, decoder_implementation=1, device=dev, seed=9)
        with torch.no_grad():
            lm.encoder.params_.mul_(1e3)
        grp = lm.flatten(accumulate_table_grad_in_place=True)
        lm.encoder.save_jacobian = True
        x = pts.clone().requires_grad_(True)
        if fused:
            loss = lm.gs_sdf_coupling(x, ids, w_all, 1e-3, delta, w_eik)
        else:
            xs = x.index_select(0, ids)
            loss = lm.gs_sdf_loss(xs, w_all, ids, 1e-3)
            if eik:
                loss = loss + w_eik * sdf.eikonal_loss(lm.get_gradient(xs.detach(), delta, None, False, True)[0])
        with sdf.grad_sinks_armed():
            loss.backward()
        out.append((loss.detach(), x.grad.clone(), grp.flat_grad.clone()))
    assert_close(out[1][0], out[0][0], 1e-5, "loss")
    assert_close(out[1][1], out[0][1], 1e-5, "d/d samples")
    # eikonal: differences of nearly equal SDF values times 1/(2 delta) -> the two evaluation orders (and the fp32 atomics of
    # the table scatter, whose summation order is not fixed) differ by up to a few 1e-4 of the mean gradient magnitude
    # (without it: the decoder's weight gradients leave the one-pass backward through fp32 atomics, one round per wave, in
    # an order that differs from launch to launch -> a few 1e-5 of the mean gradient magnitude)
    assert_close(out[1][2], out[0][2], 1e-3 if eik else 1e-4, "flat parameter gradients")
    with pytest.raises(RuntimeError):
        sdf.LocalMap([0, 0, 0], 2.0, decoder_implementation=1, device=dev).gs_sdf_coupling(pts, ids, w_all)
    with pytest.raises(RuntimeError):          # the node writes gradients in place: only inside grad_sinks_armed()
        lm.gs_sdf_coupling(pts.clone().requires_grad_(True), ids, w_all, 1e-3).backward()


def test_gradient_sinks_fire_only_when_armed(sdf):
    """After LocalMap.flatten(accumulate_table_grad_in_place=True) a traversal that is NOT the trainer's backward —
    torch.autograd.grad w.r.t. the inputs, with or without create_graph (LocalMap::get_gradient's analytic branch,
    local_map.cpp:151-172) — must not write into the flat gradient buffer; the same graph under grad_sinks_armed()
    deposits exactly what autograd would have accumulated."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    xyz = ((torch.rand(500, 3, generator=g) - 0.5) * 6.0).to(dev)
    lm = sdf.LocalMap([0.1, 0.2, -0.3], 8.0, decoder_implementation=1, device=dev, seed=9)
    with torch.no_grad():
        lm.encoder.params_.mul_(1e3)
    grp = lm.flatten(accumulate_table_grad_in_place=True)
    x = xyz.clone().requires_grad_(True)
    s = lm.get_sdf(x)[0]
    torch.autograd.grad(s.sum(), x, retain_graph=True)
    assert float(grp.flat_grad.abs().sum()) == 0.0, "autograd.grad w.r.t. inputs wrote parameter gradients"
    feat = lm.encoder.forward(lm.query_points(x))
    torch.autograd.grad(feat.sum(), x, create_graph=True)
    assert float(grp.flat_grad.abs().sum()) == 0.0, "create_graph traversal wrote parameter gradients"
    s.sum().backward(retain_graph=True)                     # plain autograd accumulation into .grad (views of flat_grad)
    plain = grp.flat_grad.clone()
    grp.flat_grad.zero_()
    with sdf.grad_sinks_armed():
        s.sum().backward()
    assert float(plain.abs().sum()) > 0
    assert_close(grp.flat_grad, plain, 1e-4, "armed sinks vs autograd accumulation")   # fp32 atomics of the weight gradients: launch-to-launch order


@pytest.mark.parametrize("jac", [False, True])
def test_xcd_partitioned_forward_is_bit_identical_to_the_plain_kernel(sdf, jac):
    """Batches >= 65536 points take the XCD-partitioned forward (each XCD owns a group of levels); smaller ones the plain
    kernel.  Same arithmetic per (point, level): features, Jacobian-based d/dx and table gradients must agree exactly."""
    dev = torch.device("cuda:0")
    B = 200_003
    x0 = torch.rand(B, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    enc = sdf.TCNNEncoding(3, None, "enc", dev, seed=3)
    with torch.no_grad():
        enc.params_.mul_(1e3)
    enc.save_jacobian = jac
    v = torch.randn(B, 32, generator=torch.Generator().manual_seed(6)).to(dev)
    xb = x0.clone().requires_grad_(True)
    big = enc.forward(xb)
    gx_big = torch.autograd.grad((big * v).sum(), xb)[0]
    parts, gparts = [], []
    for s in range(0, B, 50_000):
        xs = x0[s:s + 50_000].clone().requires_grad_(True)
        f = enc.forward(xs)
        parts.append(f.detach())
        gparts.append(torch.autograd.grad((f * v[s:s + 50_000]).sum(), xs)[0])
    assert torch.equal(big.detach(), torch.cat(parts))
    assert torch.equal(gx_big, torch.cat(gparts))


@pytest.mark.parametrize("cfg,B,concentrated", [(CFG, 70000, False), (CFG, 70000, True), (CFG_SMALL, 3000, False),
                                                (dict(CFG, n_levels=5), 1, False)])
def test_hashgrid_binned_scatter_matches_atomic_and_oracle(sdf, oracle, cfg, B, concentrated):
    """gsdf_hashgrid_bwd_binned (count -> plan -> emit -> apply, no global atomics) against the atomic kernel and the
    oracle.  `concentrated`: every point in ONE cell of the coarsest level, so that every level's buckets receive many times
    the records of a work item (32 K at this batch size) and are split: the items of a bucket meet in a 64-bit global tile by
    integer atomics, so the split buckets are bit-reproducible too (round 6; the uniform case splits level 0's nine tiles)."""
    import ctypes as C
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, generator=g)
    if concentrated:
        x = 0.501 + 0.01 * x
    if B >= 6 and not concentrated:
        x[:6] = torch.tensor([0.0, 1.0, 0.5, 0.25, 1.0 - 2 ** -20, 2 ** -21])[:, None]
    nf = cfg["n_levels"] * 2
    v = torch.randn(B, nf, generator=g)
    offs, total = oracle.grid_offsets(cfg)
    c = (cfg["n_levels"], 2, cfg["log2_hashmap"], cfg["base_res"], cfg["per_level_scale"])
    L = capi.lib()
    nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(B, *c)
    assert nbytes > 0
    xd, vd = x.to(dev), v.to(dev)
    table = torch.zeros(total, 2, device=dev)
    seed = (torch.randn(total, 2, generator=g) * (1e-3 if B < 100 else 1.0)).to(dev)     # the call ACCUMULATES into v_table
    got = seed.clone()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    capi.check(L.gsdf_hashgrid_bwd_binned(B, *c, capi.f32(xd), capi.f32(vd), capi.f32(got), capi.ptr(ws), nbytes, capi.stream()), "binned")
    atomic = seed.clone()
    capi.check(L.gsdf_hashgrid_bwd(B, *c, capi.f32(xd), capi.f32(table), capi.f32(vd), capi.f32(atomic), None, capi.stream()), "atomic")
    torch.cuda.synchronize()
    # the atomic kernel sums in fp32 in arrival order: with ~5e5 contributions of random sign per entry (concentrated)
    # its own rounding noise is ~1e-4 of the tensor's mean magnitude
    assert_close(got - seed, atomic - seed, 1e-3 if concentrated else 1e-4, "binned vs atomic scatter")
    vt_o, _ = oracle.grid_bwd(n(x), n(table.cpu()), n(v), cfg, prec="f32")     # oracle accumulates in double
    # Round 4: 8-byte records.  A contribution enters the (exact, order-independent) fixed-point sum rounded to 22 bits relative to the larger
    # value of its record, so an entry's error is bounded by 2^-23 x the sum of |contributions| of BOTH features of the entry (+ the final
    # fp32 rounding): the oracle evaluated on |v| gives that sum.  Element-wise: within the 1e-4 bar AND within 4 x that bound.
    vt_abs, _ = oracle.grid_bwd(n(x), n(table.cpu()), np.abs(n(v)), cfg, prec="f32")
    bound = 4.0 * 2.0 ** -23 * vt_abs.sum(1, keepdims=True) + 2.0 ** -22 * np.abs(vt_o) + 1e-30
    # + the resolution of the fixed point itself: 2^-38 of the level's largest |v_feat| per record (a contribution below it is dropped, as it
    # was at 2^-41 with the 12-byte records): 2^-28 of the level maximum covers a thousand records per entry
    vn = np.abs(n(v)).reshape(B, cfg["n_levels"], 2).max(axis=(0, 2))
    for l in range(cfg["n_levels"]):
        bound[offs[l]:offs[l + 1]] += 2.0 ** -28 * vn[l]

    def within_record_rounding(t, what, extra=0.0):
        e = np.abs(n(t).astype(np.float64) - vt_o)
        worst = float((e / (bound + extra)).max())
        assert worst <= 1.0, f"{what}: an entry is {worst:.2f} x the bound of the records' rounding"
    assert_close(got - seed, vt_o, 1e-4 if B >= 100 else 1e-3, "binned scatter (accumulated onto a non-zero buffer) vs oracle")
    # a second call on the same workspace (stale counters / cursors must not leak)
    got2 = torch.zeros_like(seed)
    capi.check(L.gsdf_hashgrid_bwd_binned(B, *c, capi.f32(xd), capi.f32(vd), capi.f32(got2), capi.ptr(ws), nbytes, capi.stream()), "binned")
    torch.cuda.synchronize()
    assert_close(got2, vt_o, 1e-4, "binned scatter, reused workspace")
    within_record_rounding(got2, "binned scatter, reused workspace")
    # whole and split buckets alike: every sum is an integer sum, whatever the order of arrival
    got3 = torch.zeros_like(seed)
    capi.check(L.gsdf_hashgrid_bwd_binned(B, *c, capi.f32(xd), capi.f32(vd), capi.f32(got3), capi.ptr(ws), nbytes, capi.stream()), "binned")
    torch.cuda.synchronize()
    assert torch.equal(got2, got3), "binned scatter is not bit-reproducible"


def test_encoder_backward_takes_binned_path_for_large_batches(sdf, oracle, monkeypatch):
    dev = torch.device("cuda:0")
    enc = sdf.TCNNEncoding(3, tcfg(CFG), "enc", dev, seed=1)
    g = torch.Generator().manual_seed(5)
    B = sdf.BINNED_MIN_POINTS + 123
    x = torch.rand(B, 3, generator=g).to(dev)
    v = torch.randn(B, 32, generator=g).to(dev)
    grads = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GSDF_HASHGRID_BINNED", mode)
        enc.params_.grad = None
        enc.forward(x).backward(v)
        grads[mode] = enc.params_.grad.clone()
    assert_close(grads["1"], grads["0"], 1e-4, "autograd table gradient: binned vs atomic")   # (8-byte records: 22-bit contributions)


@pytest.mark.parametrize("n,delta", [(12000, 0.02 / 16.0), (12000, 0.3), (5000, 1e-5)])
def test_binned_scatter_with_stencil_merging_matches_oracle(sdf, oracle, n, delta):
    """gsdf_hashgrid_bwd_binned_stencil: rows = n base points + their 6 central-difference points; at the coarse levels the rows
    of a group that share the base point's cell are summed in registers before they become records.  delta = 0.00125 (the
    joint iteration's: cells shared at levels 0-4), 0.3 (never shared), 1e-5 (shared at every level)."""
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(13)
    base = torch.rand(n, 3, generator=g) * 0.6 + 0.2
    offs = torch.tensor([[0, 0, 0], [delta, 0, 0], [-delta, 0, 0], [0, delta, 0], [0, -delta, 0], [0, 0, delta], [0, 0, -delta]])
    x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous()
    B = x.shape[0]
    v = torch.randn(B, 32, generator=g)
    c = (16, 2, 19, 32, 2.0)
    offs_l, total = oracle.grid_offsets(CFG)
    L = capi.lib()
    nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(B, *c)
    xd, vd = x.to(dev), v.to(dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ml = sdf.stencil_merge_levels(c, delta)
    assert ml == {0.02 / 16.0: 5, 0.3: 0, 1e-5: 12}[delta]
    got = torch.zeros(total, 2, device=dev)
    capi.check(L.gsdf_hashgrid_bwd_binned_stencil(B, n, 16, *c, capi.f32(xd), capi.f32(vd), capi.f32(got), capi.ptr(ws), nbytes, capi.stream()), "stencil")
    plain = torch.zeros(total, 2, device=dev)
    capi.check(L.gsdf_hashgrid_bwd_binned(B, *c, capi.f32(xd), capi.f32(vd), capi.f32(plain), capi.ptr(ws), nbytes, capi.stream()), "plain")
    torch.cuda.synchronize()
    vt_o, _ = oracle.grid_bwd(n(x) if False else x.numpy(), np.zeros((total, 2), np.float32), v.numpy(), CFG, prec="f32")
    # merged groups are summed in fp32 (<= 7 products of random sign) before the exact fixed-point accumulation: where they
    # cancel, the element's error relative to ITSELF reaches a few 1e-5 (sparse table: the mean |ref| floor is tiny)
    assert_close(got, vt_o, 5e-5, "stencil-merged binned scatter vs oracle")
    assert_close(got, plain, 5e-5, "stencil-merged vs plain binned scatter")
    again = torch.zeros(total, 2, device=dev)
    capi.check(L.gsdf_hashgrid_bwd_binned_stencil(B, n, 16, *c, capi.f32(xd), capi.f32(vd), capi.f32(again), capi.ptr(ws), nbytes, capi.stream()), "stencil")
    torch.cuda.synchronize()
    assert torch.equal(got, again), "not bit-reproducible"
    with pytest.raises(RuntimeError):
        capi.check(L.gsdf_hashgrid_bwd_binned_stencil(B, n + 1, 5, *c, capi.f32(xd), capi.f32(vd), capi.f32(again), capi.ptr(ws), nbytes, capi.stream()), "bad")


@pytest.mark.parametrize("n,delta,jac", [(20011, 0.02 / 16.0, True), (20011, 0.02 / 16.0, False), (777, 0.3, True), (5000, 1e-5, True),
                                         (1, 0.01, True)])
def test_stencil_forward_is_bit_identical_to_the_row_major_kernels(sdf, n, delta, jac):
    """gsdf_hashgrid_fwd_stencil walks the 7 rows of a group with the same lanes (register reuse where the rows share the base
    row's cell, cache reuse elsewhere); per row the arithmetic is the row-major kernels': features and the base rows' Jacobian
    must agree bit for bit.  7 * 20011 rows take the XCD-partitioned launch, the others the single-queue one."""
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(29)
    base = torch.rand(n, 3, generator=g) * 0.6 + 0.2
    offs = torch.tensor([[0, 0, 0], [delta, 0, 0], [-delta, 0, 0], [0, delta, 0], [0, -delta, 0], [0, 0, delta], [0, 0, -delta]])
    x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous().to(dev)
    B = x.shape[0]
    c = (16, 2, 19, 32, 2.0)
    enc = sdf.TCNNEncoding(3, None, "enc", dev, seed=3)
    table = (enc.params_.detach() * 1e3).contiguous()
    L = capi.lib()
    ref, got = torch.empty(B, 32, device=dev), torch.zeros(B, 32, device=dev)
    jref, jgot = torch.empty(n, 32, 3, device=dev), torch.zeros(n, 32, 3, device=dev)
    capi.check(L.gsdf_hashgrid_fwd_jac_rows(B, n if jac else 0, *c, capi.f32(x), capi.f32(table), capi.f32(ref), capi.f32(jref), capi.stream()), "rows")
    capi.check(L.gsdf_hashgrid_fwd_stencil(B, n, n if jac else 0, *c, capi.f32(x), capi.f32(table), capi.f32(got),
                                           capi.f32(jgot) if jac else None, capi.stream()), "stencil")
    torch.cuda.synchronize()
    assert torch.equal(ref, got)
    if jac:
        assert torch.equal(jref, jgot)
    with pytest.raises(RuntimeError):
        capi.check(L.gsdf_hashgrid_fwd_stencil(B, n + 1, 0, *c, capi.f32(x), capi.f32(table), capi.f32(got), None, capi.stream()), "bad")
    with pytest.raises(RuntimeError):
        capi.check(L.gsdf_hashgrid_fwd_stencil(B, n, 3, *c, capi.f32(x), capi.f32(table), capi.f32(got), capi.f32(jgot), capi.stream()), "bad")


def test_sdf_leg_at_the_joint_iteration_size(sdf, oracle):
    """The joint iteration pushes 7 x (visible splat samples) ~ 3 M rows through encoder and decoder in one launch each
    (neural_mapping.cpp:436-457): the persistent kernels then walk ~100 tiles per wave, the one-pass decoder backward keeps
    its weight-gradient tiles in registers over all of them, the scatter takes the binned path with stencil merging.
    Checked at 7 x 300 000 rows: (a) features, decoder outputs and per-row input gradients of 20 000 sampled rows against the
    oracle (rows are independent); (b) parameter gradients by linearity — the launch over all rows against the sum of four
    launches over its quarters (different tiles per wave, different flush grouping, different fixed-point scales)."""
    import ctypes as C
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    L = capi.lib()
    n_grp, delta = 300_000, 0.02 / 16.0
    g = torch.Generator().manual_seed(41)
    base = torch.rand(n_grp, 3, generator=g) * 0.8 + 0.1
    offs = torch.tensor([[0, 0, 0], [delta, 0, 0], [-delta, 0, 0], [0, delta, 0], [0, -delta, 0], [0, 0, delta], [0, 0, -delta]])
    x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous()
    B = x.shape[0]
    c = (16, 2, 19, 32, 2.0)
    _, total = oracle.grid_offsets(CFG)
    table = (torch.rand(total, 2, generator=g) * 2 - 1) * 0.5
    dims = [32, 64, 64, 64, 2]
    nl, dims_c = len(dims) - 1, (C.c_int * len(dims))(*dims)
    W = torch.cat([(torch.rand(o * i, generator=g) * 2 - 1) * (6.0 / i) ** 0.5 for i, o in zip(dims[:-1], dims[1:])])
    xd, td, Wd = x.to(dev), table.to(dev), W.to(dev)
    feat = torch.empty(B, 32, device=dev)
    capi.check(L.gsdf_hashgrid_fwd_stencil(B, n_grp, 0, *c, capi.f32(xd), capi.f32(td), capi.f32(feat), None, capi.stream()), "fwd")
    out = torch.empty(B, 2, device=dev)
    acts = torch.empty(L.gsdf_mlp_acts_floats(B, nl), device=dev)
    capi.check(L.gsdf_mlp_fwd(B, nl, dims_c, capi.f32(Wd), None, capi.f32(feat), capi.f32(out), capi.f32(acts), capi.stream()), "mlp fwd")
    v_out = torch.randn(B, 2, generator=g).to(dev)
    v_feat, v_w = torch.empty_like(feat), torch.zeros_like(Wd)
    capi.check(L.gsdf_mlp_bwd(B, nl, dims_c, capi.f32(Wd), None, capi.f32(feat), capi.f32(acts), capi.f32(v_out), capi.f32(v_feat),
                              capi.f32(v_w), None, None, capi.stream()), "mlp bwd")
    nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(B, *c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    v_t = torch.zeros(total, 2, device=dev)
    ml = sdf.stencil_merge_levels(c, delta)
    capi.check(L.gsdf_hashgrid_bwd_binned_stencil(B, n_grp, ml, *c, capi.f32(xd), capi.f32(v_feat), capi.f32(v_t), capi.ptr(ws), nbytes, capi.stream()), "scatter")
    torch.cuda.synchronize()
    # (a) sampled rows
    idx = torch.randperm(B, generator=g)[:20000].sort().values
    f_s = n(feat[idx.to(dev)])
    assert_close(f_s, oracle.grid_fwd(n(x[idx]), n(table), CFG, prec="f32"), 1e-5, "features of the sampled rows")
    keep = ~_near_relu_kink(f_s, dims, n(W), None)
    idk = idx[torch.from_numpy(keep)]
    f_k = f_s[keep]
    assert_close(out[idk.to(dev)], oracle.mlp_fwd(f_k, dims, n(W), None, prec="f64"), REL, "decoder outputs of the sampled rows")
    vin_o, _, _ = oracle.mlp_bwd(f_k, dims, n(W), None, n(v_out[idk.to(dev)]), prec="f64")
    assert_close(v_feat[idk.to(dev)], vin_o, REL, "decoder input gradients of the sampled rows")
    # (b) linearity of the parameter gradients over the rows (quarters = whole groups: 7 x n_grp/4 rows each, re-stacked)
    v_w_q, v_t_q = torch.zeros_like(Wd), torch.zeros(total, 2, device=dev)
    q = n_grp // 4
    for k in range(4):
        rows = torch.cat([torch.arange(r * n_grp + k * q, r * n_grp + (k + 1) * q) for r in range(7)]).to(dev)
        Bq = rows.numel()
        fq, vq, xq = feat[rows].contiguous(), v_out[rows].contiguous(), xd[rows].contiguous()
        oq, aq = torch.empty(Bq, 2, device=dev), torch.empty(L.gsdf_mlp_acts_floats(Bq, nl), device=dev)
        capi.check(L.gsdf_mlp_fwd(Bq, nl, dims_c, capi.f32(Wd), None, capi.f32(fq), capi.f32(oq), capi.f32(aq), capi.stream()), "mlp fwd")
        vfq = torch.empty_like(fq)
        capi.check(L.gsdf_mlp_bwd(Bq, nl, dims_c, capi.f32(Wd), None, capi.f32(fq), capi.f32(aq), capi.f32(vq), capi.f32(vfq),
                                  capi.f32(v_w_q), None, None, capi.stream()), "mlp bwd")
        assert torch.equal(vfq, v_feat[rows]), "per-row input gradients depend on the launch size"
        capi.check(L.gsdf_hashgrid_bwd_binned_stencil(Bq, q, ml, *c, capi.f32(xq), capi.f32(vfq), capi.f32(v_t_q), capi.ptr(ws), nbytes,
                                                      capi.stream()), "scatter")
    torch.cuda.synchronize()
    assert_close(v_w, v_w_q, REL, "decoder weight gradients: all rows vs the sum over quarters")
    assert_close(v_t, v_t_q, REL, "table gradient: all rows vs the sum over quarters")


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_hashgrid_binned_scatter_propagates_non_finite_gradients(sdf, oracle, bad):
    """A NaN / Inf upstream gradient must not become finite garbage (fmaxf drops NaN, the fixed-point conversion of a non-finite value is
    arbitrary): the level it belongs to comes out non-finite where it was touched, the other levels are exact."""
    import gs_sdf_amd.capi as capi
    dev = torch.device("cuda:0")
    B, cfg = 40000, CFG
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, generator=g)
    v = torch.randn(B, cfg["n_levels"] * 2, generator=g)
    v[1234, 2 * 7 + 1] = bad                                   # one feature of level 7 of one point
    offs, total = oracle.grid_offsets(cfg)
    c = (cfg["n_levels"], 2, cfg["log2_hashmap"], cfg["base_res"], cfg["per_level_scale"])
    L = capi.lib()
    nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(B, *c)
    got = torch.zeros(total, 2, device=dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    capi.check(L.gsdf_hashgrid_bwd_binned(B, *c, capi.f32(x.to(dev)), capi.f32(v.to(dev)), capi.f32(got), capi.ptr(ws), nbytes, capi.stream()), "binned")
    torch.cuda.synchronize()
    lvl = got[offs[7]:offs[8]]
    assert not bool(torch.isfinite(lvl).all()), "the non-finite contribution vanished"
    clean = v.clone()
    clean[1234, 2 * 7 + 1] = 0.0
    ref, _ = oracle.grid_bwd(n(x), np.zeros((total, 2), np.float32), n(clean), cfg, prec="f32")
    for l in (0, 6, 8, 15):
        assert_close(got[offs[l]:offs[l + 1]], ref[offs[l]:offs[l + 1]], 1e-5, f"level {l} beside the poisoned one")
