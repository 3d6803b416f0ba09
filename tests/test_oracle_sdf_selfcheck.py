"""Self-check of the SDF oracle (parity unpinned, see oracle/README.md): an independent torch-fp64
restatement of the multiresolution hash encoding + MLP, differentiated by torch.autograd (first AND
second order), must agree with the C oracle's forward, backward and double-backward."""
import numpy as np
import torch

CFG_SMALL = dict(n_levels=6, n_feat=2, log2_hashmap=10, base_res=4, per_level_scale=2.0)
PRIMES = (1, 2654435761, 805459861)


def torch_grid(x, table, offsets, cfg):
    """Direct restatement: per level trilinear blend of 8 corner features (dense index while the stride
    fits the level's table, coherent prime hash otherwise)."""
    feats = []
    for l in range(cfg["n_levels"]):
        scale = np.float32(np.exp2(np.float32(l) * np.log2(np.float32(cfg["per_level_scale"]))) * np.float32(cfg["base_res"]) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        hsize = int(offsets[l + 1] - offsets[l])
        pos = x * float(scale) + 0.5
        fl = torch.floor(pos.detach())
        fr = pos - fl
        g0 = fl.to(torch.int64)
        acc = 0
        for k in range(8):
            bits = [(k >> d) & 1 for d in range(3)]
            g = [(g0[:, d] + bits[d]) & 0xFFFFFFFF for d in range(3)]
            w = 1
            for d in range(3):
                w = w * (fr[:, d] if bits[d] else 1 - fr[:, d])
            stride, idx, dense = 1, torch.zeros_like(g[0]), True
            for d in range(3):
                if stride <= hsize:
                    idx = (idx + g[d] * stride) & 0xFFFFFFFF
                    stride = (stride * res) & 0xFFFFFFFF if stride * res < 2 ** 32 else stride * res
            if hsize < stride:
                idx = ((g[0] * PRIMES[0]) & 0xFFFFFFFF) ^ ((g[1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((g[2] * PRIMES[2]) & 0xFFFFFFFF)
            idx = idx % hsize
            acc = acc + w[:, None] * table[int(offsets[l]) + idx]
        feats.append(acc)
    return torch.cat(feats, -1)


def test_grid_offsets_match_reference_config(oracle):
    offs, total = oracle.grid_offsets()
    # levels 0,1 dense (32^3, 64^3), the rest hashed into 2^19 entries: 15.27 M parameters
    assert list(np.diff(offs)[:3]) == [32 ** 3, 64 ** 3, 2 ** 19] and total == 32 ** 3 + 64 ** 3 + 14 * 2 ** 19
    assert total * 2 == 15_269_888


def test_grid_fwd_bwd_bwdbwd_match_autograd(oracle):
    cfg = CFG_SMALL
    offs, total = oracle.grid_offsets(cfg)
    g = torch.Generator().manual_seed(0)
    B = 200
    x = torch.rand(B, 3, generator=g, dtype=torch.float64)
    x[:5] = torch.tensor([0.0, 1.0, 0.5, 0.25, 0.999])[:, None]            # cell boundaries / domain edges
    table = (torch.rand(total, 2, generator=g, dtype=torch.float64) * 2 - 1)
    xa, ta = x.clone().requires_grad_(True), table.clone().requires_grad_(True)
    feat = torch_grid(xa, ta, offs, cfg)
    f_o, jac = oracle.grid_fwd(x.numpy(), table.numpy(), cfg, want_jac=True, prec="f64")
    np.testing.assert_allclose(f_o, feat.detach().numpy(), rtol=1e-12, atol=1e-14)
    v = torch.randn(B, feat.shape[1], generator=g, dtype=torch.float64).requires_grad_(True)
    v_x, v_t = torch.autograd.grad(feat, (xa, ta), v, create_graph=True)
    vt_o, vx_o = oracle.grid_bwd(x.numpy(), table.numpy(), v.detach().numpy(), cfg, prec="f64")
    np.testing.assert_allclose(vx_o, v_x.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vt_o, v_t.detach().numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(np.einsum("bfd,bf->bd", jac, v.detach().numpy()), vx_o, rtol=1e-10, atol=1e-12)
    # double backward: loss2 = <vv, v_x>
    vv = torch.randn(B, 3, generator=g, dtype=torch.float64)
    vv.requires_grad_(True)
    g_v, g_t, g_x = torch.autograd.grad((v_x * vv).sum(), (v, ta, xa), create_graph=True)
    gv_o, gt_o, gx_o = oracle.grid_bwd_bwd(x.numpy(), table.numpy(), v.detach().numpy(), vv.detach().numpy(), cfg, prec="f64")
    np.testing.assert_allclose(gv_o, g_v.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gt_o, g_t.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gx_o, g_x.detach().numpy(), rtol=1e-9, atol=1e-10)
    # third order (a loss on the analytic Hessian, local_map.cpp:163-168): loss3 = <lam, g_x> + <mu, g_v>
    lam = torch.randn(B, 3, generator=g, dtype=torch.float64)
    mu = torch.randn(B, feat.shape[1], generator=g, dtype=torch.float64)
    for use_mu in (False, True):
        loss3 = (g_x * lam).sum() + ((g_v * mu).sum() if use_mu else 0.0)
        t_v, t_t, t_vv, t_x = torch.autograd.grad(loss3, (v, ta, vv, xa), retain_graph=True)
        o_v, o_t, o_vv, o_x = oracle.grid_bwd3(x.numpy(), table.numpy(), v.detach().numpy(), vv.detach().numpy(), lam.numpy(),
                                               mu.numpy() if use_mu else None, cfg, prec="f64")
        np.testing.assert_allclose(o_v, t_v.numpy(), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o_t, t_t.numpy(), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o_vv, t_vv.numpy(), rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(o_x, t_x.numpy(), rtol=1e-8, atol=1e-6)


def test_mlp_and_head_match_torch(oracle):
    g = torch.Generator().manual_seed(1)
    dims = [32, 64, 64, 64, 64, 2]                       # reference decoder, local_map.cpp:29-42
    B = 300
    layers = []
    for i, o in zip(dims[:-1], dims[1:]):
        lin = torch.nn.Linear(i, o).double()
        layers += [lin, torch.nn.ReLU()]
    net = torch.nn.Sequential(*layers[:-1])
    W = torch.cat([m.weight.detach().reshape(-1) for m in net if isinstance(m, torch.nn.Linear)])
    b = torch.cat([m.bias.detach() for m in net if isinstance(m, torch.nn.Linear)])
    x = torch.randn(B, 32, generator=g, dtype=torch.float64).requires_grad_(True)
    y = net(x)
    out, acts = oracle.mlp_fwd(x.detach().numpy(), dims, W.numpy(), b.numpy(), want_acts=True, prec="f64")
    np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-11, atol=1e-13)
    assert acts.shape == (B, 256) and (acts >= 0).all()
    v = torch.randn(B, 2, generator=g, dtype=torch.float64)
    y.backward(v)
    v_in, v_w, v_b = oracle.mlp_bwd(x.detach().numpy(), dims, W.numpy(), b.numpy(), v.numpy(), prec="f64")
    np.testing.assert_allclose(v_in, x.grad.numpy(), rtol=1e-10, atol=1e-12)
    gw = torch.cat([m.weight.grad.reshape(-1) for m in net if isinstance(m, torch.nn.Linear)])
    gb = torch.cat([m.bias.grad for m in net if isinstance(m, torch.nn.Linear)])
    np.testing.assert_allclose(v_w, gw.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v_b, gb.numpy(), rtol=1e-10, atol=1e-12)
    # bias-free (tcnn FullyFusedMLP) topology
    dims2 = [32, 64, 64, 64, 2]
    W2 = torch.randn(sum(i * o for i, o in zip(dims2[:-1], dims2[1:])), generator=g, dtype=torch.float64) * 0.2
    out2 = oracle.mlp_fwd(x.detach().numpy(), dims2, W2.numpy(), None, prec="f64")
    h, off = x.detach(), 0
    for li, (i, o) in enumerate(zip(dims2[:-1], dims2[1:])):
        h = h @ W2[off:off + i * o].reshape(o, i).T
        off += i * o
        if li < len(dims2) - 2:
            h = torch.relu(h)
    np.testing.assert_allclose(out2, h.numpy(), rtol=1e-11, atol=1e-13)
    # head: isigma = 1 + softplus_beta100(raw)/bce_sigma  (local_map.cpp:100-102)
    raw = torch.randn(B, 2, generator=g, dtype=torch.float64)
    sdf, isig = oracle.sdf_head(raw.numpy(), 1.0 / 0.02, prec="f64")
    ref = 1 + torch.nn.functional.softplus(raw[:, 1], beta=100) / 0.02
    np.testing.assert_allclose(sdf, raw[:, 0].numpy())
    np.testing.assert_allclose(isig, ref.numpy(), rtol=1e-12)


def test_mlp_double_backward_matches_autograd(oracle):
    """orc_mlp_bwd_bwd (the decoder's part of LocalMap::get_gradient's analytic branch, local_map.cpp:151-172: autograd::grad with
    create_graph, then the eikonal loss differentiated again) against torch fp64 autograd, biased and bias-free topologies."""
    g = torch.Generator().manual_seed(7)
    for dims, bias in (([32, 64, 64, 64, 64, 2], True), ([32, 64, 64, 64, 2], False)):
        B = 257
        nw = sum(i * o for i, o in zip(dims[:-1], dims[1:]))
        W = (torch.randn(nw, generator=g, dtype=torch.float64) * 0.25).requires_grad_(True)
        b = (torch.randn(sum(dims[1:]), generator=g, dtype=torch.float64) * 0.1) if bias else None
        x = torch.randn(B, dims[0], generator=g, dtype=torch.float64).requires_grad_(True)
        v_out = torch.randn(B, dims[-1], generator=g, dtype=torch.float64).requires_grad_(True)
        vv_in = torch.randn(B, dims[0], generator=g, dtype=torch.float64)
        h, off, boff = x, 0, 0
        for li, (i, o) in enumerate(zip(dims[:-1], dims[1:])):
            h = h @ W[off:off + i * o].reshape(o, i).T
            if bias:
                h = h + b[boff:boff + o]
            off, boff = off + i * o, boff + o
            if li < len(dims) - 2:
                h = torch.relu(h)
        (v_in,) = torch.autograd.grad(h, x, v_out, create_graph=True)         # first backward, differentiable
        g_vout, g_w = torch.autograd.grad(v_in, [v_out, W], vv_in)             # second: d<v_in, vv_in> / d(v_out, W)
        got_vout, got_w = oracle.mlp_bwd_bwd(x.detach().numpy(), dims, W.detach().numpy(), None if b is None else b.numpy(),
                                             v_out.detach().numpy(), vv_in.numpy(), prec="f64")
        np.testing.assert_allclose(got_vout, g_vout.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(got_w, g_w.numpy(), rtol=1e-10, atol=1e-12)


def test_knn_matches_cdist(oracle):
    g = torch.Generator().manual_seed(2)
    pts = torch.rand(500, 3, generator=g, dtype=torch.float64)
    d2 = torch.cdist(pts, pts) ** 2
    d2.fill_diagonal_(float("inf"))
    ref = d2.topk(3, largest=False).values.mean(1)
    np.testing.assert_allclose(oracle.knn_mean_dist2(pts.numpy(), prec="f64"), ref.numpy(), rtol=1e-10)
