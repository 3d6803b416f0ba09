"""End-to-end slice on the GPU: NeuralGS.render (-> C ABI) inside a short joint optimisation with the reference's
callback schedule (densification every few iterations) and the GS->SDF coupling term."""
import pytest
import torch

import gs_sdf_amd.synth as synth

pytestmark = pytest.mark.gpu


def test_render_keys_and_short_training_run():
    from gs_sdf_amd.neural_gs import Cameras, GSConfig, NeuralGS
    import gs_sdf_amd.sdf as sdfm
    dev = torch.device("cuda:0")
    W, H, N = 320, 192, 6000
    sc = synth.make_scene(N, W, H, sh_degree=1, seed=3)
    K = sc["K"][0]
    cam = Cameras(float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H)
    cfg = GSConfig(sh_degree=1, refine_start_iter=14, refine_every=4, reset_every=1000, grow_grad2d=1e-7, center_reg=True)
    gs = NeuralGS(sc["means"].to(dev), sc["log_scales"].to(dev), sc["quats"].to(dev), sc["logit_opacities"].to(dev),
                  sc["sh"][:, :1].to(dev), sc["sh"][:, 1:].to(dev), cfg, spatial_scale=1.0, num_train_data=4)
    lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, decoder_implementation=1, device=dev, seed=1)
    opt = gs.make_optimizer([dict(params=lm.parameters(), lr=5e-3)])
    assert gs.gs_param_start_idx == 1
    poses = [torch.linalg.inv(v)[:3, :4] for v in synth.make_views(4, seed=2)]          # cam2world [3,4]
    with torch.no_grad():
        target = [gs.render(p, cam)["color"].detach() * 0.5 + 0.25 for p in poses]     # a different image to fit
    out = gs.render(poses[0], cam)
    for k in ("color", "depth", "alpha", "render_normal", "render_median", "normal", "gaussian_ids", "radii", "gradient_2dgs",
              "width", "height", "n_cameras", "samples", "samples_weights", "samples_opacities", "visibilities", "xyz"):
        assert k in out, k                                                             # neural_gaussian.cpp:245-267,556-560
    assert out["color"].shape == (H, W, 3) and out["depth"].shape == (H, W, 1) and out["alpha"].shape == (1, H, W, 1)
    losses, sizes = [], []
    for it in range(1, 25):
        opt.zero_grad()
        r = gs.render(poses[it % 4], cam)
        loss = (r["color"] - target[it % 4]).abs().mean()
        vis = r["visibilities"].detach()
        ids = (vis > 0.1).squeeze(-1).nonzero().squeeze(-1)
        if ids.numel():
            s = lm.get_sdf(r["samples"].index_select(0, ids))[0]
            loss = loss + 1e-3 * sdfm.gs_sdf_loss(s, (r["samples_weights"] * vis).detach().index_select(0, ids)) / ids.numel()
        loss.backward()
        opt.step()
        log = gs.train_callback(it, 100, opt, r)
        losses.append(float(loss)); sizes.append(gs.anchors_.shape[0])
        for p in gs.PARAMS:
            t = getattr(gs, p)
            # split children carry log(0) = -inf in the unused third scale, exactly like the reference
            # (neural_gaussian.cpp:771-790: scales[:,2] = 0 then log(scales / 1.6))
            assert torch.isfinite(t[:, :2] if p == "scaling_" else t).all(), p
    assert len(set(sizes[14:])) > 1, "densification never changed the number of splats"
    assert len(set(sizes[:13])) == 1
    # before the first refinement the parameters are only moved by Adam: the loss must go down
    assert sum(losses[8:12]) < 0.9 * sum(losses[:4]), (losses[:4], losses[8:12])
    assert lm.encoder.params_.grad is not None and float(lm.encoder.params_.grad.abs().sum()) > 0


def test_init_gs_with_sdf_orients_splats_along_the_sdf_normal():
    """On an analytic SDF (a plane + mild curvature, served through the LocalMap interface) the initial quaternion must
    rotate the splat's local z axis onto the SDF gradient (neural_gaussian.cpp:19-127)."""
    from gs_sdf_amd.neural_gs import init_gs_with_sdf, normalized_quat_to_rotmat
    dev = torch.device("cuda:0")
    nrm = torch.nn.functional.normalize(torch.tensor([0.3, -0.5, 0.8], device=dev), dim=0)

    class AnalyticMap:
        def get_sdf(self, x):
            s = (x * nrm).sum(-1, keepdim=True) - 0.2 + 0.05 * (x[:, 0:1] ** 2)
            return [s, torch.full_like(s, 50.0)]

        def get_gradient(self, x, delta, sdf=None, hessian=False, numerical_grad=True):
            offs = torch.eye(3, device=x.device) * delta
            sp = torch.stack([self.get_sdf(x + offs[d])[0] for d in range(3)], 0)
            sm = torch.stack([self.get_sdf(x - offs[d])[0] for d in range(3)], 0)
            g = 0.5 / delta * torch.cat([sp[0] - sm[0], sp[1] - sm[1], sp[2] - sm[2]], 1)
            h = torch.cat([sp[d] + sm[d] - 2 * self.get_sdf(x)[0] for d in range(3)], 1) / delta ** 2
            return [g, h]

    x = torch.rand(5000, 3, device=dev) - 0.5
    out = init_gs_with_sdf(AnalyticMap(), x, 0.01, init_opa=True, batch_size=2048)
    R = normalized_quat_to_rotmat(out["quaternion"])
    z_axis = R[:, :, 2]
    g = torch.nn.functional.normalize(out["grad"], dim=-1)
    assert float((z_axis * g).sum(-1).abs().min()) > 0.999
    assert torch.allclose(out["quaternion"].norm(dim=-1), torch.ones(5000, device=dev), atol=1e-4)
    s = AnalyticMap().get_sdf(x)[0]
    assert torch.allclose(out["opacity"], torch.exp(-s.square() * 50.0).squeeze(-1))


def test_flat_neural_gs_schedule_matches_neural_gs_with_torch_adam():
    """The fast path (FlatNeuralGS: flat parameter buffer, fused activations, FusedAdam, fused update_state, row-gather
    surgery) through 30 iterations of the reference's schedule — refinement (duplicate / split / prune) every 6 iterations
    from iteration 7, an opacity reset at 18 — against NeuralGS + torch.optim.Adam on the same views: the same number of
    splats after every iteration and the same parameters at the end."""
    from gs_sdf_amd.neural_gs import Cameras, FlatNeuralGS, GSConfig, NeuralGS
    import gs_sdf_amd.ops as ops
    dev = torch.device("cuda:0")
    W, H, N = 320, 192, 6000
    sc = synth.make_scene(N, W, H, sh_degree=1, seed=3)
    K = sc["K"][0]
    cam = Cameras(float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H)
    cfg = GSConfig(sh_degree=1, refine_start_iter=6, refine_every=6, reset_every=18, grow_grad2d=2e-7, sh_degree_interval=10)
    args = (sc["means"].to(dev), sc["log_scales"].to(dev), sc["quats"].to(dev), sc["logit_opacities"].to(dev),
            sc["sh"][:, :1].to(dev), sc["sh"][:, 1:].to(dev))
    poses = [torch.linalg.inv(v)[:3, :4] for v in synth.make_views(4, seed=2)]
    models = [NeuralGS(*args, cfg, 1.0, 4), FlatNeuralGS(*args, cfg, 1.0, 4)]
    with torch.no_grad():
        target = [models[0].render(p, cam)["color"].detach() * 0.5 + 0.25 for p in poses]
    sizes = [[], []]
    for k, gs in enumerate(models):
        opt = gs.make_optimizer()
        torch.manual_seed(0)                                   # split() draws from the global generator, like the reference
        for it in range(1, 31):
            if k == 0:
                opt.zero_grad()
            else:
                gs.params.flat_grad.zero_()
            r = gs.render(poses[it % 4], cam, training=True)
            ops.l1_dssim_loss(r["color"], target[it % 4], 0.8, 0.2).backward()
            opt.step()
            gs.train_callback(it, 100, opt, r)
            sizes[k].append(int(gs.anchors_.shape[0]))
    assert sizes[0] == sizes[1], (sizes[0], sizes[1])
    assert len(set(sizes[0])) > 2, "the schedule never changed the number of splats"
    a, b = models
    assert torch.equal(a.anchors_, b.anchors_)
    for p in NeuralGS.PARAMS:
        x, y = getattr(a, p).detach(), getattr(b, p).detach()
        fin = torch.isfinite(x)
        assert torch.equal(fin, torch.isfinite(y)), p
        # Adam (eps 1e-15) turns a gradient whose sign is decided by the summation order of the compositing backward's
        # atomics into a +-lr step, so a few elements may sit a few learning rates apart; the bulk must agree closely
        rel = (x[fin] - y[fin]).abs() / (x[fin].abs().mean() + 1e-12)
        assert float((rel > 1e-3).float().mean()) < 0.01 and float(rel.max()) < 0.2, (p, float((rel > 1e-3).float().mean()), float(rel.max()))
    assert torch.equal(a.state["count"], b.state["count"])
    for key in ("grad2d", "vis"):                                  # statistics of slightly different parameter trajectories
        rel = (a.state[key] - b.state[key]).abs() / (a.state[key].abs().mean() + 1e-12)
        assert float((rel > 1e-2).float().mean()) < 0.01, (key, float((rel > 1e-2).float().mean()), float(rel.max()))
