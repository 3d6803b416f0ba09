"""Seeded inputs for the reference's in-tree functions and the code that evaluates the REFERENCE on them (TEST INFRASTRUCTURE).

`inputs()` builds the inputs (CPU, float32, fixed seeds); `evaluate(m, inp)` runs the compiled reference module (oracle/ref_link/build.py)
on them and returns numpy arrays.  tools/gen_reference_intree_golden.py stores inputs + outputs in tests/golden/reference_intree.npz;
tests/test_reference_intree_pins.py compares this repository's restatements with the stored outputs everywhere, and the stored
outputs with a live evaluation where the module exists (this container, and the GPU box through the prebuilt oracle/_ref)."""
import numpy as np
import torch

CAM = dict(fx=14.0, fy=13.0, cx=8.3, cy=5.7, w=17, h=12)
ADAM_LRS = (1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 2.5e-3 / 20)   # neural_gaussian.cpp:434-453
ADAM_SHAPES = ((3,), (3,), (4,), (), (1, 3), (3, 3))           # offsets, scaling, quaternion, opacity, features_dc, features_rest (sh degree 1)


def inputs():
    g = torch.Generator().manual_seed(20260925)
    r = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    inp = {}
    inp["img1"], inp["img2"] = r(1, 3, 24, 31), r(1, 3, 24, 31)
    inp["mask"] = (r(24, 31, 1) > 0.3).float()
    # SDF batch: some rows beyond the isigma clamp (5e2), some targets saturating the 1e-7 clamp
    inp["pred_sdf"], inp["gt_sdf"] = n(64, 1) * 0.1, n(64, 1) * 0.05
    inp["gt_sdf"][:6] = 1.0
    inp["pred_isigma"] = 1.0 + r(64, 1) * 120.0
    inp["pred_isigma"][6:12] = 900.0
    inp["grad"] = n(64, 3)
    inp["grad"][3] = 0.0                                       # norm backward at zero
    inp["hessian"] = n(64, 3)
    inp["gs_sdf"], inp["gs_w"] = n(80, 1) * 0.2, r(80, 1)
    inp["n_gs"] = torch.nn.functional.normalize(n(80, 3), dim=-1)
    inp["n_sdf"] = torch.nn.functional.normalize(n(80, 3), dim=-1)
    inp["n_sdf"][5] = float("nan")                             # nan_to_num inside gs_sdf_normal_loss
    inp["dist"] = r(24, 31, 1)
    # depth -> normal
    c, s = np.cos(0.3), np.sin(0.3)
    inp["pose"] = torch.tensor([[c, 0.0, s, 0.2], [0.0, 1.0, 0.0, -0.1], [-s, 0.0, c, 0.5]], dtype=torch.float32)
    inp["depth"] = 2.0 + r(CAM["h"], CAM["w"], 1)
    # rotations
    inp["quat"] = torch.nn.functional.normalize(n(50, 4), dim=-1)
    inp["rot6d"] = n(50, 6)
    # ray samples
    inp["ray_o"] = n(20, 3) * 0.1
    inp["ray_d"] = torch.nn.functional.normalize(n(20, 3), dim=-1)
    inp["ray_depth"] = 1.0 + 3.0 * r(20, 1)
    # a small coloured mesh for the PLY writer
    inp["mesh_v"], inp["mesh_f"] = n(7, 3), torch.randint(0, 7, (5, 3), generator=g, dtype=torch.int32)
    inp["mesh_c"] = torch.randint(0, 256, (7, 3), generator=g, dtype=torch.int32).to(torch.uint8)
    # Adam surgery: six parameter groups of 12 rows, gradients for 6 steps, the extension rows
    for k, sh in enumerate(ADAM_SHAPES):
        inp[f"adam_p{k}"] = n(12, *sh)
        inp[f"adam_g{k}"] = n(6, 20, *sh)                      # step, (up to 20) rows
        inp[f"adam_e{k}"] = n(2, 3, *sh)                       # two extensions of 3 rows
        inp[f"adam_r{k}"] = n(14, *sh)                         # the replacement
    return inp


def _np(t):
    return t.detach().cpu().numpy()


def _with_grads(fn, *xs):
    xs = [x.clone().requires_grad_(True) for x in xs]
    v = fn(*xs)
    gs = torch.autograd.grad(v, xs)
    return [_np(v)] + [_np(g) for g in gs]


def adam_script(api, inp):
    """the schedule both sides run: 2 steps, prune rows, step, append rows, step, prune + append, step, replace, step.
    `api` supplies make(params, lrs), step(grads), prune(keep), cat(ext), prune_cat(keep, ext), replace(new), snapshot() -> [p, m, v] * 6"""
    K = len(ADAM_SHAPES)
    api.make([inp[f"adam_p{k}"].clone() for k in range(K)], ADAM_LRS)
    out = []
    step = [0]

    def do_step():
        rows = api.rows()
        api.step([inp[f"adam_g{k}"][step[0], :rows].clone() for k in range(K)])
        step[0] += 1
    do_step(); do_step(); out.append(api.snapshot())
    api.prune(torch.tensor([0, 2, 3, 5, 6, 7, 9, 11])); do_step(); out.append(api.snapshot())
    api.cat([inp[f"adam_e{k}"][0].clone() for k in range(K)]); do_step(); out.append(api.snapshot())
    api.prune_cat(torch.tensor([1, 2, 4, 5, 6, 8, 9, 10]), [inp[f"adam_e{k}"][1].clone() for k in range(K)]); do_step(); out.append(api.snapshot())
    api.replace([inp[f"adam_r{k}"].clone() for k in range(K)]); do_step(); out.append(api.snapshot())
    return out


class _RefAdam:
    """optimizer_utils.cpp on a torch::optim::Adam, one parameter per group (as NeuralGS registers them)"""

    def __init__(self, m):
        self.m = m

    def make(self, params, lrs):
        self.p = [p.requires_grad_(True) for p in params]
        self.opt = self.m.Adam(self.p, list(lrs), 1e-15)

    def rows(self):
        return self.p[0].shape[0]

    def step(self, grads):
        for p, g in zip(self.p, grads):
            p.grad = g
        self.opt.step()

    def prune(self, keep):
        self.p = [self.m.prune_optimizer(self.opt, keep, p, k) for k, p in enumerate(self.p)]

    def cat(self, ext):
        self.p = [self.m.cat_tensors_to_optimizer(self.opt, e, p, k) for k, (p, e) in enumerate(zip(self.p, ext))]

    def prune_cat(self, keep, ext):
        self.p = [self.m.prune_cat_tensors_to_optimizer(self.opt, p, keep, e, k) for k, (p, e) in enumerate(zip(self.p, ext))]

    def replace(self, new):
        self.p = [self.m.replace_tensors_to_optimizer(self.opt, p, nw, k) for k, (p, nw) in enumerate(zip(self.p, new))]

    def snapshot(self):
        out = []
        for k in range(len(self.p)):
            assert self.opt.param(k).data_ptr() == self.p[k].data_ptr()
            mo = self.opt.moments(k)
            out += [_np(self.p[k]), _np(mo[0]), _np(mo[1])]
        return out


def evaluate(m, inp):
    """the reference's outputs on `inp` (dict of numpy arrays)"""
    out = {}
    out["window11"] = _np(m.gaussian(11, 1.5))
    win = m.create_window(11, 3)
    out["window2d"] = _np(win)
    out["ssim"], out["ssim_g1"] = _with_grads(lambda a: m._ssim(a, inp["img2"], win, 11, 3, True), inp["img1"])
    hw3 = lambda t: t[0].permute(1, 2, 0).contiguous()
    a, b = hw3(inp["img1"]), hw3(inp["img2"])
    out["rgb"], out["rgb_g"] = _with_grads(lambda x: m.rgb_loss(x, b, None), a)
    out["rgb_masked"], out["rgb_masked_g"] = _with_grads(lambda x: m.rgb_loss(x, b, inp["mask"]), a)
    out["sdf_loss"], out["sdf_loss_gs"], out["sdf_loss_gi"] = _with_grads(lambda s, i: m.sdf_loss(s, inp["gt_sdf"], i), inp["pred_sdf"], inp["pred_isigma"])
    out["eikonal"], out["eikonal_g"] = _with_grads(m.eikonal_loss, inp["grad"])
    out["curvate"], out["curvate_g"] = _with_grads(m.curvate_loss, inp["hessian"])
    out["gs_sdf"], out["gs_sdf_g"] = _with_grads(lambda s: m.gs_sdf_loss(s, inp["gs_w"]), inp["gs_sdf"])
    out["gs_normal"], out["gs_normal_g"] = _with_grads(lambda x: m.gs_sdf_normal_loss(x, inp["n_sdf"], inp["gs_w"]), inp["n_gs"])
    out["distortion"], out["distortion_g"] = _with_grads(m.distortion_loss, inp["dist"])
    out["psnr"] = np.float32(m.psnr(inp["img1"], inp["img2"]))
    out["depth_normal"] = _np(m.depth_to_normal(CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"], CAM["w"], CAM["h"], inp["pose"], inp["depth"]))
    out["quat_rot"] = _np(m.normalized_quat_to_rotmat(inp["quat"]))
    out["rot6d_rot"] = _np(m.rotation_6d_to_matrix(inp["rot6d"]))
    out["meshgrid"] = _np(m.meshgrid_3d(-0.5, 0.75, 0.0, 0.5, 1.0, 1.6, 0.25, "cpu"))
    out["meshgrid_flat"] = _np(m.meshgrid_3d(-0.5, 0.75, 0.0, 0.5, 1.0, 1.0, 0.25, "cpu"))
    rays = dict(origin=inp["ray_o"], direction=inp["ray_d"], depth=inp["ray_depth"], xyz=inp["ray_o"] + inp["ray_d"] * inp["ray_depth"],
                ridx=torch.arange(20))
    torch.manual_seed(7)
    fs = m.sample_free_pts(rays, 3)
    for k in ("xyz", "ray_sdf", "depth", "ridx", "origin", "direction"):
        out["free_" + k] = _np(fs[k])
    torch.manual_seed(8)
    ss = m.sample_surface_pts(rays, 3, 0.05)
    for k in ("xyz", "ray_sdf", "depth", "ridx"):
        out["surf_" + k] = _np(ss[k])
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = d + "/mesh.ply"
        m.save_mesh_as_ply(path, inp["mesh_v"], inp["mesh_f"], inp["mesh_c"])
        out["mesh_ply_bytes"] = np.frombuffer(open(path, "rb").read(), np.uint8).copy()
    for phase, snap in enumerate(adam_script(_RefAdam(m), inp)):
        for j, arr in enumerate(snap):
            out[f"adam_{phase}_{j // 3}_{'pmv'[j % 3]}"] = arr
    return out
