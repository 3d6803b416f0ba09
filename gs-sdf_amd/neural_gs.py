"""Host-side mirror of the reference's `NeuralGS` (include/neural_gaussian/neural_gaussian.{h,cpp}): parameters and
activations, `render()`, densification statistics (`update_state`) and the refinement strategy
(`train_callback`: grow = duplicate/split, prune, opacity reset, LR decay) including the Adam-state surgery of
include/optimizer/optimizer_utils/optimizer_utils.cpp.  Plus the per-ray SDF sample construction of
include/utils/utils.cpp:336-393 / neural_mapping.cpp:73-104.

Everything here is host policy over torch tensors (it runs on any device); the arithmetic of `render()` goes
through gs_sdf_amd.ops -> the C ABI.  Citations are /root/reference/<file>:<line>.
"""
import math
from dataclasses import dataclass

import torch

from . import ops


@dataclass
class Cameras:                       # sensor::Cameras (include/utils/sensor_utils/cameras.hpp:43-174): pinhole intrinsics
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int


@dataclass
class GSConfig:                      # the k_* globals this class reads (config/base.yaml:37-74, params.cpp)
    sh_degree: int = 0
    near: float = 0.05
    far: float = 300.0
    use_absgrad: bool = False
    center_reg: bool = False
    prune_opa: float = 0.05
    grow_grad2d: float = 0.0002
    grow_scale3d: float = 0.01
    grow_scale2d: float = 0.05
    prune_scale3d: float = 0.1
    refine_scale2d_stop_iter: int = 0
    refine_start_iter: int = 500
    refine_every: int = 100
    reset_every: int = 3000
    sh_degree_interval: int = 1000
    pause_refine_after_reset: int = 0
    lr_end: float = 1e-4
    detach_sdf_grad: bool = False


def normalized_quat_to_rotmat(q):   # include/utils/utils.cpp:538-558 (w,x,y,z)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def update_densify_state(state, info, n, use_absgrad=False, want_radii=False, key_for_gradient="gradient_2dgs", eager=False):
    """NeuralGS::update_state (neural_gaussian.cpp:626-680) on a state dict: grad2d / count accumulate, vis / radii take
    the maximum.  Device tensors: ONE launch (include/gsdf_hip.h: gsdf_densify_stats) instead of the reference's ~12
    eager kernels; host tensors (the CPU tests of the refinement policy): the reference's torch expressions."""
    src = info["absgrad"] if use_absgrad else info[key_for_gradient]
    n_cameras, width, height = int(info["n_cameras"]), int(info["width"]), int(info["height"])
    dev = src.device
    for k in ("grad2d", "count", "vis") + (("radii",) if want_radii else ()):
        if k not in state:
            state[k] = torch.zeros(n, device=dev)
    gs_ids = info["gaussian_ids"]
    if src.is_cuda and not eager:      # eager=True: the reference's own torch expressions on the device (bench.py --reference-loop)
        from . import capi
        capi.check(capi.lib().gsdf_densify_stats(
            gs_ids.shape[0], n, n_cameras, width, height, capi.f32(src.grad.contiguous(), "densify gradient"),
            capi.ptr(gs_ids.contiguous(), torch.int64), capi.f32(info["visibilities"].reshape(-1).contiguous()),
            capi.ptr(info["radii"].contiguous(), torch.int32) if want_radii else None, capi.f32(state["grad2d"]),
            capi.f32(state["count"]), capi.f32(state["vis"]), capi.f32(state["radii"]) if want_radii else None, capi.stream()),
            "densify_stats")
        return
    grads = src.grad.clone()
    grads[:, 0] *= width * 0.5 * n_cameras
    grads[:, 1] *= height * 0.5 * n_cameras
    state["grad2d"].index_add_(0, gs_ids, grads.norm(2, -1))
    gs_vis = info["visibilities"].reshape(-1)
    state["vis"][gs_ids] = torch.maximum(state["vis"].index_select(0, gs_ids), gs_vis)
    state["count"].index_add_(0, gs_ids, torch.ones_like(gs_ids, dtype=torch.float32))
    if want_radii:
        state["radii"][gs_ids] = torch.maximum(state["radii"].index_select(0, gs_ids), info["radii"] / float(max(width, height)))


class NeuralGS:
    """Parameter names, groups and learning rates follow neural_gaussian.cpp:426-453 (all Adam eps 1e-15)."""

    PARAMS = ("offsets_", "scaling_", "quaternion_", "opacity_", "features_dc_", "features_rest_")

    def __init__(self, anchors, scaling, quaternion, opacity, features_dc, features_rest, cfg=None, spatial_scale=1.0,
                 num_train_data=1):
        self.cfg = cfg or GSConfig()
        self.anchors_ = anchors.detach().clone()                       # registered without grad in the reference
        mk = lambda t: t.detach().clone().requires_grad_(True)
        self.offsets_ = mk(torch.zeros_like(anchors))
        self.scaling_, self.quaternion_, self.opacity_ = mk(scaling), mk(quaternion), mk(opacity.reshape(-1))
        self.features_dc_, self.features_rest_ = mk(features_dc), mk(features_rest)
        self.spatial_scale_ = min(float(spatial_scale), 2.0)
        self.original_spatial_scale_ = float(spatial_scale)
        self.sh_degree_to_use_ = 0
        self.num_train_data_ = num_train_data
        self.state = {}
        self.gs_param_start_idx = 0
        self.key_for_gradient = "gradient_2dgs"

    # ---- optimizer (neural_gaussian.cpp:434-453; neural_mapping.cpp:855-858 appends the groups after the SDF ones)
    def param_groups(self):
        s = self.spatial_scale_
        lrs = (1.6e-4 * s, 5e-3, 1e-3, 5e-2, 2.5e-3, 2.5e-3 / 20)
        return [dict(params=[getattr(self, n)], lr=lr, eps=1e-15) for n, lr in zip(self.PARAMS, lrs)]

    def make_optimizer(self, sdf_groups=()):
        groups = list(sdf_groups) + self.param_groups()
        self.gs_param_start_idx = len(list(sdf_groups))
        return torch.optim.Adam(groups, eps=1e-15)

    # ---- activations (neural_gaussian.cpp:463-492)
    def get_xyz(self):
        return (self.anchors_ + self.offsets_).view(-1, 3)

    def get_scale(self):
        return torch.exp(self.scaling_).view(-1, 3)

    def get_opacity(self, training=True):
        return torch.sigmoid(self.opacity_)

    def generate_gaussian(self, training=True):
        return (self.get_xyz(), self.quaternion_.view(-1, 4), self.get_scale(), self.get_opacity(training),
                torch.cat([self.features_dc_, self.features_rest_], 1))

    # ---- render (neural_gaussian.cpp:495-566)
    def render(self, pose_cam2world, camera, training=False, bck_color=0, sample_seed=0):
        dev = self.anchors_.device
        K = torch.tensor([[camera.fx, 0.0, camera.cx], [0.0, camera.fy, camera.cy], [0.0, 0.0, 1.0]], device=dev)[None]
        rot, pos = pose_cam2world[:, :3].to(dev), pose_cam2world[:, 3:4].to(dev)
        w2c = torch.cat([torch.cat([rot.t(), -rot.t() @ pos], 1), torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)], 0)[None]
        xyz, quat, scales, opacity, sh = self.generate_gaussian(training)
        renders, alphas, info = ops.rasterization_2dgs_sdf(
            xyz, quat, scales, opacity, sh, w2c.contiguous(), K, camera.width, camera.height, "RGB+ED", self.cfg.near, self.cfg.far,
            0.0, self.sh_degree_to_use_, True, 16, None, False, self.cfg.use_absgrad, False, self.cfg.center_reg, sample_seed)
        color, depth = info.pop("color")[0], info.pop("depth")[0]     # = renders[...,0:3][0], renders[...,3:4][0]
        out = {}
        if bck_color == 2:
            out["color"] = color + (1.0 - alphas[0]) * torch.rand(camera.height, camera.width, 3, device=dev)
        elif bck_color == 1:
            out["color"] = color + (1.0 - alphas[0])
        else:
            out["color"] = color
        out.update(depth=depth, alpha=alphas, xyz=xyz)
        out.update(info)
        if out[self.key_for_gradient].requires_grad:
            out[self.key_for_gradient].retain_grad()
        return out

    # ---- densification statistics (neural_gaussian.cpp:626-680)
    def update_state(self, info):
        update_densify_state(self.state, info, self.anchors_.shape[0], self.cfg.use_absgrad,
                             self.cfg.refine_scale2d_stop_iter > 0, self.key_for_gradient)

    def zero_state(self):
        self.state["grad2d"].zero_()
        self.state["count"].zero_()
        if self.cfg.refine_scale2d_stop_iter > 0:
            self.state["radii"].zero_()

    # ---- Adam-state surgery (optimizer_utils.cpp:5-165): keep the moments of surviving rows, zeros for new rows
    def _swap(self, optimizer, name, new_tensor, moments):
        old = getattr(self, name)
        new_tensor = new_tensor.detach().requires_grad_(True)
        setattr(self, name, new_tensor)
        if optimizer is None:
            return
        gi = self.gs_param_start_idx + self.PARAMS.index(name)
        group = optimizer.param_groups[gi]
        st = optimizer.state.pop(old, None)
        group["params"][0] = new_tensor
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = moments(st["exp_avg"]), moments(st["exp_avg_sq"])
            optimizer.state[new_tensor] = st

    def _apply(self, optimizer, keep_idx, ext):
        """rows kept (index tensor or None = all) followed by the extension rows `ext[name]` (or nothing)."""
        for name in self.PARAMS:
            old = getattr(self, name).detach()
            base = old if keep_idx is None else old.index_select(0, keep_idx)
            e = ext.get(name) if ext else None
            new = base if e is None else torch.cat([base, e], 0)
            self._swap(optimizer, name, new, lambda m, e=e: (
                (m if keep_idx is None else m.index_select(0, keep_idx)) if e is None else
                torch.cat([m if keep_idx is None else m.index_select(0, keep_idx), torch.zeros_like(e)], 0)))

    # ---- grow (neural_gaussian.cpp:690-827)
    def grow_gs(self, it, optimizer, generator=None):
        cfg = self.cfg
        grads = self.state["grad2d"] / self.state["count"].clamp_min(1)
        is_grad_high = grads > cfg.grow_grad2d
        is_small = self.get_scale()[:, :2].max(-1).values <= cfg.grow_scale3d * self.spatial_scale_
        is_dupli = is_grad_high & is_small
        is_split = is_grad_high & ~is_small
        if it < cfg.refine_scale2d_stop_iter:
            is_split |= self.state["radii"] > cfg.grow_scale2d
        n_dupli = self.duplicate(optimizer, is_dupli)
        is_split = torch.cat([is_split, torch.zeros(n_dupli, dtype=torch.bool, device=is_split.device)])
        n_split = self.split(optimizer, is_split, generator)
        return n_dupli, n_split

    def duplicate(self, optimizer, is_dupli):
        n = int(is_dupli.sum())
        if n > 0:
            idx = is_dupli.nonzero().squeeze(-1)
            self.anchors_ = torch.cat([self.anchors_, self.anchors_.index_select(0, idx)], 0)
            self._apply(optimizer, None, {p: getattr(self, p).detach().index_select(0, idx) for p in self.PARAMS})
            for k in self.state:
                self.state[k] = torch.cat([self.state[k], self.state[k].index_select(0, idx)])
        return n

    def split(self, optimizer, is_split, generator=None):
        n = int(is_split.sum())
        if n > 0:
            K = 2
            sel, rest = is_split.nonzero().squeeze(-1), (~is_split).nonzero().squeeze(-1)
            scales = self.get_scale().detach().index_select(0, sel)
            scales = torch.cat([scales[:, :2], torch.zeros(n, 1, device=scales.device)], 1)
            sample_scales = scales[None] * torch.randn(K, n, 3, device=scales.device, generator=generator)
            quats = torch.nn.functional.normalize(self.quaternion_.detach().index_select(0, sel), dim=-1)
            rot = normalized_quat_to_rotmat(quats)
            split_offsets = (torch.einsum("nij,nj,bnj->bni", rot, scales, sample_scales)
                             + self.offsets_.detach().index_select(0, sel)[None]).reshape(-1, 3)
            ext = dict(offsets_=split_offsets, scaling_=torch.log(scales / 1.6).repeat(K, 1),
                       quaternion_=self.quaternion_.detach().index_select(0, sel).repeat(K, 1),
                       opacity_=self.opacity_.detach().index_select(0, sel).repeat(K),
                       features_dc_=self.features_dc_.detach().index_select(0, sel).repeat(K, 1, 1),
                       features_rest_=self.features_rest_.detach().index_select(0, sel).repeat(K, 1, 1))
            self.anchors_ = torch.cat([self.anchors_.index_select(0, rest), self.anchors_.index_select(0, sel).repeat(K, 1)], 0)
            self._apply(optimizer, rest, ext)
            for k in self.state:
                self.state[k] = torch.cat([self.state[k].index_select(0, rest), self.state[k].index_select(0, sel).repeat(K)])
        return n

    # ---- prune (neural_gaussian.cpp:829-916)
    def _prune(self, optimizer, is_prune):
        n = int(is_prune.sum())
        if n > 0:
            valid = (~is_prune).nonzero().squeeze(-1)
            self.anchors_ = self.anchors_.index_select(0, valid)
            self._apply(optimizer, valid, None)
            for k in self.state:
                self.state[k] = self.state[k].index_select(0, valid)
        return n

    def prune_gs(self, it, optimizer, prune_opa_only=False):
        is_prune = self.get_opacity().detach() < self.cfg.prune_opa
        scale = self.get_scale().detach()[:, :2]
        is_prune |= scale.min(-1).values < 1e-4
        if not prune_opa_only and it > self.cfg.reset_every:
            is_prune |= scale.max(-1).values > self.cfg.prune_scale3d * self.original_spatial_scale_
        return self._prune(optimizer, is_prune)

    def prune_invisible_gs(self, it, optimizer):
        if it > 0 and it % self.num_train_data_ == 0 and "vis" in self.state:
            is_prune = self.state["vis"] < 1e-4
            self.state["vis"].zero_()
            return self._prune(optimizer, is_prune)
        return 0

    def prune_nan_gs(self, optimizer):        # neural_gaussian.cpp:907-916
        if self.offsets_.is_cuda:
            # one launch (count + mask) instead of 3 x (isnan, any) + 2 x or; the mask is only gathered from when the count is non-zero
            from . import ops
            count, mask = ops.nan_rows(self.offsets_.detach(), self.scaling_.detach(), self.quaternion_.detach(), want_mask=True)
            if int(count.item()) == 0:
                return 0
            return self._prune(optimizer, mask)
        is_prune = (self.offsets_.detach().isnan().any(-1) | self.scaling_.detach().isnan().any(-1)
                    | self.quaternion_.detach().isnan().any(-1))
        return self._prune(optimizer, is_prune)

    def reset_opacity(self, optimizer):       # neural_gaussian.cpp:918-926
        cap = math.log(self.cfg.prune_opa * 2.0 / (1.0 - self.cfg.prune_opa * 2.0))
        new = self.opacity_.detach().clamp_max(cap)
        self._swap(optimizer, "opacity_", new, lambda m: torch.zeros_like(m))

    # ---- per-iteration callback (neural_gaussian.cpp:568-624)
    @torch.no_grad()
    def train_callback(self, it, total_iter, optimizer, info, view_parallel=None):
        """`view_parallel` (trainer.ViewParallel, not in the reference, which is single-GPU): before every decision that
        changes the splat set the ranks merge their densification statistics (sum / max) so that all of them take it
        identically (parameters are identical after the gradient all-reduce, so NaN pruning needs no exchange)."""
        cfg = self.cfg
        refine_stop = total_iter // 2
        log = {}
        if info:
            if it >= refine_stop:
                return log
            self.update_state(info)
            refine_now = (0 < it < refine_stop and it > cfg.refine_start_iter and it % cfg.refine_every == 0
                          and (it % cfg.reset_every) >= cfg.pause_refine_after_reset)
            if view_parallel is not None and (refine_now or (it > 0 and it % self.num_train_data_ == 0)):
                # grad2d / count are reset right after a refine step (zero_state), so that is the only place to sum them
                view_parallel.sync_refine_state(self.state, sums=refine_now, maxs=True)
            log["nan"] = self.prune_nan_gs(optimizer)
            log["invisible"] = self.prune_invisible_gs(it, optimizer)
            self.sh_degree_to_use_ = min(cfg.sh_degree, it // cfg.sh_degree_interval)
            if 0 < it < refine_stop:
                if it > cfg.refine_start_iter and it % cfg.refine_every == 0 and (it % cfg.reset_every) >= cfg.pause_refine_after_reset:
                    gen = None
                    if view_parallel is not None:
                        # replicas must draw the SAME split offsets (neural_gaussian.cpp:781 uses the global RNG of its
                        # single process): ranks render different views and consume different amounts of randomness, so
                        # the generator is seeded from the iteration alone
                        gen = torch.Generator(device=self.anchors_.device).manual_seed(0x5EED0000 + int(it))
                    log["dupli"], log["split"] = self.grow_gs(it, optimizer, gen)
                    log["pruned"] = self.prune_gs(it, optimizer)
                    self.zero_state()
                if it % cfg.reset_every == 0:
                    self.reset_opacity(optimizer)
                    log["reset_opacity"] = True
        if optimizer is not None:               # exponential LR decay of the position group (:608-623)
            r = it / total_iter
            lr0, lr1 = 1.6e-4 * self.spatial_scale_, 1.6e-6 * self.spatial_scale_
            lr = math.exp(math.log(lr0) * (1 - r) + math.log(lr1) * r)
            self._set_lrs(optimizer, lr, 0.0 if cfg.detach_sdf_grad else min(lr, cfg.lr_end))
        return log

    def _set_lrs(self, optimizer, xyz_lr, sdf_lr):
        optimizer.param_groups[self.gs_param_start_idx]["lr"] = xyz_lr
        for i in range(self.gs_param_start_idx):
            optimizer.param_groups[i]["lr"] = sdf_lr


class FlatNeuralGS(NeuralGS):
    """NeuralGS over the trainer's flat buffers: the six parameters are views into ONE flat buffer (trainer.SplatParams, one
    all-reduce message, fused activations) and the optimizer is the fused Adam (trainer.FusedAdam, one launch).  The whole
    refinement policy of NeuralGS (grow = duplicate / split, prune, opacity reset, LR decay; neural_gaussian.cpp:568-926) runs
    unchanged on top: only the storage primitives differ — `_apply` rebuilds the flat parameter buffer and both Adam moments
    with one row-gather launch each (surviving rows keep their moments, new rows start from zero, optimizer_utils.cpp:5-165),
    `_swap` (reset_opacity) writes the opacity segment in place and zeroes its moments."""

    def __init__(self, anchors, scaling, quaternion, opacity, features_dc, features_rest, cfg=None, spatial_scale=1.0,
                 num_train_data=1):
        from .trainer import SplatParams
        self.cfg = cfg or GSConfig()
        n = anchors.shape[0]
        self.params = SplatParams(anchors.detach().clone(), torch.zeros_like(anchors), scaling.detach().clone(),
                                  quaternion.detach().clone(), opacity.detach().reshape(n).clone(),
                                  features_dc.detach().reshape(n, -1).clone(), features_rest.detach().reshape(n, -1).clone())
        self.spatial_scale_ = min(float(spatial_scale), 2.0)
        self.original_spatial_scale_ = float(spatial_scale)
        self.sh_degree_to_use_ = 0
        self.num_train_data_ = num_train_data
        self.state = {}
        self.gs_param_start_idx = 0
        self.key_for_gradient = "gradient_2dgs"
        self.adam_group = 0

    _FIELD = dict(offsets_="offsets", scaling_="scaling", quaternion_="quaternion", opacity_="opacity", features_dc_="features_dc",
                  features_rest_="features_rest")

    def _field(self, name):
        v = self.params.views[self._FIELD[name]]
        n = v.shape[0]
        if name == "opacity_":
            return v.view(n)
        if name in ("features_dc_", "features_rest_"):
            return v.view(n, -1, 3)
        return v

    anchors_ = property(lambda self: self.params.anchors, lambda self, v: setattr(self.params, "anchors", v.contiguous()))
    offsets_ = property(lambda self: self._field("offsets_"))
    scaling_ = property(lambda self: self._field("scaling_"))
    quaternion_ = property(lambda self: self._field("quaternion_"))
    opacity_ = property(lambda self: self._field("opacity_"))
    features_dc_ = property(lambda self: self._field("features_dc_"))
    features_rest_ = property(lambda self: self._field("features_rest_"))

    def generate_gaussian(self, training=True):
        return self.params.activated()            # one fused launch; gradients accumulate into the flat gradient buffer

    def make_optimizer(self, sdf_groups=()):
        from .trainer import FusedAdam
        if sdf_groups:
            raise RuntimeError("FlatNeuralGS.make_optimizer: the SDF network has its own FusedAdam (one per leg)")
        opt = FusedAdam(eps=1e-15)
        s = self.spatial_scale_
        lrs = dict(offsets=1.6e-4 * s, scaling=5e-3, quaternion=1e-3, opacity=5e-2, features_dc=2.5e-3, features_rest=2.5e-3 / 20)
        self.adam_group = opt.add_group(self.params.flat, self.params.flat_grad,
                                        [(self.params.views[k].numel(), lrs[k]) for k in self.params.views])
        return opt

    def _apply(self, optimizer, keep_idx, ext):
        e = None if not ext else {self._FIELD[k]: v for k, v in ext.items()}
        self.params.resize(keep_idx, e, optimizer, self.adam_group)

    def _swap(self, optimizer, name, new_tensor, moments):
        """only reset_opacity comes through here: same rows, new values, fresh moments"""
        with torch.no_grad():
            self._field(name).copy_(new_tensor.view_as(self._field(name)))
        if optimizer is not None:
            optimizer.zero_segment_moments(self.adam_group, list(self.params.views).index(self._FIELD[name]))

    def _set_lrs(self, optimizer, xyz_lr, sdf_lr):
        optimizer.set_lr(self.adam_group, 0, xyz_lr)          # segment 0 = offsets (the position group)
        self.sdf_lr = sdf_lr                                   # the SDF leg's optimizer reads it (bench / trainer loop)


# ------------------------------------------------------------------------------------------------------------------
# per-ray SDF batch construction (a16): utils.cpp:336-393, neural_mapping.cpp:73-104 (without the octree voxel sample,
# which belongs to the kaolin_wisp_cpp replacement, SURVEY 8f "next #2")
# ------------------------------------------------------------------------------------------------------------------
def sample_surface_pts(origin, direction, depth, surface_sample_num, std, generator=None):
    n = origin.shape[0]
    ray_sdf = torch.randn(n, surface_sample_num, 1, device=origin.device, generator=generator) * std
    end = origin + direction * depth
    xyz = (end[:, None] - direction[:, None] * ray_sdf).reshape(-1, 3)
    ridx = torch.arange(n, device=origin.device)[:, None].repeat(1, surface_sample_num).reshape(-1)
    return xyz, ray_sdf.reshape(-1, 1), ridx


def sample_free_pts(origin, direction, depth, sample_num, generator=None):
    n = origin.shape[0]
    steps = torch.arange(sample_num, device=origin.device, dtype=torch.float32)[None].repeat(n, 1)
    steps = (steps + torch.rand(n, sample_num, device=origin.device, generator=generator)) / sample_num
    ridx = torch.arange(n, device=origin.device)[:, None].repeat(1, sample_num).reshape(-1)
    d = depth.index_select(0, ridx) * steps.reshape(-1, 1)
    xyz = origin.index_select(0, ridx) + direction.index_select(0, ridx) * d
    return xyz, depth.index_select(0, ridx) - d, ridx


def sample_rays(origin, direction, depth, sample_std, truncated_dis, surface_sample_num=3, free_sample_num=3,
                inrange=None, generator=None):
    """-> xyz [B,3], ray_sdf [B,1], ridx [B]: free-space + near-surface + end-point samples, SDF targets truncated to
    +-truncated_dis, filtered by `inrange(xyz) -> bool mask` (SubMap::get_inrange_mask)."""
    fx, fs, fr = sample_free_pts(origin, direction, depth, free_sample_num, generator)
    sx, ss, sr = sample_surface_pts(origin, direction, depth, surface_sample_num, sample_std, generator)
    xyz, sdf, ridx = torch.cat([fx, sx]), torch.cat([fs, ss]), torch.cat([fr, sr])
    sdf = torch.where(sdf.abs() > truncated_dis, sdf.sign() * truncated_dis, sdf)
    n = origin.shape[0]
    xyz = torch.cat([xyz, origin + direction * depth])
    sdf = torch.cat([sdf, torch.zeros(n, 1, device=origin.device)])
    ridx = torch.cat([ridx, torch.arange(n, device=origin.device)])
    if inrange is not None:
        keep = inrange(xyz).reshape(-1).nonzero().squeeze(-1)
        xyz, sdf, ridx = xyz.index_select(0, keep), sdf.index_select(0, keep), ridx.index_select(0, keep)
    return xyz, sdf, ridx


# ------------------------------------------------------------------------------------------------------------------
# SDF-aided splat initialisation (a17): neural_gaussian.cpp:19-127
# ------------------------------------------------------------------------------------------------------------------
def rotation_6d_to_matrix(r6):       # include/utils/utils.cpp:693-719 (Gram-Schmidt, columns b1,b2,b3)
    a1, a2 = r6[..., 0:3], r6[..., 3:6]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack([b1, b2, torch.cross(b1, b2, dim=-1)], -1)


@torch.no_grad()
def init_gs_with_sdf(local_map, xyzs, mesh_res, init_opa=False, batch_size=50 * 32768):
    """Splat orientation from the SDF: normal = grad, in-plane axis = diagonal-Hessian direction, rotation ->
    axis-angle -> quaternion (w,x,y,z); optional opacity exp(-sdf^2 * isigma).  Batched like the reference
    (k_vis_batch_pt_num = 50 * batch_pt_num, params.cpp:360)."""
    n = xyzs.shape[0]
    grad, curv = torch.empty(n, 3, device=xyzs.device), torch.empty(n, 3, device=xyzs.device)
    quat = torch.empty(n, 4, device=xyzs.device)
    opa = torch.empty(n, device=xyzs.device) if init_opa else None
    for s in range(0, n, batch_size):
        x = xyzs[s:s + batch_size]
        g, h = local_map.get_gradient(x, mesh_res, None, True, True)
        c = torch.nn.functional.normalize(g, dim=-1)
        basis = torch.nn.functional.normalize(h, dim=-1)
        rot = rotation_6d_to_matrix(torch.cat([c, basis], -1))
        rot = torch.stack([rot[..., 1], rot[..., 2], rot[..., 0]], -1)
        trace = rot[:, 0, 0] + rot[:, 1, 1] + rot[:, 2, 2]
        angle = torch.acos((trace[:, None] - 1.0) * 0.5)
        axis = torch.stack([rot[:, 2, 1] - rot[:, 1, 2], rot[:, 0, 2] - rot[:, 2, 0], rot[:, 1, 0] - rot[:, 0, 1]], -1) / (2.0 * torch.sin(angle))
        axis = torch.nn.functional.normalize(axis, dim=-1)
        q = torch.cat([torch.cos(angle * 0.5), torch.sin(angle * 0.5) * axis], -1).nan_to_num()
        grad[s:s + batch_size], curv[s:s + batch_size], quat[s:s + batch_size] = g, h, q
        if init_opa:
            sdf, isig = local_map.get_sdf(x)
            opa[s:s + batch_size] = torch.exp(-sdf.square() * isig).squeeze(-1)
    out = dict(quaternion=quat, grad=grad, curv_dom=curv)
    if init_opa:
        out["opacity"] = opa
    return out


# ------------------------------------------------------------------------------------------------------------------
# gs.ply checkpoint (3DGS-compatible binary PLY): NeuralGS::export_gs_to_ply / load_ply_to_gs, neural_gaussian.cpp:928-1188
# ------------------------------------------------------------------------------------------------------------------
def export_gs_to_ply(gs, path):
    """vertex properties (float32, binary little endian): x y z, f_dc_*, f_rest_* (channel-major, transpose(1,2).flatten(1)),
    opacity (logit), scale_0..2 (log; the third is log(1e-6) for 3DGS viewers), rot_0..3 (w,x,y,z)."""
    import numpy as np
    xyz = gs.get_xyz().detach().cpu().float().numpy()
    n = xyz.shape[0]
    f_dc = gs.features_dc_.detach().transpose(1, 2).flatten(1).cpu().float().numpy()
    f_rest = gs.features_rest_.detach().transpose(1, 2).flatten(1).cpu().float().numpy() if gs.cfg.sh_degree > 0 else np.zeros((n, 0), np.float32)
    opa = gs.opacity_.detach().cpu().float().numpy().reshape(n, 1)
    scale = gs.scaling_.detach().cpu().float().numpy().copy()
    scale[:, 2] = math.log(1e-6)
    rot = gs.quaternion_.detach().cpu().float().numpy()
    names = (["x", "y", "z"] + [f"f_dc_{i}" for i in range(f_dc.shape[1])] + [f"f_rest_{i}" for i in range(f_rest.shape[1])]
             + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    data = np.concatenate([xyz, f_dc, f_rest, opa, scale, rot], 1).astype("<f4")
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {k}\n" for k in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())


def load_ply_to_gs(path, cfg=None, device="cpu", spatial_scale=1.0):
    """Inverse of export_gs_to_ply (anchors = xyz, offsets = 0, as the reference's loader does)."""
    import numpy as np
    with open(path, "rb") as f:
        names, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property float"):
                names.append(line.split()[-1])
            elif line.startswith("format") and "binary_little_endian" not in line:
                raise RuntimeError("load_ply_to_gs: only binary_little_endian PLY is supported")
            elif line == "end_header":
                break
        data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    col = {k: i for i, k in enumerate(names)}
    t = lambda keys: torch.from_numpy(np.ascontiguousarray(data[:, [col[k] for k in keys]])).to(device)
    n_dc = sum(k.startswith("f_dc_") for k in names)
    n_rest = sum(k.startswith("f_rest_") for k in names)
    f_dc = t([f"f_dc_{i}" for i in range(n_dc)]).reshape(n, 3, -1).transpose(1, 2).contiguous()
    f_rest = (t([f"f_rest_{i}" for i in range(n_rest)]).reshape(n, 3, -1).transpose(1, 2).contiguous() if n_rest
              else torch.zeros(n, 0, 3, device=device))
    cfg = cfg or GSConfig(sh_degree=int(round(math.sqrt(1 + n_rest // 3))) - 1)
    return NeuralGS(t(["x", "y", "z"]), t(["scale_0", "scale_1", "scale_2"]), t(["rot_0", "rot_1", "rot_2", "rot_3"]),
                    t(["opacity"]).reshape(-1), f_dc, f_rest, cfg, spatial_scale)


def sample_ray_batch(local_map, origin, direction, depth, sample_std, truncated_dis, surface_sample_num=3, free_sample_num=3,
                     sample_free=True, generator=None):
    """NeuralSLAM::sample (neural_mapping.cpp:73-104) with the occupancy structure: per ray one sample in every occupied
    voxel it crosses (LocalMap::sample, octree ray march) [+ free-space samples], the near-surface samples, SDF targets
    truncated to +-truncated_dis, the ray end points, all filtered to the map's inner cube.
    -> DepthSamples(xyz [B,3], ray_sdf [B,1], ridx [B], direction, origin, depth)."""
    from .sdf import DepthSamples
    n = origin.shape[0]
    rays = DepthSamples(origin=origin, direction=direction, depth=depth, xyz=origin + direction * depth,
                        ray_sdf=torch.zeros(n, 1, device=origin.device), ridx=torch.arange(n, device=origin.device))
    pts = local_map.sample(rays, 1, sample_free, free_sample_num, generator)
    sx, ss, sr = sample_surface_pts(origin, direction, depth, surface_sample_num, sample_std, generator)
    surf = rays.index_select(sr)
    surf.xyz, surf.ray_sdf, surf.depth = sx, ss, surf.depth - ss
    pts = pts.cat(surf)
    pts.ray_sdf = torch.where(pts.ray_sdf.abs() > truncated_dis, pts.ray_sdf.sign() * truncated_dis, pts.ray_sdf)
    pts = pts.cat(rays)
    return pts.index_select(local_map.get_inrange_mask(pts.xyz).nonzero().reshape(-1))


def isotropic_loss(scale, gaussian_ids):
    """neural_mapping.cpp:267-276: mean |s - mean(s)| over the two in-plane scales of the visible splats."""
    s = scale.index_select(0, gaussian_ids)[..., 0:2]
    return (s - s.mean(-1, True)).abs().mean()
