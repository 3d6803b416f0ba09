"""View-parallel training plumbing: replicated splat parameters, one view per GPU per step,
ONE RCCL all-reduce of the dense parameter gradients over xGMI.

`SplatParams` mirrors the parameter set and activations of the reference's `NeuralGS`
(/root/reference/include/neural_gaussian/neural_gaussian.cpp:426-453 parameter groups; :463-492 activations):
anchors (no grad) + offsets, log-scales, quaternions, logit-opacities, SH dc/rest.  All trainable tensors are
views into ONE flat buffer and their .grad are views into ONE flat gradient buffer, so the collective is a
single large all-reduce (xGMI is point-to-point: few large messages, not many small ones) and no
flatten/unflatten copies are needed.  The reference has no multi-GPU path at all: this is new.
"""
import torch


class _Activate(torch.autograd.Function):
    """(offsets, scaling, opacity) -> (xyz, scales, opacities) in one launch; the backward accumulates straight into the
    three parameters' gradient buffers (`sinks`, views of the flat gradient buffer) and returns no gradient to autograd."""

    @staticmethod
    def forward(ctx, anchors, offsets, scaling, opacity, sinks):
        from . import capi
        n = anchors.shape[0]
        xyz, scales = torch.empty_like(anchors), torch.empty_like(scaling)
        opac = torch.empty(n, dtype=torch.float32, device=anchors.device)
        capi.check(capi.lib().gsdf_splat_activations_fwd(n, capi.f32(anchors), capi.f32(offsets), capi.f32(scaling),
                                                         capi.f32(opacity), capi.f32(xyz), capi.f32(scales), capi.f32(opac),
                                                         capi.stream()), "splat_activations_fwd")
        ctx.save_for_backward(scales, opac)
        ctx.sinks = sinks
        return xyz, scales, opac

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v_xyz, v_scales, v_opac):
        from . import capi
        scales, opac = ctx.saved_tensors
        c = lambda g: None if g is None else g.contiguous()
        g_off, g_sc, g_op = ctx.sinks
        capi.check(capi.lib().gsdf_splat_activations_bwd(opac.shape[0], capi.f32(scales), capi.f32(opac), capi.f32(c(v_xyz)),
                                                         capi.f32(c(v_scales)), capi.f32(c(v_opac)), capi.f32(g_off),
                                                         capi.f32(g_sc), capi.f32(g_op), capi.stream()), "splat_activations_bwd")
        return None, None, None, None, None


def morton_order(xyz, bits=16):
    """Permutation that sorts points [n,3] by the Morton (Z-order) key of their position in their own bounding box,
    `bits` bits per axis.  The splat set is kept in this order in HBM (at initialisation and whenever a refinement step
    re-materialises it): packed visible rows, the GS<->SDF sample points and the tile lists' gathers are then spatially
    coherent — the hash-grid gathers of neighbouring sample points hit the same L2 lines and the binned table-gradient scatter
    emits long runs.  The reference attaches no meaning to the order of the splats (it appends and removes rows at every
    refinement step, neural_gaussian.cpp:690-926)."""
    lo, hi = xyz.min(0).values, xyz.max(0).values
    q = ((xyz - lo) / (hi - lo).clamp_min(1e-20) * (2 ** bits - 1)).round().to(torch.int64).clamp_(0, 2 ** bits - 1)

    def spread(x):          # 21-bit value -> every third bit of a 63-bit word
        x = (x | (x << 32)) & 0x1F00000000FFFF
        x = (x | (x << 16)) & 0x1F0000FF0000FF
        x = (x | (x << 8)) & 0x100F00F00F00F00F
        x = (x | (x << 4)) & 0x10C30C30C30C30C3
        return (x | (x << 2)) & 0x1249249249249249
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(key, stable=True).indices


class SplatParams:
    FIELDS = (("offsets", 3), ("scaling", 3), ("quaternion", 4), ("opacity", 1), ("features_dc", 3), ("features_rest", None))

    def __init__(self, anchors, offsets, scaling, quaternion, opacity, features_dc, features_rest):
        dev = anchors.device
        N = anchors.shape[0]
        self.anchors = anchors.contiguous()                                   # registered without grad in the reference
        parts = dict(offsets=offsets, scaling=scaling, quaternion=quaternion, opacity=opacity.reshape(N, 1),
                     features_dc=features_dc.reshape(N, -1), features_rest=features_rest.reshape(N, -1))
        total = sum(p.numel() for p in parts.values())
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = {}, 0
        for name, p in parts.items():
            n = p.numel()
            v = self.flat[off:off + n].view(p.shape)
            v.copy_(p)
            v.requires_grad_(True)
            v.grad = self.flat_grad[off:off + n].view(p.shape)
            self.views[name] = v
            off += n
        self.n_rest = features_rest.reshape(N, -1).shape[1] // 3

    @classmethod
    def from_scene(cls, sc, dev, order=None):
        """order: None (rows as given) or an int64 permutation (e.g. morton_order(sc["means"])) applied to every field."""
        N = sc["means"].shape[0]
        sh = sc["sh"]
        o = (lambda t: t) if order is None else (lambda t: t.index_select(0, order.to(t.device)))
        return cls(o(sc["means"]).to(dev), torch.zeros(N, 3, device=dev), o(sc["log_scales"]).to(dev), o(sc["quats"]).to(dev),
                   o(sc["logit_opacities"]).to(dev), o(sh[:, :1]).contiguous().to(dev), o(sh[:, 1:]).contiguous().to(dev))

    def activated(self):
        """generate_gaussian(): xyz = anchors+offsets, scales = exp, opacity = sigmoid, sh = cat(dc, rest)."""
        v = self.views
        N = self.anchors.shape[0]
        if self.anchors.is_cuda and torch.is_grad_enabled():
            # one launch forward, one backward that accumulates into the flat gradient buffer (include/gsdf_hip.h, a2)
            xyz, scales, opacity = _Activate.apply(self.anchors, v["offsets"], v["scaling"], v["opacity"],
                                                   (v["offsets"].grad, v["scaling"].grad, v["opacity"].grad))
        else:       # host tensors: only the CPU tests of the collective plumbing (gloo) come through here
            xyz = self.anchors + v["offsets"]
            scales = torch.exp(v["scaling"])
            opacity = torch.sigmoid(v["opacity"]).reshape(N)
        dc = v["features_dc"].reshape(N, 1, 3)
        sh = dc if self.n_rest == 0 else torch.cat([dc, v["features_rest"].reshape(N, self.n_rest, 3)], 1)
        return xyz, v["quaternion"], scales, opacity, sh

    def parameters(self):
        return list(self.views.values())

    # ---- refinement-step surgery on the flat buffers (neural_gaussian.cpp:690-926 + optimizer_utils.cpp:5-165) -------------
    def _bind(self, flat, flat_grad, n):
        """views (and their .grad views) over a new pair of flat buffers holding n rows per field"""
        widths = [v.shape[1] if v.dim() > 1 else 1 for v in self.views.values()]
        off = 0
        for (name, old), w in zip(list(self.views.items()), widths):
            v = flat[off:off + n * w].view(n, w)
            v.requires_grad_(True)
            v.grad = flat_grad[off:off + n * w].view(n, w)
            self.views[name] = v
            off += n * w
        self.flat, self.flat_grad = flat, flat_grad

    @staticmethod
    def _gather_rows(widths, n_src, n_dst, keep_idx, src, dst):
        """dst rows [0, n_keep) of every field = src rows keep_idx (None: all n_src rows)."""
        n_keep = n_src if keep_idx is None else int(keep_idx.numel())
        if src.is_cuda:
            import ctypes as C
            from . import capi
            capi.check(capi.lib().gsdf_flat_rows_gather(len(widths), (C.c_int32 * len(widths))(*widths), n_src, n_dst, n_keep,
                                                        capi.ptr(None if keep_idx is None else keep_idx.contiguous(), torch.int64),
                                                        capi.f32(src), capi.f32(dst), capi.stream()), "flat_rows_gather")
            return
        so = do = 0                                       # host tensors (CPU tests of the policy): the same thing in torch
        for w in widths:
            rows = src[so:so + n_src * w].view(n_src, w)
            dst[do:do + n_keep * w].view(n_keep, w).copy_(rows if keep_idx is None else rows.index_select(0, keep_idx))
            so, do = so + n_src * w, do + n_dst * w

    @torch.no_grad()
    def resize(self, keep_idx=None, ext=None, optimizer=None, group=0):
        """The splat set becomes: rows `keep_idx` of the current one (int64 index tensor; None = all rows, in order) followed by
        the rows of `ext` = {field: [n_ext, ...]} (None = nothing appended; every field must then be given).  The flat
        parameter buffer, the flat gradient buffer (zeroed) and — when `optimizer` (FusedAdam) is given — both Adam moments of
        `group` are rebuilt with ONE gather launch each: surviving rows keep their moments, appended rows start from zero
        moments (optimizer_utils.cpp:5-165); the step count is kept, as torch::optim::Adam's per-parameter state is there."""
        names = list(self.views)
        widths = [self.views[k].shape[1] for k in names]
        n_src = self.views[names[0]].shape[0]         # (the caller may already have replaced the anchors)
        n_keep = n_src if keep_idx is None else int(keep_idx.numel())
        n_ext = 0 if not ext else int(next(iter(ext.values())).shape[0])
        n_dst, wt = n_keep + n_ext, sum(widths)
        dev = self.flat.device
        new_flat = torch.empty(n_dst * wt, dtype=torch.float32, device=dev)
        self._gather_rows(widths, n_src, n_dst, keep_idx, self.flat, new_flat)
        if n_ext:
            off = 0
            for k, w in zip(names, widths):
                if w:
                    new_flat[off + n_keep * w:off + n_dst * w].view(n_ext, w).copy_(ext[k].reshape(n_ext, w))
                off += n_dst * w
        if optimizer is not None:
            g = optimizer.groups[group]
            moments = []
            for key in ("m", "v"):
                buf = torch.zeros(n_dst * wt, dtype=torch.float32, device=dev)
                self._gather_rows(widths, n_src, n_dst, keep_idx, g[key], buf)
                moments.append(buf)
        new_grad = torch.zeros(n_dst * wt, dtype=torch.float32, device=dev)
        self._bind(new_flat, new_grad, n_dst)
        if optimizer is not None:
            optimizer.replace_group(group, new_flat, new_grad, moments[0], moments[1], [n_dst * w for w in widths])
        return n_dst


class ViewParallel:
    """Gradient synchronisation for view-parallel training.  `dist` is torch.distributed (backend nccl == RCCL
    on ROCm, gloo in the CPU tests) or None for a single process."""

    def __init__(self, params, dist=None, extra_groups=()):
        self.params, self.dist = params, dist
        self.groups = [params] + list(extra_groups)
        self.world = dist.get_world_size() if dist is not None else 1

    def zero_grad(self):
        for g in self.groups:
            g.flat_grad.zero_()

    def all_reduce_group_async(self, group):
        """Starts the all-reduce of ONE parameter family as soon as its gradients are final, so that it overlaps
        with the rest of the backward pass (the SDF network's 61 MB hide under the splat rasteriser's backward).
        Completed by finish()."""
        if self.dist is None or self.world == 1:
            return
        self._pending = getattr(self, "_pending", [])
        self._pending.append((group, self.dist.all_reduce(group.flat_grad, op=self.dist.ReduceOp.SUM, async_op=True)))

    def finish(self):
        for group, work in getattr(self, "_pending", []):
            work.wait()
            group.flat_grad.mul_(1.0 / self.world)
        self._pending = []

    def all_reduce_group(self, group, collective="all_reduce", extra_sum=()):
        """Mean of ONE parameter family's gradients over the ranks, on the CURRENT stream (RCCL's internal stream is
        ordered after it and the current stream after RCCL): called from the stream of the leg that owns the family, so
        that the other leg's kernels keep running beside the collective.

        collective = "all_reduce": one all-reduce of the flat buffer.
        collective = "reduce_scatter_all_gather": the same sum as an explicit reduce-scatter (every rank reduces 1/G of the
            buffer) + all-gather; on xGMI (point-to-point links, no switch) both phases keep all 7 links of a GPU busy with
            1/G-sized pieces, and the 1/G scaling runs on the shard only.  Results are identical to the all-reduce's up to
            the order of the fp32 additions.
        extra_sum: tensors summed over the ranks IN THE SAME MESSAGE (the refine statistics grad2d / count at a refinement
            step: SURVEY 8e's second collective rides on the first); they are updated in place, unscaled."""
        if self.dist is None or self.world == 1:
            return
        G, flat = self.world, group.flat_grad
        extra = [t for t in extra_sum if t is not None]
        if extra:
            buf = torch.cat([flat] + [t.reshape(-1).to(flat.dtype) for t in extra])
        else:
            buf = flat
        if collective == "reduce_scatter_all_gather":
            n = buf.numel()
            per = (n + G - 1) // G
            if per * G != n:
                buf = torch.cat([buf, buf.new_zeros(per * G - n)])
            shard = torch.empty(per, dtype=buf.dtype, device=buf.device)
            self.dist.reduce_scatter_tensor(shard, buf, op=self.dist.ReduceOp.SUM)
            self.dist.all_gather_into_tensor(buf, shard)
            buf = buf[:n]
        elif collective == "all_reduce":
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        else:
            raise ValueError(f"unknown collective {collective!r}")
        if buf is not flat:
            flat.copy_(buf[:flat.numel()])
            off = flat.numel()
            for t in extra:
                t.copy_(buf[off:off + t.numel()].view_as(t))
                off += t.numel()
        flat.mul_(1.0 / G)

    def sync_refine_state(self, state, sums=True, maxs=True):
        """The second, small collective of view-parallel training: before a refine step every rank must
        hold the SAME densification statistics so that duplicate / split / prune take identical decisions everywhere
        (the reference accumulates them in NeuralGS::update_state, neural_gaussian.cpp:626-680): `grad2d` and `count` are
        summed over the ranks' views, `vis` and `radii` are max-merged.  Two messages: one SUM, one MAX.
        The SUM must only be taken right before the statistics are consumed AND reset (a sum of sums would count twice);
        the MAX is idempotent and can be taken at any time."""
        if self.dist is None or self.world == 1:
            return
        sum_keys = [k for k in ("grad2d", "count") if k in state] if sums else []
        max_keys = [k for k in ("vis", "radii") if k in state] if maxs else []
        for keys, op in ((sum_keys, self.dist.ReduceOp.SUM), (max_keys, self.dist.ReduceOp.MAX)):
            if not keys:
                continue
            buf = torch.cat([state[k].reshape(-1) for k in keys])
            self.dist.all_reduce(buf, op=op)
            off = 0
            for k in keys:
                n = state[k].numel()
                state[k].copy_(buf[off:off + n].view_as(state[k]))
                off += n

    def all_reduce_grads(self):
        if self.dist is None or self.world == 1:
            return
        # mean over the G views of the step (effective batch G; SURVEY 8e "semantics change to report").
        # One large message per parameter family: splats (56 B/splat) and the SDF network (61 MB table + MLP).
        for g in self.groups:
            self.dist.all_reduce(g.flat_grad, op=self.dist.ReduceOp.SUM)
            g.flat_grad.mul_(1.0 / self.world)


class _InjectGrads(torch.autograd.Function):
    """sum_k <x_k, g_k> as ONE autograd node whose backward hands the fixed upstream gradients g_k straight to the
    producers of x_k (no per-term mul / sum / expand kernels).  Benchmark plumbing: op-level upstream gradients on the
    render outputs that the photometric loss does not touch."""

    @staticmethod
    def forward(ctx, n, *tensors_and_grads):
        xs, gs = tensors_and_grads[:n], tensors_and_grads[n:]
        ctx.gs = gs
        return torch.zeros((), dtype=xs[0].dtype, device=xs[0].device)     # the value is not used by anything

    @staticmethod
    def backward(ctx, v):
        # v is 1 for a loss that is a plain sum of terms (asserted by the caller's test, not here: no host sync)
        return (None,) + tuple(ctx.gs) + (None,) * len(ctx.gs)


def inject_grads(pairs):
    """pairs = [(tensor, upstream_gradient), ...] -> scalar 0 whose backward delivers the given gradients (times the
    incoming gradient, which must be 1: add the result to the loss unscaled)."""
    xs, gs = [p[0] for p in pairs], [p[1] for p in pairs]
    return _InjectGrads.apply(len(xs), *xs, *gs)


class GradGate:
    """Carries the HIP event that marks 'the gradient that flows in here is complete' from the stream that
    produces it to the stream that consumes it (see join_grad)."""

    def __init__(self):
        self.event = None


class _JoinGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        ctx.gate = gate
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.gate.event is not None:
            torch.cuda.current_stream().wait_event(ctx.gate.event)     # device-side wait, the host does not block
            ctx.gate.event = None
        return g, None


def join_grad(x, gate):
    """Identity on `x`.  In the backward pass the stream that consumes x's gradient first waits (on the device) for
    `gate.event`.  Used to run the SDF leg of a training step (hash grid + MLP, bound by the memory-side atomic units) on
    a second HIP stream concurrently with the splat rasteriser (bound by VALU issue): the two legs only meet where the
    gradient of the splat sample points enters the projection backward, and this node sits exactly there."""
    return _JoinGrad.apply(x, gate)


def flatten_leaves(tensors):
    """Re-homes a list of leaf parameters into ONE flat buffer (+ one flat gradient buffer) and returns
    (flat, flat_grad, views).  The views replace the original leaves (same values, same shapes)."""
    dev = tensors[0].device
    total = sum(t.numel() for t in tensors)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
    views, off = [], 0
    for t in tensors:
        n = t.numel()
        v = flat[off:off + n].view(t.shape)
        v.copy_(t.detach())
        v.requires_grad_(True)
        v.grad = flat_grad[off:off + n].view(t.shape)
        views.append(v)
        off += n
    return flat, flat_grad, views


class FlatGroup:
    """Any flat (params, grads) pair that takes part in the per-step all-reduce (e.g. the SDF network)."""

    def __init__(self, flat, flat_grad):
        self.flat, self.flat_grad = flat, flat_grad


class FusedAdam:
    """torch.optim.Adam semantics over flat (params, grads) groups, ONE HIP launch per group
    (include/gsdf_hip.h: gsdf_adam_step).  `segments` = [(n_elements, lr), ...] in buffer order, e.g. the reference's
    six splat groups with their learning rates (neural_gaussian.cpp:434-453)."""

    def __init__(self, betas=(0.9, 0.999), eps=1e-15):
        self.betas, self.eps, self.groups, self.t = betas, eps, [], 0

    def add_group(self, flat, flat_grad, segments):
        import ctypes as C
        begins, off = [], 0
        for n, _ in segments:
            begins.append(off)
            off += n
        assert off == flat.numel(), (off, flat.numel())
        self.groups.append(dict(flat=flat, grad=flat_grad, m=torch.zeros_like(flat), v=torch.zeros_like(flat),
                                begins=(C.c_int64 * len(begins))(*begins), lrs=[lr for _, lr in segments]))
        return len(self.groups) - 1

    def set_lr(self, group, segment, lr):
        self.groups[group]["lrs"][segment] = float(lr)

    def replace_group(self, group, flat, flat_grad, m, v, segment_sizes):
        """New buffers for a group after a refinement step changed the number of rows (SplatParams.resize); learning rates
        and the step count are kept."""
        import ctypes as C
        g = self.groups[group]
        assert len(segment_sizes) == len(g["lrs"]) and sum(segment_sizes) == flat.numel()
        begins, off = [], 0
        for n in segment_sizes:
            begins.append(off)
            off += n
        g.update(flat=flat, grad=flat_grad, m=m, v=v, begins=(C.c_int64 * len(begins))(*begins))

    def zero_segment_moments(self, group, segment):
        """reset_opacity (neural_gaussian.cpp:918-926): the moments of one parameter of the group start over."""
        g = self.groups[group]
        b = list(g["begins"]) + [g["flat"].numel()]
        g["m"][b[segment]:b[segment + 1]].zero_()
        g["v"][b[segment]:b[segment + 1]].zero_()

    @torch.no_grad()
    def step(self, zero_grad: bool = False):
        """zero_grad: the gradient buffers are zeroed by the same launch that consumes them (gsdf_adam_step_zero_grad)"""
        import ctypes as C
        from . import capi
        L = capi.lib()
        self.t += 1
        fn = L.gsdf_adam_step_zero_grad if zero_grad else L.gsdf_adam_step
        for g in self.groups:
            lrs = (C.c_float * len(g["lrs"]))(*g["lrs"])
            capi.check(fn(g["flat"].numel(), len(g["lrs"]), g["begins"], lrs, capi.f32(g["flat"]), capi.f32(g["grad"]),
                                        capi.f32(g["m"]), capi.f32(g["v"]), self.betas[0], self.betas[1], self.eps, self.t,
                                        capi.stream()), "adam_step")
