"""Host-side mirror of the reference's SDF network interface on top of the C ABI.

Mirrors, with the same names and behaviour,
  * `TCNNEncoding` / `TCNNNetwork` of the (absent) tcnn_binding submodule as the reference uses them
    (/root/reference/include/neural_net/encoding_map.cpp:15-26,59; local_map.cpp:44-55,94): objects with a
    flat fp32 leaf `params_`, `forward(x)`, `get_out_dim()`; the encoding supports first AND second order
    autograd (the reference calls torch::autograd::grad(..., create_graph=true), local_map.cpp:151-172);
  * `LocalMap::get_sdf / get_gradient` (local_map.cpp:87-173) and the coordinate normalisation of
    `SubMap::xyz_to_zp1_pts` (sub_map.cpp:82-97);
  * the SDF losses of include/optimizer/loss/loss.cpp:7-11,49-90.
"""
import ctypes as C
import math
import os

import torch

from . import capi
from .capi import f32, ptr
from .ops import _timed
from .streams import launch_on

GRID_DEFAULT = dict(otype="Grid", type="Hash", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                    base_resolution=32, per_level_scale=2.0, interpolation="Linear")


def _gcfg(cfg):
    return (int(cfg["n_levels"]), int(cfg["n_features_per_level"]), int(cfg["log2_hashmap_size"]),
            int(cfg["base_resolution"]), float(cfg["per_level_scale"]))


class _SinkState:
    armed = 0


class grad_sinks_armed:
    """Context manager around the trainer's `loss.backward()`.  The in-place gradient sinks installed by
    LocalMap.flatten(accumulate_table_grad_in_place=True) deposit parameter gradients straight into the flat gradient
    buffer; they do so ONLY inside this context.  Any other traversal of the graph (torch.autograd.grad w.r.t. inputs,
    create_graph=True for the analytic eikonal term, ...) gets ordinary autograd behaviour: the parameter gradients are
    returned to autograd and nothing is written behind its back."""

    def __enter__(self):
        _SinkState.armed += 1
        return self

    def __exit__(self, *exc):
        _SinkState.armed -= 1
        return False


def _sinks_live():
    return _SinkState.armed > 0 and not torch.is_grad_enabled()


BINNED_MIN_POINTS = 24576     # measured crossover on MI355X: 14 K points 0.058 (atomic) / 0.069 ms, 28 K points 0.109 / 0.084 ms


def stencil_merge_levels(cfg, delta_unit):
    """Number of coarse levels at which the +-delta stencil points of a sample usually share the sample's grid cell
    (scale_l * delta < 1, scale_l = base_resolution * per_level_scale^l - 1): the levels the binned scatter checks for merging."""
    n_levels, _, _, base_res, pls = cfg
    return sum(1 for l in range(n_levels) if (base_res * pls ** l - 1.0) * float(delta_unit) < 1.0)


def _stencil_fwd(stencil, B):
    """forward of a stencil batch (n base rows + 6 blocks of n central-difference rows) through the group-walking kernel"""
    return stencil is not None and stencil[0] > 0 and stencil[0] * 7 == B and os.environ.get("GSDF_STENCIL_FWD", "1") != "0"


def scatter_table_grad(B, cfg, x, table, v_feat, v_table, v_x=None, stencil=None):
    """v_table += (d feat / d table)^T v_feat [and v_x = (d feat / d x)^T v_feat].  Large batches take the binned scatter
    (include/gsdf_hip.h: gsdf_hashgrid_bwd_binned, no global atomics); GSDF_HASHGRID_BINNED=0/1 forces one path (tests).
    stencil = (n, merge_levels): the batch is n base rows + 6 blocks of n central-difference rows (query_points layout)."""
    L = capi.lib()
    mode = os.environ.get("GSDF_HASHGRID_BINNED", "auto")
    nbytes = 0
    if v_table is not None and mode != "0" and (mode == "1" or B >= BINNED_MIN_POINTS or L.gsdf_deterministic(-1)):
        nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(B, *cfg)
    if nbytes:
        if v_x is not None:
            capi.check(_timed("hashgrid_bwd_input", L.gsdf_hashgrid_bwd, B, *cfg, f32(x), f32(table), f32(v_feat), None,
                              f32(v_x), capi.stream()), "hashgrid_bwd")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        sn, ml = stencil if (stencil is not None and stencil[0] * 7 == B and os.environ.get("GSDF_STENCIL_MERGE", "1") != "0") else (0, 0)
        capi.check(_timed("hashgrid_bwd", L.gsdf_hashgrid_bwd_binned_stencil, B, sn, ml, *cfg, f32(x), f32(v_feat), f32(v_table),
                          ptr(ws), nbytes, capi.stream()), "hashgrid_bwd_binned")
    else:
        capi.check(_timed("hashgrid_bwd" if v_table is not None else "hashgrid_bwd_input", L.gsdf_hashgrid_bwd, B, *cfg,
                          f32(x), f32(table), f32(v_feat), f32(v_table), f32(v_x), capi.stream()), "hashgrid_bwd")


class _GridBwd(torch.autograd.Function):
    """(v_feat, x, table) -> (v_x, v_table): the encoding's backward as a differentiable op, so that
    grad-of-grad (eikonal on the analytic SDF gradient) works."""

    @staticmethod
    def forward(ctx, v_feat, x, table, cfg, want_table, grad_sink=None, scatter_stream=None, want_x=True, stencil=None):
        L = capi.lib()
        B = x.shape[0]
        v_feat = v_feat.contiguous()
        v_x = torch.empty_like(x) if want_x else None
        empty = lambda: torch.zeros(0, device=x.device)

        def launch(vt, vx):
            scatter_table_grad(B, cfg, x, table, v_feat, vt, vx, stencil)

        if want_table and grad_sink is not None:
            # accumulate straight into the parameter's (pre-zeroed) gradient buffer: the kernel's atomics already
            # ACCUMULATE, so the 61 MB zero-fill + the autograd add per call disappear; autograd sees no table grad
            v_table = grad_sink.view(table.shape)
            if scatter_stream is None:
                launch(v_table, v_x)
            else:
                # the input gradient (no atomics) here; the scatter (atomics only) ASYNCHRONOUSLY on its own stream:
                # whoever consumes the table gradient waits for `scatter_stream`
                if want_x:
                    launch(None, v_x)
                scatter_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(scatter_stream):
                    launch(v_table, None)
                v_feat.record_stream(scatter_stream)
                x.record_stream(scatter_stream)
            v_table = None
        else:
            v_table = torch.zeros_like(table) if want_table else None
            if v_table is not None or v_x is not None:
                launch_on(scatter_stream, lambda: launch(v_table, v_x))
        ctx.save_for_backward(v_feat, x, table)
        ctx.cfg = cfg
        if v_table is None:
            v_table = empty()
            ctx.mark_non_differentiable(v_table)
        if v_x is None:
            v_x = empty()
            ctx.mark_non_differentiable(v_x)
        return v_x, v_table

    @staticmethod
    def backward(ctx, vv_x, _vv_table):
        # second order w.r.t. the table gradient output is never requested by the reference
        v_feat, x, table = ctx.saved_tensors
        need_vf, need_x, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if vv_x is None:
            return None, None, None, None, None, None, None, None, None
        if torch.is_grad_enabled():      # create_graph=True: the double backward stays on the graph (a loss on the analytic Hessian)
            g_vfeat, g_x, g_table = _GridBwd2.apply(vv_x, v_feat, x, table, ctx.cfg, need_vf, need_x, need_t)
            return (g_vfeat if need_vf else None), (g_x if need_x else None), (g_table if need_t else None), None, None, None, None, None, None
        g_vfeat, g_x, g_table = _grid_double_backward(ctx.cfg, vv_x, v_feat, x, table, need_vf, need_x, need_t)
        return g_vfeat, g_x, g_table, None, None, None, None, None, None


def _grid_double_backward(cfg, vv_x, v_feat, x, table, need_vf, need_x, need_t):
    L = capi.lib()
    B = x.shape[0]
    g_vfeat = torch.empty_like(v_feat) if need_vf else None
    g_x = torch.empty_like(x) if need_x else None
    g_table = torch.zeros_like(table) if need_t else None
    vv_x = vv_x.contiguous()
    # table part: large batches take the binned scatter's second-order form (no global atomics), like the first order
    nbytes = 0
    if need_t and os.environ.get("GSDF_HASHGRID_BINNED", "auto") != "0" and \
            (os.environ.get("GSDF_HASHGRID_BINNED") == "1" or B >= BINNED_MIN_POINTS or L.gsdf_deterministic(-1)):
        nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(B, *cfg)
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        capi.check(_timed("hashgrid_bwd_bwd_table", L.gsdf_hashgrid_bwd_binned2, B, *cfg, f32(x), None, f32(v_feat), f32(vv_x),
                          f32(g_table), ptr(ws), nbytes, capi.stream()), "hashgrid_bwd_binned2")
    if need_vf or need_x or (need_t and not nbytes):
        capi.check(_timed("hashgrid_bwd_bwd", L.gsdf_hashgrid_bwd_bwd, B, *cfg, f32(x), f32(table), f32(v_feat),
                          f32(vv_x), f32(g_vfeat), f32(None if nbytes else g_table), f32(g_x), capi.stream()), "hashgrid_bwd_bwd")
    return g_vfeat, g_x, g_table


class _GridBwd2(torch.autograd.Function):
    """(vv_x, v_feat, x, table) -> (g_vfeat, g_x, g_table): the encoding's double backward as an operator of its own; its backward is the
    THIRD order (include/gsdf_hip.h: gsdf_hashgrid_bwd_bwd_bwd) — LocalMap::get_gradient(hessian=True, numerical_grad=False) followed by
    loss::curvate_loss, /root/reference/include/neural_net/local_map.cpp:151-168, include/neural_mapping/neural_mapping.cpp:117-121."""

    @staticmethod
    def forward(ctx, vv_x, v_feat, x, table, cfg, need_vf, need_x, need_t):
        g_vfeat, g_x, g_table = _grid_double_backward(cfg, vv_x, v_feat, x, table, need_vf, need_x, need_t)
        ctx.save_for_backward(vv_x.contiguous(), v_feat, x, table)
        ctx.cfg = cfg
        out, dead = [], []
        for i, t in enumerate((g_vfeat, g_x, g_table)):
            if t is None:
                t = torch.zeros(0, device=x.device)
            if t.numel() == 0 or i == 2:      # (a loss on the TABLE gradient of the double backward is nobody's path)
                dead.append(t)
            out.append(t)
        ctx.mark_non_differentiable(*dead)
        return tuple(out)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, mu, lam, _nu):
        vv_x, v_feat, x, table = ctx.saved_tensors
        if mu is None and lam is None:
            return (None,) * 8
        L = capi.lib()
        B = x.shape[0]
        lam = torch.zeros_like(x) if lam is None else lam.contiguous()
        mu = None if mu is None else mu.contiguous()
        need = ctx.needs_input_grad
        t_vv = torch.empty_like(vv_x) if need[0] else None
        t_vfeat = torch.empty_like(v_feat) if need[1] else None
        t_x = torch.empty_like(x) if need[2] else None
        t_table = torch.zeros_like(table) if need[3] else None
        capi.check(_timed("hashgrid_bwd_bwd_bwd", L.gsdf_hashgrid_bwd_bwd_bwd, B, *ctx.cfg, f32(x), f32(table), f32(v_feat), f32(vv_x), f32(lam),
                          f32(mu), f32(t_vfeat), f32(t_table), f32(t_vv), f32(t_x), capi.stream()), "hashgrid_bwd_bwd_bwd")
        return t_vv, t_vfeat, t_x, t_table, None, None, None, None


class _GridFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, cfg, grad_sink=None, scatter_stream=None, save_jacobian=False, stencil=None):
        L = capi.lib()
        x, table = x.contiguous(), table.contiguous()
        B = x.shape[0]
        feat = torch.empty(B, cfg[0] * cfg[1], dtype=torch.float32, device=x.device)
        jac = None
        if save_jacobian and ctx.needs_input_grad[0] and cfg[0] * cfg[1] <= 32:
            jac = torch.empty(B, cfg[0] * cfg[1], 3, dtype=torch.float32, device=x.device)
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd_jac, B, *cfg, f32(x, "x"), f32(table, "params"), f32(feat),
                              f32(jac), capi.stream()), "hashgrid_fwd_jac")
        elif _stencil_fwd(stencil, B):
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd_stencil, B, stencil[0], 0, *cfg, f32(x, "x"), f32(table, "params"),
                              f32(feat), None, capi.stream()), "hashgrid_fwd_stencil")
        else:
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd, B, *cfg, f32(x, "x"), f32(table, "params"), f32(feat),
                              capi.stream()), "hashgrid_fwd")
        ctx.save_for_backward(x, table, jac)
        ctx.cfg = cfg
        ctx.grad_sink, ctx.scatter_stream, ctx.stencil = grad_sink, scatter_stream, stencil
        return feat

    @staticmethod
    def backward(ctx, v_feat):
        x, table, jac = ctx.saved_tensors
        want_x, v_x = bool(ctx.needs_input_grad[0]), None
        if jac is not None and want_x and not torch.is_grad_enabled():
            # first-order fast path: d/dx from the Jacobian saved by the forward, only the scatter touches the table
            v_feat = v_feat.contiguous()
            v_x = torch.empty_like(x)
            capi.check(_timed("hashgrid_bwd_input", capi.lib().gsdf_hashgrid_bwd_jac, x.shape[0], ctx.cfg[0], ctx.cfg[1],
                              f32(jac), f32(v_feat), f32(v_x), capi.stream()), "hashgrid_bwd_jac")
            want_x = False
        sink = ctx.grad_sink if _sinks_live() else None
        g_x, v_table = _GridBwd.apply(v_feat, x, table, ctx.cfg, bool(ctx.needs_input_grad[1]), sink,
                                      ctx.scatter_stream if sink is not None else None, want_x, ctx.stencil)
        v_x = g_x if want_x else v_x
        want_t = ctx.needs_input_grad[1] and sink is None
        return v_x, (v_table if want_t else None), None, None, None, None, None


class TCNNEncoding:
    """tcnn_binding's TCNNEncoding as the reference uses it: `TCNNEncoding(3, encoding_config, name)`,
    `.params_` (flat fp32 leaf, registered as an nn parameter by the caller, local_map.cpp:73-75),
    `.forward(x)` with x in [0,1]^3, `.get_out_dim()`."""

    def __init__(self, n_input_dims=3, config=None, name="encoder", device="cuda", seed=None):
        cfg = dict(GRID_DEFAULT if config is None else config)
        if cfg.get("otype", "Grid") != "Grid" or cfg.get("type", "Hash") != "Hash" or cfg.get("interpolation", "Linear") != "Linear":
            raise RuntimeError("TCNNEncoding: only {otype: Grid, type: Hash, interpolation: Linear} is implemented")
        if n_input_dims != 3:
            raise RuntimeError("TCNNEncoding: only 3 input dims are implemented")
        self.cfg, self.name_ = _gcfg(cfg), name
        offs = (C.c_int64 * (self.cfg[0] + 1))()
        total = capi.lib().gsdf_hashgrid_offsets(*self.cfg, offs)
        if total < 0:
            raise RuntimeError("TCNNEncoding: " + capi.lib().gsdf_last_error().decode())
        self.offsets = list(offs)
        g = None if seed is None else torch.Generator(device="cpu").manual_seed(seed)
        # tiny-cuda-nn initialises grid parameters U(-1e-4, 1e-4)
        init = (torch.rand(total * self.cfg[1], generator=g) * 2 - 1) * 1e-4
        self.params_ = init.to(device).requires_grad_(True)
        # optional: a pre-zeroed buffer shaped like params_ into which the table gradient is accumulated IN PLACE
        # (first order only) instead of being returned to autograd; set by LocalMap.flatten() for the trainer
        self.grad_sink = None
        # optional: a HIP stream on XCDs of its own (streams.xcd_partition_streams) for the scatter (table-gradient)
        # kernel.  With grad_sink set the scatter is launched there ASYNCHRONOUSLY (the input gradient is computed by a
        # separate launch on the current stream): the table gradient in grad_sink is complete only once that stream
        # has been waited for.
        self.scatter_stream = None
        # optional: the forward also stores d feat / d x (384 B per point) whenever x requires grad, and a first-order
        # backward (no create_graph) then gets d/dx from it instead of re-walking the table
        self.save_jacobian = False

    def get_out_dim(self):
        return self.cfg[0] * self.cfg[1]

    def forward(self, x, stencil=None):
        """`stencil` = (n, merge_levels) (not in the reference's interface): x holds n base rows followed by the 6 blocks of
        their central-difference points (LocalMap.query_points layout); lets the backward's binned scatter merge the rows of
        a group that share a grid cell."""
        if x.dim() != 2 or x.shape[1] != 3:
            raise RuntimeError("TCNNEncoding.forward: expected [B,3]")
        return _GridFwd.apply(x, self.params_.view(-1, self.cfg[1]), self.cfg, self.grad_sink, self.scatter_stream,
                              self.save_jacobian, stencil)

    __call__ = forward


def _mlp_fused_bwd(dims):
    """does gsdf_mlp_bwd with both gradients take the one-pass kernel (csrc/mlp_split.hip) for this topology?"""
    if os.environ.get("GSDF_MLP_FUSED_BWD", "1") == "0":
        return False
    dims_c = (C.c_int * len(dims))(*dims)
    return capi.lib().gsdf_mlp_bwd_is_one_pass(len(dims) - 1, dims_c) == 1


class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weights, biases, dims, sinks=None, aux_stream=None):
        L = capi.lib()
        x, weights = x.contiguous(), weights.contiguous()
        B, nl = x.shape[0], len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        out = torch.empty(B, dims[-1], dtype=torch.float32, device=x.device)
        need = any(ctx.needs_input_grad[:3])
        acts = torch.empty(L.gsdf_mlp_acts_floats(B, nl), dtype=torch.float32, device=x.device) if need else None
        capi.check(_timed("mlp_fwd", L.gsdf_mlp_fwd, B, nl, dims_c, f32(weights, "weights"), f32(biases), f32(x, "x"),
                          f32(out), f32(acts), capi.stream()), "mlp_fwd")
        ctx.save_for_backward(x, weights, biases, acts)
        ctx.dims, ctx.sinks, ctx.aux_stream = dims, sinks, aux_stream
        return out

    @staticmethod
    def backward(ctx, v_out):
        L = capi.lib()
        x, weights, biases, acts = ctx.saved_tensors
        dims = ctx.dims
        if torch.is_grad_enabled():
            # create_graph=True (the analytic eikonal term of the reference's default configuration differentiates
            # d sdf / d features again, local_map.cpp:151-172): the backward itself as a differentiable operator
            v_in, v_w, v_b = _MlpBwd.apply(v_out, x, weights, biases, acts, dims, bool(ctx.needs_input_grad[1]),
                                           biases is not None and bool(ctx.needs_input_grad[2]))
            return (v_in if ctx.needs_input_grad[0] else None), (v_w if ctx.needs_input_grad[1] else None), \
                   (v_b if (biases is not None and ctx.needs_input_grad[2]) else None), None, None, None
        v_out = v_out.detach()
        B, nl = x.shape[0], len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        v_out = v_out.contiguous()
        v_in = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if ctx.sinks is not None and ctx.needs_input_grad[1] and _sinks_live() and _mlp_fused_bwd(dims):
            # trainer fast path, fused: input and parameter gradients in one pass (v_pre stays in registers), the parameter
            # gradients accumulate straight into the flat gradient buffer
            w_sink, b_sink = ctx.sinks
            ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes_for(B, nl, dims_c, 1), dtype=torch.uint8, device=x.device)   # the waves' partial weight gradients
            capi.check(_timed("mlp_bwd", L.gsdf_mlp_bwd, B, nl, dims_c, f32(weights), f32(biases), f32(x), f32(acts),
                              f32(v_out), f32(v_in), f32(w_sink), f32(b_sink), ptr(ws), capi.stream()), "mlp_bwd")
            return v_in, None, None, None, None, None
        ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes(B, nl), dtype=torch.uint8, device=x.device)
        if ctx.sinks is not None and ctx.needs_input_grad[1] and _sinks_live():
            # trainer fast path: the parameter gradients ACCUMULATE straight into the flat gradient buffer (no zero-fill,
            # no autograd add), and their kernel runs on `aux_stream` so that d/d input — what the rest of the backward
            # pass waits for — is not queued behind it.  The gradients are complete once aux_stream has been waited for.
            w_sink, b_sink = ctx.sinks
            capi.check(_timed("mlp_bwd_data", L.gsdf_mlp_bwd, B, nl, dims_c, f32(weights), f32(biases), f32(x), f32(acts),
                              f32(v_out), f32(v_in), None, None, ptr(ws), capi.stream()), "mlp_bwd")

            def weights_half():
                capi.check(_timed("mlp_bwd_weights", L.gsdf_mlp_bwd_weights, B, nl, dims_c, int(biases is not None), f32(x),
                                  f32(acts), f32(v_out), ptr(ws), f32(w_sink), f32(b_sink), capi.stream()), "mlp_bwd_weights")
            aux = ctx.aux_stream
            if aux is None:
                weights_half()
            else:
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    weights_half()
                for t in (x, acts, v_out, ws):
                    t.record_stream(aux)
            return v_in, None, None, None, None, None
        v_w = torch.zeros_like(weights) if ctx.needs_input_grad[1] else None
        v_b = torch.zeros_like(biases) if (biases is not None and ctx.needs_input_grad[2]) else None
        capi.check(_timed("mlp_bwd", L.gsdf_mlp_bwd, B, nl, dims_c, f32(weights), f32(biases), f32(x), f32(acts),
                          f32(v_out), f32(v_in), f32(v_w), f32(v_b), ptr(ws), capi.stream()), "mlp_bwd")
        return v_in, v_w, v_b, None, None, None


class _MlpBwd(torch.autograd.Function):
    """(v_out, x, weights[, biases]) -> (v_in, v_weights, v_biases): the decoder's first-order backward as a differentiable
    operator.  Its own backward (include/gsdf_hip.h: gsdf_mlp_bwd_bwd) serves dL/d v_in: a masked bias-free forward pass for
    dL/d v_out and the weight-gradient GEMM on (vv_in, masked-forward activations) for dL/d weights; a ReLU network is piecewise
    linear, so nothing flows to x or the biases, and second derivatives through v_weights are not offered (the reference
    never asks for them)."""

    @staticmethod
    def forward(ctx, v_out, x, weights, biases, acts, dims, want_w, want_b):
        L = capi.lib()
        B, nl = x.shape[0], len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        v_out = v_out.contiguous()
        v_in = torch.empty_like(x)
        ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes(B, nl), dtype=torch.uint8, device=x.device)
        # the two-kernel form: the per-layer gradients stay in `ws` for the double backward
        capi.check(_timed("mlp_bwd_data", L.gsdf_mlp_bwd, B, nl, dims_c, f32(weights), f32(biases), f32(x), f32(acts), f32(v_out),
                          f32(v_in), None, None, ptr(ws), capi.stream()), "mlp_bwd")
        v_w = torch.zeros_like(weights) if want_w else torch.zeros(0, device=x.device)
        v_b = torch.zeros_like(biases) if want_b else torch.zeros(0, device=x.device)
        if want_w:
            capi.check(_timed("mlp_bwd_weights", L.gsdf_mlp_bwd_weights, B, nl, dims_c, int(want_b), f32(x), f32(acts), f32(v_out),
                              ptr(ws), f32(v_w), f32(v_b) if want_b else None, capi.stream()), "mlp_bwd_weights")
        ctx.save_for_backward(v_out, weights, acts, ws)
        ctx.dims = dims
        ctx.mark_non_differentiable(v_w, v_b)
        return v_in, v_w, v_b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, vv_in, _vv_w, _vv_b):
        L = capi.lib()
        v_out, weights, acts, ws = ctx.saved_tensors
        dims = ctx.dims
        B, nl = vv_in.shape[0], len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        g_vout = torch.empty_like(v_out)
        g_w = torch.zeros_like(weights) if ctx.needs_input_grad[2] else None
        ws2 = torch.empty(L.gsdf_mlp_bwd_bwd_ws_bytes(B, nl), dtype=torch.uint8, device=vv_in.device)
        capi.check(_timed("mlp_bwd_bwd", L.gsdf_mlp_bwd_bwd, B, nl, dims_c, f32(weights), f32(acts), f32(v_out), ptr(ws),
                          f32(vv_in.contiguous()), f32(g_vout), f32(g_w), ptr(ws2), capi.stream()), "mlp_bwd_bwd")
        return (g_vout if ctx.needs_input_grad[0] else None), None, g_w, None, None, None, None, None


class TCNNNetwork:
    """tcnn_binding's TCNNNetwork as the reference uses it (local_map.cpp:44-55,94): FullyFusedMLP, ReLU,
    output activation None, `n_neurons` 64, `n_hidden_layers` h -> h+1 bias-free linear layers.  fp32 weights
    (the reference's fork keeps fp16; DESIGN.md SPEC A.8).  `bias=True` gives the default torch decoder's topology."""

    def __init__(self, n_input_dims, n_output_dims, config, name="decoder", device="cuda", bias=False, seed=None):
        if config.get("otype", "FullyFusedMLP") != "FullyFusedMLP" or config.get("activation", "ReLU") != "ReLU" \
                or config.get("output_activation", "None") != "None":
            raise RuntimeError("TCNNNetwork: only FullyFusedMLP / ReLU / output_activation None is implemented")
        h, nh = int(config["n_neurons"]), int(config["n_hidden_layers"])
        self.dims = [int(n_input_dims)] + [h] * nh + [int(n_output_dims)]
        self.name_ = name
        g = None if seed is None else torch.Generator(device="cpu").manual_seed(seed)
        ws, bs = [], []
        for i, o in zip(self.dims[:-1], self.dims[1:]):        # Kaiming-uniform(a=sqrt(5)) like torch::nn::Linear
            bound = 1.0 / math.sqrt(i)
            ws.append((torch.rand(o * i, generator=g) * 2 - 1) * bound)
            bs.append((torch.rand(o, generator=g) * 2 - 1) * bound)
        self.params_ = torch.cat(ws).to(device).requires_grad_(True)
        self.biases_ = torch.cat(bs).to(device).requires_grad_(True) if bias else None
        # optional (set by LocalMap.flatten): pre-zeroed buffers shaped like params_ / biases_ into which the parameter
        # gradients are accumulated IN PLACE, and a stream for that half of the backward (see _MlpFn.backward)
        self.grad_sinks, self.aux_stream = None, None

    def forward(self, x):
        if x.dim() != 2 or x.shape[1] != self.dims[0]:
            raise RuntimeError(f"TCNNNetwork.forward: expected [B,{self.dims[0]}]")
        return _MlpFn.apply(x, self.params_, self.biases_, tuple(self.dims), self.grad_sinks, self.aux_stream)

    __call__ = forward


class _QueryPoints(torch.autograd.Function):
    """world points -> unit-cube encoder inputs (+ the 6 central-difference stencil points), one launch
    (include/gsdf_hip.h: gsdf_sdf_query_points).  The backward is a plain (twice differentiable) torch expression."""

    @staticmethod
    def forward(ctx, xyz, origin, map_size_inv, delta):
        xyz = xyz.contiguous()
        n, K = xyz.shape[0], (1 if delta is None else 7)
        out = torch.empty(K * n, 3, dtype=torch.float32, device=xyz.device)
        org = (C.c_float * 3)(*origin)
        capi.check(_timed("sdf_query_points", capi.lib().gsdf_sdf_query_points, n, int(K == 7), f32(xyz),
                          float(delta or 0.0), org, float(map_size_inv), f32(out), capi.stream()), "sdf_query_points")
        ctx.n, ctx.K, ctx.c = n, K, float(map_size_inv)
        return out

    @staticmethod
    def backward(ctx, v):
        g = v if ctx.K == 1 else v.view(ctx.K, ctx.n, 3).sum(0)
        return g * ctx.c, None, None, None                  # d out / d xyz = 0.5 * 2 * map_size_inv


class _SdfRayLoss(torch.autograd.Function):
    """loss::sdf_loss + w_eik * loss::eikonal_loss of the numerical gradient, value and gradient in one launch
    (include/gsdf_hip.h: gsdf_sdf_ray_loss)."""

    @staticmethod
    def forward(ctx, attr, gt_sdf, bce_isigma, delta, w_eik, n):
        attr, gt_sdf = attr.contiguous(), gt_sdf.contiguous()
        stencil = attr.shape[0] == 7 * n
        assert stencil or attr.shape[0] == n
        loss = torch.empty((), dtype=torch.float32, device=attr.device)
        v_attr = torch.empty_like(attr)
        capi.check(_timed("sdf_ray_loss", capi.lib().gsdf_sdf_ray_loss, n, int(stencil), f32(attr), attr.shape[1],
                          f32(gt_sdf), float(bce_isigma), float(delta or 0.0), float(w_eik), f32(loss), f32(v_attr),
                          capi.stream()), "sdf_ray_loss")
        ctx.save_for_backward(v_attr)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v_loss):
        return ctx.saved_tensors[0] * v_loss, None, None, None, None, None


class DepthSamples:
    """utils::DepthSamples (include/utils/utils.h): a batch of per-ray records; every field is [n, ...] or None."""
    FIELDS = ("origin", "direction", "depth", "xyz", "ray_sdf", "ridx")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw.get(f))

    def size(self, dim=0):
        return next(getattr(self, f) for f in self.FIELDS if getattr(self, f) is not None).shape[dim]

    def index_select(self, idx):
        return DepthSamples(**{f: (None if getattr(self, f) is None else getattr(self, f).index_select(0, idx)) for f in self.FIELDS})

    def cat(self, other):
        return DepthSamples(**{f: (None if getattr(self, f) is None or getattr(other, f) is None
                                   else torch.cat([getattr(self, f), getattr(other, f)], 0)) for f in self.FIELDS})


class _GsSdfLoss(torch.autograd.Function):
    """scale * loss::gs_sdf_loss(attr[:, 0:1], weights[ids]) : value and gradient in one launch (gsdf_gs_sdf_loss)."""

    @staticmethod
    def forward(ctx, attr, weights, ids, scale):
        attr, weights = attr.contiguous(), weights.contiguous().reshape(-1)
        loss = torch.empty((), dtype=torch.float32, device=attr.device)
        v_attr = torch.empty_like(attr)
        capi.check(_timed("gs_sdf_loss", capi.lib().gsdf_gs_sdf_loss, attr.shape[0], f32(attr), attr.shape[1], f32(weights),
                          ptr(None if ids is None else ids.contiguous(), torch.int64), float(scale), f32(loss), f32(v_attr),
                          capi.stream()), "gs_sdf_loss")
        ctx.save_for_backward(v_attr)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v_loss):
        return ctx.saved_tensors[0] * v_loss, None, None, None


class _CouplingLeg(torch.autograd.Function):
    """The GS<->SDF coupling of the joint iteration as ONE autograd node (trainer fast path; same kernels, same results
    as the composed operators): forward = row gather, query points (+ the 6 central-difference points when `delta` is
    given), encoder (+ Jacobian of the base rows), decoder, loss = scale * gs_sdf_loss [+ w_eik * eikonal_loss of the
    numerical gradient: NeuralSLAM::sdf_regularization(gs_samples.detach(), ...), neural_mapping.cpp:448-451]; backward =
    decoder data / weight gradients, d/dx of the base rows from the Jacobian, table scatter, row scatter.  Saves the
    host time of ~10 autograd nodes per step.  Requires the in-place gradient sinks of
    LocalMap.flatten(accumulate_table_grad_in_place=True) and grad_sinks_armed() around the backward call."""

    @staticmethod
    def forward(ctx, samples, ids, weights, lm, scale, delta, w_eik):
        L = capi.lib()
        enc, dec = lm.encoder, lm.decoder
        cfg, dims = enc.cfg, tuple(dec.dims)
        xs = samples.index_select(0, ids)
        n = xs.shape[0]
        K = 1 if delta is None else 7
        dev = xs.device
        x01 = torch.empty(K * n, 3, dtype=torch.float32, device=dev)
        capi.check(L.gsdf_sdf_query_points(n, int(K == 7), f32(xs), float(delta or 0.0), (C.c_float * 3)(*lm._origin),
                                           float(lm.map_size_inv), f32(x01), capi.stream()), "sdf_query_points")
        table = enc.params_.view(-1, cfg[1])
        nf = cfg[0] * cfg[1]
        feat = torch.empty(K * n, nf, dtype=torch.float32, device=dev)
        jac = torch.empty(n, nf, 3, dtype=torch.float32, device=dev)
        if _stencil_fwd((n, 0), K * n):
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd_stencil, K * n, n, n, *cfg, f32(x01), f32(table), f32(feat), f32(jac),
                              capi.stream()), "hashgrid_fwd_stencil")
        else:
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd_jac_rows, K * n, n, *cfg, f32(x01), f32(table), f32(feat), f32(jac),
                              capi.stream()), "hashgrid_fwd_jac")
        nl = len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        attr = torch.empty(K * n, dims[-1], dtype=torch.float32, device=dev)
        acts = torch.empty(L.gsdf_mlp_acts_floats(K * n, nl), dtype=torch.float32, device=dev)
        capi.check(_timed("mlp_fwd", L.gsdf_mlp_fwd, K * n, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(feat), f32(attr),
                          f32(acts), capi.stream()), "mlp_fwd")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        v_attr = torch.empty_like(attr)
        capi.check(_timed("gs_sdf_loss", L.gsdf_gs_sdf_eik_loss, n, int(K == 7), f32(attr), attr.shape[1], f32(weights.reshape(-1)),
                          ptr(ids, torch.int64), float(scale), float(delta or 0.0), float(w_eik), f32(loss), f32(v_attr),
                          capi.stream()), "gs_sdf_loss")
        ctx.save_for_backward(ids, x01, feat, jac, acts, v_attr)
        ctx.lm, ctx.n_rows, ctx.delta = lm, samples.shape[0], float(delta or 0.0)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v_loss):
        if _SinkState.armed <= 0:
            raise RuntimeError("gs_sdf_coupling: call backward inside `with sdf.grad_sinks_armed():` (the node accumulates "
                               "parameter gradients in place)")
        L = capi.lib()
        ids, x01, feat, jac, acts, v_attr = ctx.saved_tensors
        lm = ctx.lm
        enc, dec = lm.encoder, lm.decoder
        cfg, dims = enc.cfg, tuple(dec.dims)
        nq, n, nl = x01.shape[0], jac.shape[0], len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        v_out = v_attr * v_loss
        v_feat = torch.empty_like(feat)
        w_sink, b_sink = dec.grad_sinks
        cur = torch.cuda.current_stream()
        if _mlp_fused_bwd(dims):
            ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes_for(nq, nl, dims_c, 1), dtype=torch.uint8, device=x01.device)
            capi.check(_timed("mlp_bwd", L.gsdf_mlp_bwd, nq, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(feat), f32(acts),
                              f32(v_out), f32(v_feat), f32(w_sink), f32(b_sink), ptr(ws), capi.stream()), "mlp_bwd")
        else:
            ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes(nq, nl), dtype=torch.uint8, device=x01.device)
            capi.check(_timed("mlp_bwd_data", L.gsdf_mlp_bwd, nq, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(feat), f32(acts),
                              f32(v_out), f32(v_feat), None, None, ptr(ws), capi.stream()), "mlp_bwd")

            def weights_half():
                capi.check(_timed("mlp_bwd_weights", L.gsdf_mlp_bwd_weights, nq, nl, dims_c, int(dec.biases_ is not None), f32(feat),
                                  f32(acts), f32(v_out), ptr(ws), f32(w_sink), f32(b_sink), capi.stream()), "mlp_bwd_weights")
            if dec.aux_stream is None:
                weights_half()
            else:
                dec.aux_stream.wait_stream(cur)
                with torch.cuda.stream(dec.aux_stream):
                    weights_half()
                for t in (feat, acts, v_out, ws):
                    t.record_stream(dec.aux_stream)
        # d/dx of the base rows only: the stencil rows are evaluated at gs_samples.detach() (neural_mapping.cpp:449)
        v_x = torch.empty(n, 3, dtype=torch.float32, device=x01.device)
        capi.check(_timed("hashgrid_bwd_input", L.gsdf_hashgrid_bwd_jac, n, cfg[0], cfg[1], f32(jac), f32(v_feat), f32(v_x),
                          capi.stream()), "hashgrid_bwd_jac")
        table = enc.params_.view(-1, cfg[1])
        sink = enc.grad_sink.view(table.shape)

        stencil = None if nq == n else (n, stencil_merge_levels(cfg, ctx.delta * lm.map_size_inv))

        def scatter():
            scatter_table_grad(nq, cfg, x01, table, v_feat, sink, None, stencil)
        if enc.scatter_stream is None:
            scatter()
        else:
            enc.scatter_stream.wait_stream(cur)
            with torch.cuda.stream(enc.scatter_stream):
                scatter()
            v_feat.record_stream(enc.scatter_stream)
            x01.record_stream(enc.scatter_stream)
        v_samples = torch.zeros(ctx.n_rows, 3, dtype=torch.float32, device=x01.device)
        v_samples.index_add_(0, ids, v_x * float(lm.map_size_inv))          # d x01 / d xyz = 0.5 * 2 * map_size_inv
        return v_samples, None, None, None, None, None, None


class _SdfBatchAnalytic(torch.autograd.Function):
    """The SDF work of one joint iteration in the reference's DEFAULT configuration (decoder_implementation 0 / numerical_grad 0,
    config/base.yaml:12-13) as ONE autograd node on the fused kernels, first and second order.  One batch holds both point sets:
        rows [0, n_ray)  per-ray batch:   w_sdf * loss::sdf_loss(get_sdf(xyz), gt)                     (neural_mapping.cpp:165-170)
        rows [n_ray, n)  splat samples:   w_gs * loss::gs_sdf_loss(get_sdf(samples[ids]), w[ids])      (:436-457; d/d samples returned)
        each set:  + w_eik * loss::eikonal_loss(g),  g = autograd.grad(sdf, xyz, create_graph=True)     (local_map.cpp:151-172; the splat
                                                                                                         samples detached, :448-451)
                   + w_align * mean |g - get_gradient(xyz, delta, numerical).detach()|                 (neural_mapping.cpp:126-134)
    forward : query points (+ 6 stencil rows for the align term) -> encoder (+ Jacobian of the base rows) -> decoder (activations
              saved for the base rows only; the stencil rows are forward-only) -> decoder backward of e_0 (g0 = d sdf / d features,
              its per-layer gradients kept) -> ONE loss launch (value, d/d decoder output, dL/d(J^T g0), dL/d g0)
    backward: one-pass decoder backward of the data terms; decoder DOUBLE backward (gsdf_mlp_bwd_bwd); ONE binned scatter that
              carries the first-order and the second-order table gradient of every corner (gsdf_hashgrid_bwd_binned2); d/dx of
              the splat samples' data term from the Jacobian.
    Parameter gradients accumulate in place (LocalMap.flatten(accumulate_table_grad_in_place=True) + grad_sinks_armed())."""

    @staticmethod
    def forward(ctx, ray_xyz, gt_sdf, samples, ids, weights, lm, w_sdf, w_gs, delta, w_eik, w_align, _anchor):
        # _anchor = the encoder's parameter tensor: makes the node part of the graph when no input carries a gradient (ray
        # batch alone); its gradient is deposited in the sink, autograd gets None
        L = capi.lib()
        enc, dec = lm.encoder, lm.decoder
        cfg, dims = enc.cfg, tuple(dec.dims)
        parts = []
        if ray_xyz is not None and ray_xyz.shape[0] > 0:
            parts.append(ray_xyz.detach().reshape(-1, 3))
        n_ray = parts[0].shape[0] if parts else 0
        if samples is not None:
            parts.append(samples.detach().reshape(-1, 3) if ids is None else samples.detach().index_select(0, ids))
        xs = (parts[0] if len(parts) == 1 else torch.cat(parts, 0)).contiguous()
        n, dev = xs.shape[0], xs.device
        stencil = bool(w_align != 0.0) and n > 0
        K = 7 if stencil else 1
        x01 = torch.empty(K * n, 3, dtype=torch.float32, device=dev)
        table = enc.params_.view(-1, cfg[1])
        nf, nl = cfg[0] * cfg[1], len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        feat = torch.empty(K * n, nf, dtype=torch.float32, device=dev)
        jac = torch.empty(n, nf, 3, dtype=torch.float32, device=dev)
        org = (C.c_float * 3)(*lm._origin)
        if stencil and delta and _stencil_fwd((n, 0), K * n):
            # the query points are made inside the encoder's launch (gsdf_hashgrid_fwd_stencil_points: the same rows, features and Jacobians, bit for bit)
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd_stencil_points, n, f32(xs), 0, None, None, float(delta), org, float(lm.map_size_inv), 1,
                              *cfg, f32(table), f32(x01), f32(feat), f32(jac), capi.stream()), "hashgrid_fwd_stencil_points")
        else:
            capi.check(L.gsdf_sdf_query_points(n, int(stencil), f32(xs), float(delta or 0.0), org, float(lm.map_size_inv), f32(x01), capi.stream()),
                       "sdf_query_points")
            capi.check(_timed("hashgrid_fwd", L.gsdf_hashgrid_fwd_jac_rows, K * n, n, *cfg, f32(x01), f32(table), f32(feat), f32(jac),
                              capi.stream()), "hashgrid_fwd_jac")
        attr = torch.empty(K * n, dims[-1], dtype=torch.float32, device=dev)
        acts = torch.empty(L.gsdf_mlp_acts_floats(n, nl), dtype=torch.float32, device=dev)
        fb, ab = feat[:n], attr[:n]
        capi.check(_timed("mlp_fwd", L.gsdf_mlp_fwd, n, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(fb), f32(ab), f32(acts),
                          capi.stream()), "mlp_fwd")
        if stencil:   # forward-only rows: the numerical gradient of the align term is detached
            capi.check(_timed("mlp_fwd", L.gsdf_mlp_fwd, 6 * n, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(feat[n:]), f32(attr[n:]),
                              None, capi.stream()), "mlp_fwd")
        # g0 = d sdf / d features: the decoder's backward of e_0, its per-layer gradients stay in `bws` for the double backward
        e0 = torch.zeros(n, dims[-1], dtype=torch.float32, device=dev)
        e0[:, 0] = 1.0
        g0 = torch.empty(n, nf, dtype=torch.float32, device=dev)
        # topologies of the one-pass backward: the lean e0 backward (the chain alone on the bf16 pipe, nothing saved); the double backward then
        # recomputes it from the ReLU masks (gsdf_mlp_bwd_bwd with bwd_ws = NULL); other topologies: the fp32-pipe pair + v_pre images
        lean = L.gsdf_mlp_bwd_is_one_pass(nl, dims_c) == 1
        bws = None if lean else torch.empty(L.gsdf_mlp_bwd_ws_bytes(n, nl), dtype=torch.uint8, device=dev)
        capi.check(_timed("mlp_bwd_data", L.gsdf_mlp_bwd, n, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(fb), f32(acts), f32(e0),
                          f32(g0), None, None, ptr(bws), capi.stream()), "mlp_bwd")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        v_attr = torch.empty(n, dims[-1], dtype=torch.float32, device=dev)
        vv_x = torch.empty(n, 3, dtype=torch.float32, device=dev)
        u0 = torch.empty(n, nf, dtype=torch.float32, device=dev)
        gt_c = None if n_ray == 0 else gt_sdf.contiguous().reshape(-1)
        w_c = None if n == n_ray else weights.contiguous().reshape(-1)
        capi.check(_timed("sdf_analytic_loss", L.gsdf_sdf_analytic_loss, n, n_ray, int(stencil), f32(attr), attr.shape[1], f32(g0), nf,
                          f32(jac), f32(gt_c), f32(w_c), ptr(ids, torch.int64) if (n > n_ray and ids is not None) else None,
                          float(lm.bce_isigma), float(w_sdf), float(w_gs), float(lm.map_size_inv), float(delta or 0.0), float(w_eik),
                          float(w_align), f32(loss), f32(v_attr), f32(vv_x), f32(u0), capi.stream()), "sdf_analytic_loss")
        ctx.save_for_backward(ids, x01, feat, jac, acts, bws, e0, g0, v_attr, vv_x, u0)
        ctx.lm, ctx.n, ctx.n_ray = lm, n, n_ray
        ctx.n_rows = 0 if samples is None else samples.shape[0]
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v_loss):
        if _SinkState.armed <= 0:
            raise RuntimeError("analytic SDF batch: call backward inside `with sdf.grad_sinks_armed():` (the node accumulates "
                               "parameter gradients in place)")
        L = capi.lib()
        ids, x01, feat, jac, acts, bws, e0, g0, v_attr, vv_x, u0 = ctx.saved_tensors
        lm, n, n_ray = ctx.lm, ctx.n, ctx.n_ray
        enc, dec = lm.encoder, lm.decoder
        cfg, dims = enc.cfg, tuple(dec.dims)
        nl = len(dims) - 1
        dims_c = (C.c_int * len(dims))(*dims)
        dev = x01.device
        want_x = ctx.needs_input_grad[2] and n > n_ray
        none = (None,) * 9
        if n == 0:
            return (None, None, (torch.zeros(ctx.n_rows, 3, device=dev) if ctx.needs_input_grad[2] else None)) + none
        fb, xb = feat[:n], x01[:n]
        v_out = (v_attr * v_loss).contiguous()
        v_feat = torch.empty(n, cfg[0] * cfg[1], dtype=torch.float32, device=dev)
        w_sink, b_sink = dec.grad_sinks
        # first order: data terms through the decoder (one pass: input + parameter gradients)
        ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes_for(n, nl, dims_c, 1), dtype=torch.uint8, device=dev)
        capi.check(_timed("mlp_bwd", L.gsdf_mlp_bwd, n, nl, dims_c, f32(dec.params_), f32(dec.biases_), f32(fb), f32(acts), f32(v_out),
                          f32(v_feat), f32(w_sink), f32(b_sink), ptr(ws) if ws.numel() else None, capi.stream()), "mlp_bwd")
        v_samples = None
        if want_x:   # d (data term) / d samples from the Jacobian of the splat rows (the regularisers see samples.detach())
            ng = n - n_ray
            v_x = torch.empty(ng, 3, dtype=torch.float32, device=dev)
            capi.check(_timed("hashgrid_bwd_input", L.gsdf_hashgrid_bwd_jac, ng, cfg[0], cfg[1], f32(jac[n_ray:]), f32(v_feat[n_ray:]), f32(v_x),
                              capi.stream()), "hashgrid_bwd_jac")
            v_samples = torch.zeros(ctx.n_rows, 3, dtype=torch.float32, device=dev)
            if ids is None:
                v_samples.copy_(v_x * float(lm.map_size_inv))
            else:
                v_samples.index_add_(0, ids, v_x * float(lm.map_size_inv))
        # second order: the regularisers reach the decoder weights through g0 = W_0^T D_0 ... e_0 (only the optimizer waits for it:
        # issued after the samples' gradient)
        vv_in = (u0 * v_loss).contiguous()
        g_vout = torch.empty_like(e0)
        ws2 = torch.empty(L.gsdf_mlp_bwd_bwd_ws_bytes(n, nl), dtype=torch.uint8, device=dev)
        capi.check(_timed("mlp_bwd_bwd", L.gsdf_mlp_bwd_bwd, n, nl, dims_c, f32(dec.params_), f32(acts), f32(e0), ptr(bws), f32(vv_in),
                          f32(g_vout), f32(w_sink), ptr(ws2), capi.stream()), "mlp_bwd_bwd")
        # table: first-order (v_feat) and second-order (g0, vv_x) contributions of every corner in ONE scatter
        table = enc.params_.view(-1, cfg[1])
        sink = enc.grad_sink.view(table.shape)
        vvx = (vv_x * v_loss).contiguous()

        def scatter():
            mode = os.environ.get("GSDF_HASHGRID_BINNED", "auto")
            nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(n, *cfg) if (mode != "0" and (mode == "1" or n >= BINNED_MIN_POINTS)) else 0
            if nbytes:
                bws2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                capi.check(_timed("hashgrid_bwd", L.gsdf_hashgrid_bwd_binned2, n, *cfg, f32(xb), f32(v_feat), f32(g0), f32(vvx), f32(sink),
                                  ptr(bws2), nbytes, capi.stream()), "hashgrid_bwd_binned2")
            else:
                capi.check(_timed("hashgrid_bwd", L.gsdf_hashgrid_bwd, n, *cfg, f32(xb), f32(table), f32(v_feat), f32(sink), None,
                                  capi.stream()), "hashgrid_bwd")
                capi.check(_timed("hashgrid_bwd_bwd", L.gsdf_hashgrid_bwd_bwd, n, *cfg, f32(xb), f32(table), f32(g0), f32(vvx), None,
                                  f32(sink), None, capi.stream()), "hashgrid_bwd_bwd")
        if enc.scatter_stream is None:
            scatter()
        else:
            cur = torch.cuda.current_stream()
            enc.scatter_stream.wait_stream(cur)
            with torch.cuda.stream(enc.scatter_stream):
                scatter()
            for t in (v_feat, x01, g0, vvx):
                t.record_stream(enc.scatter_stream)
        return (None, None, v_samples) + none


class LocalMap:
    """The SDF half of the reference's `LocalMap` (include/neural_net/local_map.{h,cpp}): hash-grid encoder +
    decoder, `get_sdf`, `get_gradient` (numerical 6-point stencil or analytic via autograd)."""

    def __init__(self, map_origin, map_size, bce_sigma=0.02, decoder_implementation=0, hidden_dim=64, geo_num_layer=3,
                 device="cuda", seed=0, encoding_config=None, decoder_backend="fused"):
        """decoder_implementation as config/base.yaml:12: 0 = the torch::nn::Sequential topology (biases, geo_num_layer + 1 hidden
        matmuls; the reference's DEFAULT), 1 = tcnn FullyFusedMLP (bias free, geo_num_layer hidden matmuls).  Both run on the
        fused MFMA kernels, first AND second order; decoder_backend="torch" builds implementation 0 as an eager
        torch.nn.Sequential instead (rocBLAS GEMMs: the A/B reference of the tests)."""
        self.pos_W_M = torch.as_tensor(map_origin, dtype=torch.float32, device=device).reshape(1, 3)
        self.map_size_inv = 1.0 / float(map_size)
        self._origin = [float(v) for v in map_origin]
        self.bce_isigma = 1.0 / float(bce_sigma)
        self.encoder = TCNNEncoding(3, encoding_config, "encoder_local_map", device, seed)
        feat = self.encoder.get_out_dim()
        self.decoder_implementation = decoder_implementation
        if decoder_implementation == 0 and decoder_backend == "torch":      # torch::nn::Sequential, local_map.cpp:29-42 (rocBLAS GEMMs)
            torch.manual_seed(seed)
            layers = [torch.nn.Linear(feat, hidden_dim), torch.nn.ReLU(True)]
            for _ in range(geo_num_layer):
                layers += [torch.nn.Linear(hidden_dim, hidden_dim), torch.nn.ReLU(True)]
            layers += [torch.nn.Linear(hidden_dim, 2)]
            self.decoder = torch.nn.Sequential(*layers).to(device)
        elif decoder_implementation == 1:    # tcnn FullyFusedMLP, local_map.cpp:44-55
            self.decoder = TCNNNetwork(feat, 2, dict(otype="FullyFusedMLP", activation="ReLU", output_activation="None",
                                                     n_neurons=hidden_dim, n_hidden_layers=geo_num_layer), "decoder", device, seed=seed + 1)
        else:                                # 0 (or 2): fused MFMA kernels with the torch decoder's topology (biases, 4 hidden matmuls)
            self.decoder = TCNNNetwork(feat, 2, dict(n_neurons=hidden_dim, n_hidden_layers=geo_num_layer + 1), "decoder", device,
                                       bias=True, seed=seed + 1)

    def parameters(self):
        ps = [self.encoder.params_]
        if isinstance(self.decoder, torch.nn.Module):
            ps += list(self.decoder.parameters())
        else:
            ps += [self.decoder.params_] + ([self.decoder.biases_] if self.decoder.biases_ is not None else [])
        return ps

    def flatten(self, accumulate_table_grad_in_place=False):
        """Moves every parameter into one flat buffer (single all-reduce message); returns a FlatGroup."""
        from .trainer import FlatGroup, flatten_leaves
        flat, flat_grad, views = flatten_leaves(self.parameters())
        if isinstance(self.decoder, torch.nn.Module):
            self.encoder.params_ = views[0]
            for p, v in zip(self.decoder.parameters(), views[1:]):
                p.data = v.data
                p.grad = v.grad
        else:
            self.encoder.params_, self.decoder.params_ = views[0], views[1]
            if self.decoder.biases_ is not None:
                self.decoder.biases_ = views[2]
        if accumulate_table_grad_in_place:
            self.encoder.grad_sink = self.encoder.params_.grad
            if not isinstance(self.decoder, torch.nn.Module):
                self.decoder.grad_sinks = (self.decoder.params_.grad,
                                           None if self.decoder.biases_ is None else self.decoder.biases_.grad)
        return FlatGroup(flat, flat_grad)

    # ---- occupancy structure: SubMap (sub_map.cpp:7-80) + LocalMap::sample / filter_sample (local_map.cpp:449-516) ------
    def set_bounds(self, inner_map_size, leaf_size):
        """The k_* globals of params.cpp:243-256: cube of side inner_map_size around the origin, octree level
        ceil(log2((inner + 2 leaf) / leaf)), map_size = 2^level * leaf (must equal this map's map_size)."""
        self.leaf_size = float(leaf_size)
        self.octree_level = int(math.ceil(math.log2((inner_map_size + 2 * leaf_size) / leaf_size)))
        want = (2 ** self.octree_level) * leaf_size
        if abs(want * self.map_size_inv - 1.0) > 1e-6:
            raise RuntimeError(f"LocalMap.set_bounds: map_size must be 2^level * leaf_size = {want}")
        half = torch.full((1, 3), 0.5 * float(inner_map_size), device=self.pos_W_M.device)
        self.xyz_min_W, self.xyz_max_W = self.pos_W_M - half, self.pos_W_M + half
        self.acc_struct_occ = None

    def xyz_to_m1p1_pts(self, xyz):                # sub_map.cpp:82-93
        return (xyz - self.pos_W_M) * 2 * self.map_size_inv

    def m1p1_pts_to_xyz(self, pts):
        return pts * 0.5 * (1.0 / self.map_size_inv) + self.pos_W_M

    def scale_from_m1p1(self, t):
        return t * 0.5 * (1.0 / self.map_size_inv)

    def update_octree_as(self, xyz, is_prior=False):
        """quantize -> unique -> (27-neighbour dilation unless is_prior) -> occupancy structure (sub_map.cpp:22-35)."""
        from .occupancy import OctreeAS
        self.acc_struct_occ = OctreeAS.from_points(self.xyz_to_m1p1_pts(xyz), self.octree_level, dilate27=not is_prior)

    def get_valid_mask(self, xyz, level=-1):       # sub_map.cpp:76-80 (normalisation + query + "> -1" in one launch)
        return self.acc_struct_occ.query_world_mask(xyz, self._origin, self.map_size_inv, level)

    def get_inrange_mask(self, xyz, padding=0.0):  # sub_map.cpp:37-45
        return ((xyz < self.xyz_max_W - padding - 1e-6) & (xyz > self.xyz_min_W + padding + 1e-6)).all(1)

    def get_intersect_point(self, points, rays, padding=0.0):   # sub_map.cpp:47-74 -> (z_nears, z_fars, mask_intersect)
        tmp = torch.where(rays == 0, rays + 1e-6, rays)
        a, b = (self.xyz_min_W + padding - points) / tmp, (self.xyz_max_W - padding - points) / tmp
        z_nears, z_fars = torch.minimum(a, b).max(1).values, torch.maximum(a, b).min(1).values
        return z_nears, z_fars, z_nears < z_fars

    def sample(self, rays, voxel_sample_num, sample_free, free_sample_num=3, generator=None):
        """LocalMap::sample (local_map.cpp:449-509): `voxel_sample_num` points in every occupied voxel each ray crosses
        (SDF target = ray depth - sample depth), optionally the stratified free-space samples, keeping what lies in
        front of the surface.  `rays` / result: DepthSamples."""
        from .neural_gs import sample_free_pts
        if voxel_sample_num < 1:
            return DepthSamples(xyz=rays.xyz, ray_sdf=torch.zeros_like(rays.depth), direction=rays.direction, ridx=rays.ridx)
        rm = self.acc_struct_occ.raymarch(self.xyz_to_m1p1_pts(rays.origin).contiguous(), rays.direction.contiguous(),
                                          "voxel", voxel_sample_num)
        out = rays.index_select(rm.ridx)
        out.ridx = rm.ridx
        out.xyz = self.m1p1_pts_to_xyz(rm.samples)
        d = self.scale_from_m1p1(rm.depth_samples)
        out.ray_sdf, out.depth = out.depth - d, d
        if sample_free:
            fx, fs, fr = sample_free_pts(rays.origin, rays.direction, rays.depth, free_sample_num, generator)
            free = rays.index_select(fr)
            free.xyz, free.ray_sdf, free.ridx, free.depth = fx, fs, rays.ridx.index_select(0, fr), free.depth - fs
            out = out.cat(free)
        return out.index_select((out.ray_sdf > 0).reshape(-1).nonzero().reshape(-1))

    def export_as_occ_prior(self, path):
        """as_occ_prior.ply (neural_mapping.cpp:755-762): the occupied voxels' minimum corners in world coordinates."""
        from .occupancy import spc_ops, write_points_ply
        vox = self.acc_struct_occ.get_quantized_points()
        write_points_ply(path, self.m1p1_pts_to_xyz(spc_ops.quantized_points_to_fpoints(vox, self.octree_level)))

    def load_as_occ_prior(self, path):
        """checkpoint load (neural_mapping.cpp:1367-1373): update_octree_as(xyz, is_prior=true)."""
        from .occupancy import read_points_ply
        self.update_octree_as(read_points_ply(path, self.pos_W_M.device)["xyz"], is_prior=True)

    def export_checkpoint(self, path):
        """local_map_checkpoint.pt = torch::save(local_map_ptr, path) (neural_mapping.cpp:1331-1342); gs_sdf_amd/checkpoint.py."""
        from .checkpoint import save_local_map_checkpoint
        save_local_map_checkpoint(self, path)

    def load_checkpoint(self, path):
        """torch::load(local_map_ptr, path) (neural_mapping.cpp:1344-1351)."""
        from .checkpoint import load_local_map_checkpoint
        return load_local_map_checkpoint(self, path)

    def filter_sample(self, samples):              # local_map.cpp:511-516
        keep = (self.acc_struct_occ.query(self.xyz_to_m1p1_pts(samples.xyz)).pidx > -1).nonzero().reshape(-1)
        return samples.index_select(keep)

    def xyz_to_zp1_pts(self, xyz):                 # sub_map.cpp:82-97
        return 0.5 * ((xyz - self.pos_W_M) * (2.0 * self.map_size_inv)) + 0.5

    def query_points(self, xyz, delta=None):
        """xyz_to_zp1_pts (bit-identical) in one launch; with `delta` also the 6 stencil points of get_gradient:
        rows [0,n) base, then (+x,-x,+y,-y,+z,-z) x n."""
        return _QueryPoints.apply(xyz, self._origin, self.map_size_inv, delta)

    def get_feat(self, xyz, normalized=False):
        return self.encoder.forward(xyz if normalized else self.query_points(xyz))

    def gs_sdf_loss(self, xyz, weights, ids=None, scale=1.0):
        """scale * loss::gs_sdf_loss(get_sdf(xyz)[0], weights[ids]) (neural_mapping.cpp:436-462) with the decoder output
        consumed by ONE loss launch (no slicing / square / mul / sum kernels, forward or backward)."""
        attr = self.decoder(self.encoder.forward(self.query_points(xyz)))
        return _GsSdfLoss.apply(attr, weights, ids, scale)

    def gs_sdf_coupling(self, samples, ids, weights, scale=1.0, delta=None, w_eik=0.0):
        """= gs_sdf_loss(samples[ids], weights, ids, scale) [+ w_eik * eikonal_loss(get_gradient(samples[ids].detach(), delta,
        numerical)) when `delta` is given: the GS-sample regulariser of the joint iteration, neural_mapping.cpp:448-451], as
        a single autograd node (see _CouplingLeg)."""
        if self.encoder.grad_sink is None or getattr(self.decoder, "grad_sinks", None) is None:
            raise RuntimeError("gs_sdf_coupling needs LocalMap.flatten(accumulate_table_grad_in_place=True) with the fused decoder")
        return _CouplingLeg.apply(samples, ids, weights, self, scale, delta, w_eik)

    def _analytic_ready(self):
        if self.encoder.grad_sink is None or getattr(self.decoder, "grad_sinks", None) is None:
            raise RuntimeError("the fused analytic SDF batch needs LocalMap.flatten(accumulate_table_grad_in_place=True) with a fused decoder")

    def ray_loss_analytic(self, xyz, gt_sdf, delta, w_sdf=1.0, w_eik=0.1, w_align=0.1):
        """The per-ray batch of the reference's DEFAULT configuration (neural_mapping.cpp:138-188 with numerical_grad 0):
        w_sdf * sdf_loss(get_sdf(xyz)) + w_eik * eikonal_loss(analytic gradient) + w_align * |analytic - numerical.detach()|.mean(),
        one autograd node (_SdfBatchAnalytic)."""
        return self.joint_sdf_loss_analytic(xyz, gt_sdf, None, None, None, delta, w_sdf, 0.0, w_eik, w_align)

    def gs_sdf_coupling_analytic(self, samples, ids, weights, scale=1.0, delta=None, w_eik=0.0, w_align=0.0):
        """The GS<->SDF block of the DEFAULT configuration (neural_mapping.cpp:420-462): scale * gs_sdf_loss(get_sdf(samples[ids]),
        weights[ids]) + sdf_regularization(samples[ids].detach()) with the ANALYTIC gradient (eikonal + align), one autograd node."""
        return self.joint_sdf_loss_analytic(None, None, samples, ids, weights, delta, 0.0, scale, w_eik, w_align)

    def joint_sdf_loss_analytic(self, ray_xyz, gt_sdf, samples, ids, weights, delta, w_sdf=1.0, w_gs=1e-3, w_eik=0.1, w_align=0.1):
        """ray_loss_analytic(ray_xyz, gt_sdf) + gs_sdf_coupling_analytic(samples, ids, weights) as ONE batch through the encoder /
        decoder / scatter (the SDF work of a whole joint iteration in ~12 launches); either part may be None."""
        self._analytic_ready()
        if ray_xyz is None and samples is None:
            raise RuntimeError("joint_sdf_loss_analytic: no points")
        return _SdfBatchAnalytic.apply(ray_xyz, gt_sdf, samples, ids, weights, self, w_sdf, w_gs, delta, w_eik, w_align, self.encoder.params_)

    def ray_loss(self, xyz, gt_sdf, delta, w_eik):
        """sdf_loss(get_sdf(xyz)) + w_eik * eikonal_loss(get_gradient(xyz, delta, numerical)) of the per-ray batch
        (neural_mapping.cpp:138-188) as ONE encoder launch, ONE decoder launch and ONE loss launch over the 7n points."""
        stencil = (xyz.shape[0], stencil_merge_levels(self.encoder.cfg, delta * self.map_size_inv))
        attr = self.decoder(self.encoder.forward(self.query_points(xyz, delta), stencil))
        return _SdfRayLoss.apply(attr, gt_sdf, self.bce_isigma, delta, w_eik, xyz.shape[0])

    def get_sdf(self, xyz, with_isigma=True):
        """-> [sdf [B,1], isigma [B,1]]  (local_map.cpp:87-103); with_isigma=False (not in the reference) returns [sdf]
        only and skips the three elementwise launches of the isigma head for callers that drop it."""
        attr = self.decoder(self.get_feat(xyz))
        if not with_isigma:
            return [attr[:, 0:1]]
        sdf, raw = attr[:, 0:1], attr[:, 1:2]
        return [sdf, 1 + torch.nn.functional.softplus(raw, beta=100) * self.bce_isigma]

    def get_gradient(self, xyz, delta, sdf=None, hessian=False, numerical_grad=True):
        """(local_map.cpp:105-173) numerical: central differences (+ diagonal Hessian); analytic: autograd."""
        if numerical_grad:
            offs = torch.tensor([[delta, 0, 0], [-delta, 0, 0], [0, delta, 0], [0, -delta, 0], [0, 0, delta], [0, 0, -delta]],
                                dtype=torch.float32, device=xyz.device)[:, None, :]
            pts = xyz[None] + offs
            ps = self.get_sdf(pts.reshape(-1, 3))[0].view(6, xyz.shape[0], 1)
            inv = 1.0 / delta
            grad = 0.5 * inv * torch.cat([ps[0] - ps[1], ps[2] - ps[3], ps[4] - ps[5]], 1)
            if hessian:
                if sdf is None:
                    sdf = self.get_sdf(xyz)[0]
                hess = inv * inv * (torch.cat([ps[0] + ps[1], ps[2] + ps[3], ps[4] + ps[5]], 1) - 2 * sdf)
                return [grad, hess]
            return [grad]
        with torch.enable_grad():
            if not xyz.requires_grad or sdf is None:
                xyz.requires_grad_(True)
                sdf = self.get_sdf(xyz)[0]
            grad = torch.autograd.grad([sdf], [xyz], [torch.ones_like(sdf)], retain_graph=True, create_graph=True)[0]
            if hessian:
                hess = torch.autograd.grad([grad], [xyz], [torch.ones_like(grad)], retain_graph=True, create_graph=True)[0]
                return [grad, hess]
        return [grad]


# ---- losses (include/optimizer/loss/loss.cpp) -------------------------------------------------------------------
def sdf_loss(pred_sdf, gt_sdf, pred_isigma):                  # loss.cpp:49-79
    isigma = pred_isigma.clamp_max(5e2)
    return torch.nn.functional.binary_cross_entropy_with_logits(-pred_sdf * isigma,
                                                                torch.sigmoid(-gt_sdf * isigma).clamp(1e-7, 1 - 1e-7))


def eikonal_loss(grad):                                        # loss.cpp:81-83
    return (grad.norm(2, 1) - 1.0).square().mean()


def curvate_loss(hessian):                                     # loss.cpp:85-90
    return hessian.sum(-1).abs().mean().nan_to_num(0.0, 0.0, 0.0)


def gs_sdf_loss(gs_sdf, weight):                               # loss.cpp:7-11
    return 0.5 * (weight * gs_sdf.square()).sum()
