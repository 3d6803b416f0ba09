"""Host-side mirror of the reference's splat operator interface, on top of the C ABI.

Same operator names, argument order and error behaviour as the functions the reference's host code
calls into its (absent) `gsplat_cpp` submodule — /root/reference/include/neural_gaussian/
neural_gaussian.cpp:188-223 — so that the parity tests read like calls from `rasterization_2dgs_sdf`.
Each differentiable operator is a torch.autograd.Function whose forward/backward are single calls
into libgsdf_hip.so.  (The C++/libtorch flavour of the same layer lives in gs-sdf_amd/host/.)
"""
import torch

from . import capi
from .capi import f32, ptr


class _Timers:
    """Optional per-operator device timing with HIP events on the launch stream (bench.py's roofline leg)."""

    def __init__(self):
        self.on, self.ev, self.only = False, {}, None

    def enable(self, only=None):
        """only: optional set of operator names to time (two events per launch cost ~10 us of host time each)."""
        self.on, self.ev, self.only = True, {}, (None if only is None else set(only))

    def disable(self):
        self.on = False

    def start(self, name):
        if not self.on or (self.only is not None and name not in self.only):
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        self.ev.setdefault(name, []).append((a, b))
        return b

    @staticmethod
    def stop(b):
        if b is not None:
            b.record()

    def summary_ms(self, stat="mean"):
        torch.cuda.synchronize()
        out = {}
        for k, v in self.ev.items():
            ts = sorted(a.elapsed_time(b) for a, b in v)
            out[k] = (ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])) if stat == "median" \
                else sum(ts) / len(ts)
        return out

    def calls(self):
        return {k: len(v) for k, v in self.ev.items()}


TIMERS = _Timers()


def _timed(name, fn, *args):
    t = TIMERS.start(name)
    r = fn(*args)
    TIMERS.stop(t)
    return r


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ------------------------------------------------------------------------------------------------
# P1  fully_fused_projection_2dgs  (neural_gaussian.cpp:188-192)
# ------------------------------------------------------------------------------------------------
class _Projection2DGS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, near_plane, far_plane, radius_clip,
                sample_seed):
        L = capi.lib()
        means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
        viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
        N, C = means.shape[0], viewmats.shape[0]
        dev = means.device
        radii_dense = torch.empty(max(N * C, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(L.gsdf_projection_2dgs_ws_bytes(N, C), dtype=torch.uint8, device=dev)
        M = capi.count_via_host_word(lambda n_vis: capi.check(
            _timed("projection_2dgs_cull", L.gsdf_projection_2dgs_cull, N, C, f32(means, "means"), f32(quats, "quats"), f32(scales, "scales"),
                   f32(viewmats, "viewmats"), f32(Ks, "Ks"), width, height, near_plane, far_plane, radius_clip, ptr(radii_dense), ptr(ws), n_vis,
                   capi.stream()), "projection_2dgs_cull"), dev, upper=N * C)
        camera_ids = _empty((M,), torch.int64, means); gaussian_ids = _empty((M,), torch.int64, means)
        radii = _empty((M,), torch.int32, means); means2d = _empty((M, 2), torch.float32, means)
        depths = _empty((M,), torch.float32, means); rt = _empty((M, 3, 3), torch.float32, means)
        normals = _empty((M, 3), torch.float32, means); samples = _empty((M, 3), torch.float32, means)
        sw = _empty((M, 1), torch.float32, means)
        capi.check(_timed("projection_2dgs_fill", L.gsdf_projection_2dgs_fill, N, C, f32(means), f32(quats), f32(scales), f32(viewmats), f32(Ks),
                                               width, height, sample_seed, ptr(radii_dense), ptr(ws), M,
                                               ptr(camera_ids), ptr(gaussian_ids), ptr(radii), f32(means2d),
                                               f32(depths), f32(rt), f32(normals), f32(samples), f32(sw),
                                               capi.stream()), "projection_2dgs_fill")
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, camera_ids, gaussian_ids)
        ctx.dims = (width, height, sample_seed)
        ctx.mark_non_differentiable(camera_ids, gaussian_ids, radii, sw)
        return camera_ids, gaussian_ids, radii, means2d, depths, rt, normals, samples, sw

    @staticmethod
    def backward(ctx, _vc, _vg, _vr, v_means2d, v_depths, v_rt, v_normals, v_samples, _vsw):
        L = capi.lib()
        means, quats, scales, viewmats, Ks, camera_ids, gaussian_ids = ctx.saved_tensors
        width, height, seed = ctx.dims
        N, C, M = means.shape[0], viewmats.shape[0], camera_ids.shape[0]
        z = lambda g, shape: (torch.zeros(shape, dtype=torch.float32, device=means.device) if g is None
                              else g.contiguous())
        v_means2d, v_depths = z(v_means2d, (M, 2)), z(v_depths, (M,))
        v_rt, v_normals = z(v_rt, (M, 3, 3)), z(v_normals, (M, 3))
        v_samples = None if v_samples is None else v_samples.contiguous()
        v_means, v_quats, v_scales = torch.zeros_like(means), torch.zeros_like(quats), torch.zeros_like(scales)
        capi.check(_timed("projection_2dgs_bwd", L.gsdf_projection_2dgs_bwd, N, C, M, f32(means), f32(quats), f32(scales), f32(viewmats), f32(Ks),
                                              width, height, seed, ptr(camera_ids), ptr(gaussian_ids),
                                              f32(v_means2d), f32(v_depths), f32(v_rt), f32(v_normals),
                                              f32(v_samples), f32(v_means), f32(v_quats), f32(v_scales),
                                              capi.stream()), "projection_2dgs_bwd")
        return v_means, v_quats, v_scales, None, None, None, None, None, None, None, None


def fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, width, height, near_plane, far_plane,
                                radius_clip, packed=True, sparse_grad=False, sample_seed=0):
    """-> (camera_ids, gaussian_ids, radii, means2d, depths, ray_transforms, normals, samples, samples_weights).
    Only the reference's configuration is implemented: packed=True, sparse_grad=False
    (neural_gaussian.cpp:526-530)."""
    if not packed:
        raise RuntimeError("fully_fused_projection_2dgs: only packed=True is implemented (reference uses packed)")
    if sparse_grad:
        raise RuntimeError("fully_fused_projection_2dgs: sparse_grad=True is not implemented (reference passes false)")
    N = means.shape[0]
    if tuple(means.shape) != (N, 3) or tuple(quats.shape) != (N, 4) or tuple(scales.shape) != (N, 3):
        raise RuntimeError("fully_fused_projection_2dgs: invalid means/quats/scales shape")
    C = viewmats.shape[0]
    if tuple(viewmats.shape) != (C, 4, 4) or tuple(Ks.shape) != (C, 3, 3):
        raise RuntimeError("fully_fused_projection_2dgs: invalid viewmats/Ks shape")
    return _Projection2DGS.apply(means, quats, scales, viewmats, Ks, int(width), int(height), float(near_plane),
                                 float(far_plane), float(radius_clip), int(sample_seed))


# ------------------------------------------------------------------------------------------------
# P2  gsplat_cpp::get_view_colors  (neural_gaussian.cpp:199-200)
# ------------------------------------------------------------------------------------------------
class _ViewColors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, viewmats, means, sh, camera_ids, gaussian_ids, sh_degree):
        L = capi.lib()
        viewmats, means, sh = viewmats.contiguous(), means.contiguous(), sh.contiguous()
        M, K = camera_ids.shape[0], sh.shape[1]
        colors = _empty((M, 3), torch.float32, means)
        capi.check(_timed("view_colors_fwd", L.gsdf_view_colors_fwd, M, K, sh_degree, f32(viewmats), f32(means), f32(sh), ptr(camera_ids, torch.int64),
                                          ptr(gaussian_ids, torch.int64), f32(colors), capi.stream()), "view_colors_fwd")
        ctx.save_for_backward(viewmats, means, sh, camera_ids, gaussian_ids)
        ctx.sh_degree = sh_degree
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        L = capi.lib()
        viewmats, means, sh, camera_ids, gaussian_ids = ctx.saved_tensors
        M, K = camera_ids.shape[0], sh.shape[1]
        v_sh, v_means = torch.zeros_like(sh), torch.zeros_like(means)
        capi.check(_timed("view_colors_bwd", L.gsdf_view_colors_bwd, M, K, ctx.sh_degree, f32(viewmats), f32(means), f32(sh), ptr(camera_ids),
                                          ptr(gaussian_ids), f32(v_colors.contiguous()), f32(v_sh), f32(v_means),
                                          1 if viewmats.shape[0] == 1 else 0, capi.stream()), "view_colors_bwd")
        return None, v_means, v_sh, None, None, None


def get_view_colors(viewmats, means, radii, colors, camera_ids, gaussian_ids, sh_degree=None):
    """colors: SH coefficients [N,K,3] when sh_degree is given, else post-activation colours [N,D]."""
    if sh_degree is None:
        return colors.index_select(0, gaussian_ids)
    if colors.dim() != 3 or colors.shape[2] != 3 or (sh_degree + 1) ** 2 > colors.shape[1]:
        raise RuntimeError("get_view_colors: invalid colors shape")
    return _ViewColors.apply(viewmats, means, colors, camera_ids, gaussian_ids, int(sh_degree))


# ------------------------------------------------------------------------------------------------
# P3  gsplat_cpp::tile_encode  (neural_gaussian.cpp:207-209) — non-differentiable, integer outputs
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def tile_encode(width, height, tile_size, means2d, radii, depths, packed, C, camera_ids, gaussian_ids,
                return_isect_ids=False):
    if not packed:
        raise RuntimeError("tile_encode: only packed=True is implemented (reference uses packed)")
    L = capi.lib()
    means2d, radii, depths = means2d.detach().contiguous(), radii.contiguous(), depths.detach().contiguous()
    M = radii.shape[0]
    dev = means2d.device
    tw, th = (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size
    tpg = _empty((M,), torch.int32, means2d); cum = _empty((max(M, 1),), torch.int64, means2d)
    ws = torch.empty(L.gsdf_tile_count_ws_bytes(M), dtype=torch.uint8, device=dev)
    I = capi.count_via_host_word(lambda n_is: capi.check(
        _timed("tile_count", L.gsdf_tile_count, M, width, height, tile_size, f32(means2d), ptr(radii, torch.int32), ptr(tpg), ptr(cum), ptr(ws), n_is,
               capi.stream()), "tile_count"), dev, upper=M * tw * th)
    isect_ids = _empty((I,), torch.int64, means2d); flatten_ids = _empty((I,), torch.int32, means2d)
    offsets = _empty((C, th, tw), torch.int32, means2d)
    ws2 = torch.empty(L.gsdf_tile_encode_ws_bytes(M, I), dtype=torch.uint8, device=dev)
    capi.check(_timed("tile_encode", L.gsdf_tile_encode, M, C, I, width, height, tile_size, f32(means2d), ptr(radii), f32(depths),
                                  ptr(camera_ids, torch.int64), ptr(cum), ptr(ws2), ptr(isect_ids), ptr(flatten_ids),
                                  ptr(offsets), capi.stream()), "tile_encode")
    if return_isect_ids:
        return tpg, flatten_ids, offsets, isect_ids
    return tpg, flatten_ids, offsets


# ------------------------------------------------------------------------------------------------
# P4  rasterize_to_pixels_2dgs  (neural_gaussian.cpp:215-223)
# ------------------------------------------------------------------------------------------------
class _Rasterize2DGS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, ray_transforms, colors, opacities, normals, densify, means2d_absgrad, backgrounds, masks,
                width, height, tile_size, isect_offsets, flatten_ids):
        L = capi.lib()
        means2d, rt, colors = means2d.contiguous(), ray_transforms.contiguous(), colors.contiguous()
        opacities, normals = opacities.contiguous(), normals.contiguous()
        C, M, I = isect_offsets.shape[0], opacities.shape[0], flatten_ids.shape[0]
        bg = None if backgrounds is None else backgrounds.contiguous()
        mk = None if masks is None else masks.to(torch.uint8).contiguous()
        e = lambda *s: _empty(s, torch.float32, means2d)
        rc, rd, ra, rn, rm = e(C, height, width, 3), e(C, height, width, 1), e(C, height, width, 1), e(C, height, width, 3), e(C, height, width, 1)
        last = _empty((C, height, width), torch.int32, means2d); med = _empty((C, height, width), torch.int32, means2d)
        vis = _empty((M, 1), torch.float32, means2d)
        # the transmittance each pixel ended with, kept for the backward: render_alphas = 1 - T cannot give it back once T << 1
        need = any(ctx.needs_input_grad[:7])
        fT = _empty((C, height, width), torch.float32, means2d) if need else None
        # packed splat records + reach masks of the (tile, splat) pairs: written by the forward, reused by the backward
        fws = torch.empty(L.gsdf_rasterize_2dgs_fwd_ws_bytes(M, I), dtype=torch.uint8, device=means2d.device)
        capi.check(_timed("rasterize_2dgs_fwd", L.gsdf_rasterize_2dgs_fwd, C, M, I, width, height, tile_size, f32(means2d), f32(rt), f32(colors),
                                             f32(opacities), f32(normals), f32(bg), ptr(mk), ptr(isect_offsets, torch.int32),
                                             ptr(flatten_ids, torch.int32), f32(rc), f32(rd), f32(ra), f32(rn), f32(rm),
                                             ptr(last), ptr(med), f32(vis), f32(fT), ptr(fws), capi.stream()), "rasterize_2dgs_fwd")
        ctx.save_for_backward(means2d, rt, colors, opacities, normals, bg, mk, isect_offsets, flatten_ids, ra, last, med, fT, fws if need else None)
        ctx.dims = (width, height, tile_size)
        ctx.absgrad = bool(ctx.needs_input_grad[6])
        distort = torch.zeros((C, height, width, 1), dtype=torch.float32, device=means2d.device)
        ctx.mark_non_differentiable(vis, distort)
        return rc, rd, ra, rn, distort, rm, vis

    @staticmethod
    def backward(ctx, v_rc, v_rd, v_ra, v_rn, _v_dist, v_rm, _v_vis):
        L = capi.lib()
        means2d, rt, colors, opacities, normals, bg, mk, isect_offsets, flatten_ids, ra, last, med, fT, fws = ctx.saved_tensors
        width, height, tile_size = ctx.dims
        C, M, I = isect_offsets.shape[0], opacities.shape[0], flatten_ids.shape[0]
        zz = lambda g, ch: (torch.zeros((C, height, width, ch), dtype=torch.float32, device=means2d.device) if g is None
                            else g.contiguous())
        v_rc, v_rd, v_ra, v_rn, v_rm = zz(v_rc, 3), zz(v_rd, 1), zz(v_ra, 1), zz(v_rn, 3), zz(v_rm, 1)
        e = lambda *s: _empty(s, torch.float32, means2d)
        v_means2d, v_rt, v_colors, v_opac, v_normals, v_dens = e(M, 2), e(M, 3, 3), e(M, 3), e(M), e(M, 3), e(M, 2)
        v_abs = e(M, 2) if ctx.absgrad else None
        ws = torch.empty(L.gsdf_rasterize_2dgs_bwd_ws_bytes(M, I), dtype=torch.uint8, device=means2d.device)
        capi.check(_timed("rasterize_2dgs_bwd", L.gsdf_rasterize_2dgs_bwd, C, M, I, width, height, tile_size, f32(means2d), f32(rt), f32(colors),
                                             f32(opacities), f32(normals), f32(bg), ptr(mk), ptr(isect_offsets),
                                             ptr(flatten_ids), f32(ra), ptr(last), ptr(med), f32(v_rc), f32(v_rd), f32(v_ra),
                                             f32(v_rn), f32(v_rm), f32(v_means2d), f32(v_rt), f32(v_colors), f32(v_opac),
                                             f32(v_normals), f32(v_dens), f32(v_abs), ptr(ws), f32(fT), ptr(fws), capi.stream()), "rasterize_2dgs_bwd")
        return (v_means2d, v_rt, v_colors, v_opac, v_normals, v_dens, v_abs, None, None, None, None, None, None, None)


def rasterize_fwd_instr(means2d, ray_transforms, colors, opacities, normals, width, height, isect_offsets, flatten_ids, backgrounds=None,
                        masks=None, counters=None, trace_rows=None, trace_stride=0):
    """Instrumented forward launch (include/gsdf_hip.h: gsdf_rasterize_2dgs_fwd_instr; tests / diagnostics, no autograd).
    counters: int64 [16] device tensor the launch adds to, OR trace_rows int32 [C,H,W] + trace_stride: the decision record of the
    parity gate (`trace_bits` uint8 [rows, stride] in the result).  -> dict of the operator's outputs (+ last_ids, median_ids, final_T)."""
    import ctypes
    L = capi.lib()
    C, M, I = isect_offsets.shape[0], opacities.shape[0], flatten_ids.shape[0]
    e = lambda *s: _empty(s, torch.float32, means2d)
    rc, rd, ra, rn, rm = e(C, height, width, 3), e(C, height, width, 1), e(C, height, width, 1), e(C, height, width, 3), e(C, height, width, 1)
    last = _empty((C, height, width), torch.int32, means2d); med = _empty((C, height, width), torch.int32, means2d)
    vis, fT = _empty((M, 1), torch.float32, means2d), _empty((C, height, width), torch.float32, means2d)
    fws = torch.empty(L.gsdf_rasterize_2dgs_fwd_ws_bytes(M, I), dtype=torch.uint8, device=means2d.device)
    bits = None
    if trace_rows is not None:
        n_rows = int(trace_rows.max().item()) + 1
        bits = torch.zeros((max(n_rows, 1), int(trace_stride)), dtype=torch.uint8, device=means2d.device)
    instr = capi.RasterInstr(capi.ptr(counters).value if counters is not None else None,
                             capi.ptr(trace_rows, torch.int32).value if trace_rows is not None else None, int(trace_stride),
                             capi.ptr(bits).value if bits is not None else None)
    mk = None if masks is None else masks.to(torch.uint8).contiguous()
    capi.check(L.gsdf_rasterize_2dgs_fwd_instr(C, M, I, width, height, 16, f32(means2d), f32(ray_transforms), f32(colors), f32(opacities),
                                               f32(normals), f32(backgrounds), ptr(mk), ptr(isect_offsets, torch.int32),
                                               ptr(flatten_ids, torch.int32), f32(rc), f32(rd), f32(ra), f32(rn), f32(rm), ptr(last), ptr(med),
                                               f32(vis), f32(fT), ptr(fws), ctypes.cast(ctypes.pointer(instr), ctypes.c_void_p), capi.stream()),
               "rasterize_2dgs_fwd_instr")
    return dict(render_colors=rc, render_depths=rd, render_alphas=ra, render_normals=rn, render_median=rm, last_ids=last, median_ids=med,
                visibilities=vis, final_T=fT, trace_bits=bits, fwd_ws=fws)


def rasterize_bwd_instr(means2d, ray_transforms, colors, opacities, normals, width, height, isect_offsets, flatten_ids, fwd, upstream,
                        counters, backgrounds=None, masks=None, absgrad=False):
    """Instrumented backward launch (gsdf_rasterize_2dgs_bwd_instr): `fwd` = the dict of rasterize_fwd_instr, `upstream` = dict of the
    five v_render_* tensors.  -> dict of the operator's gradients."""
    import ctypes
    L = capi.lib()
    C, M, I = isect_offsets.shape[0], opacities.shape[0], flatten_ids.shape[0]
    e = lambda *s: _empty(s, torch.float32, means2d)
    g = dict(v_means2d=e(M, 2), v_ray_transforms=e(M, 3, 3), v_colors=e(M, 3), v_opacities=e(M), v_normals=e(M, 3), v_densify=e(M, 2),
             v_means2d_abs=e(M, 2) if absgrad else None)
    ws = torch.empty(L.gsdf_rasterize_2dgs_bwd_ws_bytes(M, I), dtype=torch.uint8, device=means2d.device)
    instr = capi.RasterInstr(capi.ptr(counters).value if counters is not None else None, None, 0, None)
    mk = None if masks is None else masks.to(torch.uint8).contiguous()
    u = lambda k: f32(upstream[k].contiguous())
    capi.check(L.gsdf_rasterize_2dgs_bwd_instr(C, M, I, width, height, 16, f32(means2d), f32(ray_transforms), f32(colors), f32(opacities),
                                               f32(normals), f32(backgrounds), ptr(mk), ptr(isect_offsets), ptr(flatten_ids),
                                               f32(fwd["render_alphas"]), ptr(fwd["last_ids"]), ptr(fwd["median_ids"]),
                                               u("v_render_colors"), u("v_render_depths"), u("v_render_alphas"), u("v_render_normals"),
                                               u("v_render_median"), f32(g["v_means2d"]), f32(g["v_ray_transforms"]), f32(g["v_colors"]),
                                               f32(g["v_opacities"]), f32(g["v_normals"]), f32(g["v_densify"]), f32(g["v_means2d_abs"]), ptr(ws),
                                               f32(fwd["final_T"]), ptr(fwd.get("fwd_ws")), ctypes.cast(ctypes.pointer(instr), ctypes.c_void_p),
                                               capi.stream()),
               "rasterize_2dgs_bwd_instr")
    return g


def rasterize_to_pixels_2dgs(means2d, ray_transforms, colors, opacities, normals, densify, width, height, tile_size,
                             isect_offsets, flatten_ids, backgrounds=None, masks=None, packed=True,
                             means2d_absgrad=None, distloss=False):
    """-> (render_colors [C,H,W,3], render_depths [C,H,W,1], render_alphas [C,H,W,1], render_normals [C,H,W,3],
    render_distort, render_median [C,H,W,1], visibilities [M,1]).  `densify` and `means2d_absgrad` are leaf
    tensors whose .grad is read by the caller after backward (neural_gaussian.cpp:215-217, 626-633)."""
    if distloss:
        raise RuntimeError("rasterize_to_pixels_2dgs: distloss=True is not implemented (reference passes false)")
    if not packed:
        raise RuntimeError("rasterize_to_pixels_2dgs: only packed=True is implemented")
    if colors.shape[-1] != 3:
        raise RuntimeError("rasterize_to_pixels_2dgs: colors must be [M,3]")
    if means2d_absgrad is None:
        means2d_absgrad = torch.zeros_like(means2d)
    return _Rasterize2DGS.apply(means2d, ray_transforms, colors, opacities, normals, densify, means2d_absgrad,
                                backgrounds, masks, int(width), int(height), int(tile_size), isect_offsets, flatten_ids)


def ssim_window(window_size=11, sigma=1.5):
    """The reference's 1-D window (loss_utils.cpp:6-14): exp(-floor((x - window_size)/2)^2 / (2 sigma^2)), normalised —
    NOT the symmetric Gaussian; reproduced verbatim."""
    import math
    g = [math.exp(-(math.floor((x - window_size) / 2.0) ** 2) / (2.0 * sigma * sigma)) for x in range(window_size)]
    s = sum(g)
    return [v / s for v in g]


class _L1DSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, w_l1, w_ssim):
        import ctypes as C
        L = capi.lib()
        img, gt = img.contiguous(), gt.contiguous()
        H, W = img.shape[0], img.shape[1]
        win = (C.c_float * 11)(*ssim_window())
        sums = torch.empty(2, dtype=torch.float32, device=img.device)
        maps = torch.empty(3, H, W, 3, dtype=torch.float32, device=img.device) if ctx.needs_input_grad[0] else None
        capi.check(_timed("l1_dssim_fwd", L.gsdf_l1_dssim_fwd, H, W, f32(img, "image"), f32(gt, "gt"), win, f32(sums),
                          f32(maps), capi.stream()), "l1_dssim_fwd")
        ctx.save_for_backward(img, gt, maps)
        ctx.w = (float(w_l1), float(w_ssim))
        n = 3.0 * H * W
        return w_l1 * sums[0] / n + w_ssim * (1.0 - sums[1] / n)

    @staticmethod
    def backward(ctx, v_loss):
        import ctypes as C
        L = capi.lib()
        img, gt, maps = ctx.saved_tensors
        H, W = img.shape[0], img.shape[1]
        win = (C.c_float * 11)(*ssim_window())
        v_img = torch.empty_like(img)
        capi.check(_timed("l1_dssim_bwd", L.gsdf_l1_dssim_bwd, H, W, f32(img), f32(gt), win, f32(maps),
                          f32(v_loss.contiguous().reshape(1)), ctx.w[0], ctx.w[1], f32(v_img), capi.stream()), "l1_dssim_bwd")
        return v_img, None, None, None


def l1_dssim_loss(render_color, gt_color, rgb_weight=0.8, dssim_weight=0.2):
    """k_rgb_weight * loss::rgb_loss + k_dssim_weight * loss::dssim_loss on [H,W,3] images (neural_mapping.cpp:237-240,
    weights config/base.yaml:35-36), one fused HIP kernel each way."""
    if render_color.dim() != 3 or render_color.shape[2] != 3 or render_color.shape != gt_color.shape:
        raise RuntimeError("l1_dssim_loss: expected two [H,W,3] images")
    return _L1DSSIM.apply(render_color, gt_color, rgb_weight, dssim_weight)


class _NormalConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, alpha, render_normal, intr, pose):
        import ctypes as C
        L = capi.lib()
        depth, alpha, rn = depth.contiguous(), alpha.detach().contiguous(), render_normal.contiguous()
        H, W = depth.shape[0], depth.shape[1]
        ci, cp = (C.c_float * 4)(*intr), (C.c_float * 12)(*pose)
        loss = torch.empty(1, dtype=torch.float32, device=depth.device)
        capi.check(_timed("normal_consistency_fwd", L.gsdf_normal_consistency_fwd, H, W, ci, cp, f32(depth, "depth"), f32(alpha),
                          f32(rn), f32(loss), capi.stream()), "normal_consistency_fwd")
        ctx.save_for_backward(depth, alpha, rn)
        ctx.cam = (intr, pose)
        return loss[0]

    @staticmethod
    def backward(ctx, v_loss):
        import ctypes as C
        L = capi.lib()
        depth, alpha, rn = ctx.saved_tensors
        H, W = depth.shape[0], depth.shape[1]
        ci, cp = (C.c_float * 4)(*ctx.cam[0]), (C.c_float * 12)(*ctx.cam[1])
        v_depth, v_rn = torch.empty_like(depth), torch.empty_like(rn)
        capi.check(_timed("normal_consistency_bwd", L.gsdf_normal_consistency_bwd, H, W, ci, cp, f32(depth), f32(alpha), f32(rn),
                          f32(v_loss.contiguous().reshape(1)), f32(v_depth), f32(v_rn), capi.stream()), "normal_consistency_bwd")
        return v_depth, None, v_rn, None, None


def normal_consistency_loss(depth, alpha, render_normal, fx, fy, cx, cy, pose_cam2world):
    """mean(alpha^2 - nan_to_num((depth_to_normal(depth) * alpha) . render_normal)) with alpha detached
    (neural_mapping.cpp:243-266; cameras.hpp:176-226).  depth, alpha [H,W,1]; render_normal [H,W,3] (world);
    pose_cam2world [3,4] tensor or 12 host floats (row-major; the values are baked into the launch)."""
    # a tensor costs a device->host copy per call; a trainer that knows its poses ahead passes the 12 host floats
    pose = [float(v) for v in (pose_cam2world.detach().cpu().reshape(-1)[:12] if torch.is_tensor(pose_cam2world) else pose_cam2world)]
    return _NormalConsistency.apply(depth, alpha, render_normal, (float(fx), float(fy), float(cx), float(cy)), tuple(pose))


class _Isotropic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scales, gaussian_ids):
        L = capi.lib()
        scales, ids = scales.contiguous(), gaussian_ids.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=scales.device)
        capi.check(_timed("isotropic_loss", L.gsdf_isotropic_loss_fwd, ids.shape[0], f32(scales), ptr(ids, torch.int64), f32(loss),
                          capi.stream()), "isotropic_loss_fwd")
        ctx.save_for_backward(scales, ids)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v_loss):
        scales, ids = ctx.saved_tensors
        v_scales = torch.zeros_like(scales)
        capi.check(_timed("isotropic_loss_bwd", capi.lib().gsdf_isotropic_loss_bwd, ids.shape[0], f32(scales), ptr(ids, torch.int64),
                          f32(v_loss.contiguous().reshape(1)), f32(v_scales), capi.stream()), "isotropic_loss_bwd")
        return v_scales, None


def isotropic_loss(scales, gaussian_ids):
    """(scale - scale.mean(-1, True)).abs().mean() with scale = scales[gaussian_ids][:, 0:2] (neural_mapping.cpp:268-276);
    scales [N,3] (activated), gaussian_ids int64 [M]."""
    if scales.dim() != 2 or scales.shape[1] != 3:
        raise RuntimeError("isotropic_loss: scales must be [N,3]")
    return _Isotropic.apply(scales, gaussian_ids)


@torch.no_grad()
def nan_rows(offsets, scaling, quaternion, want_mask=False):
    """prune_nan_gs's test (neural_gaussian.cpp:907-916) in one launch -> (count int32 device scalar [1], mask bool [N] or None)."""
    n = offsets.shape[0]
    count = torch.empty(1, dtype=torch.int32, device=offsets.device)
    mask = torch.empty(n, dtype=torch.uint8, device=offsets.device) if want_mask else None
    capi.check(_timed("nan_rows", capi.lib().gsdf_nan_rows, n, f32(offsets.contiguous()), f32(scaling.contiguous()),
                      f32(quaternion.contiguous()), ptr(count), ptr(mask), capi.stream()), "nan_rows")
    return count, (None if mask is None else mask.bool())


@torch.no_grad()
def distCUDA2(points):
    """simple-knn's distCUDA2 (neural_gaussian.cpp:314): mean squared distance to the 3 nearest neighbours, [N,3] -> [N]."""
    L = capi.lib()
    points = points.detach().contiguous()
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("distCUDA2: expected [N,3]")
    N = points.shape[0]
    out = torch.empty(N, dtype=torch.float32, device=points.device)
    ws = torch.empty(L.gsdf_knn_ws_bytes(N), dtype=torch.uint8, device=points.device)
    capi.check(_timed("knn_mean_dist2", L.gsdf_knn_mean_dist2, N, f32(points, "points"), f32(out), ptr(ws), capi.stream()),
               "distCUDA2")
    return out


class _RenderPost(torch.autograd.Function):
    """Fused per-pixel epilogue (neural_gaussian.cpp:229-240): depth/alpha + nan_to_num, cat, normals to world.
    Also emits the two slices NeuralGS::render takes of the result (colour [.,3], depth [.,1], :533-543) so that callers
    need no slicing kernels (and no zero-fill + add in their backward)."""

    @staticmethod
    def forward(ctx, rc, rd, ra, rn, viewmats, expected_depth):
        L = capi.lib()
        rc, rd, ra, rn, viewmats = rc.contiguous(), rd.contiguous(), ra.contiguous(), rn.contiguous(), viewmats.contiguous()
        n_pix = ra.numel()
        renders = torch.empty(*ra.shape[:-1], 4, dtype=torch.float32, device=ra.device)
        nw, color3, depth1 = torch.empty_like(rn), torch.empty_like(rc), torch.empty_like(rd)
        capi.check(_timed("render_post_fwd", L.gsdf_render_post_fwd, n_pix, int(expected_depth), f32(viewmats), f32(rc),
                          f32(rd), f32(ra), f32(rn), f32(renders), f32(nw), f32(color3), f32(depth1), capi.stream()),
                   "render_post_fwd")
        ctx.save_for_backward(rd, ra, viewmats)
        ctx.expected_depth = int(expected_depth)
        return renders, nw, color3, depth1

    @staticmethod
    def backward(ctx, v_renders, v_nw, v_c3, v_d1):
        L = capi.lib()
        rd, ra, viewmats = ctx.saved_tensors
        n_pix = ra.numel()
        c = lambda g: None if g is None else g.contiguous()
        v_rc, v_rn = torch.empty(*ra.shape[:-1], 3, device=ra.device), torch.empty(*ra.shape[:-1], 3, device=ra.device)
        v_rd, v_ra = torch.empty_like(ra), torch.empty_like(ra)
        capi.check(_timed("render_post_bwd", L.gsdf_render_post_bwd, n_pix, ctx.expected_depth, f32(viewmats), f32(rd),
                          f32(ra), f32(c(v_renders)), f32(c(v_nw)), f32(c(v_c3)), f32(c(v_d1)), f32(v_rc), f32(v_rd), f32(v_ra),
                          f32(v_rn), capi.stream()), "render_post_bwd")
        return v_rc, v_rd, v_ra, v_rn, None, None


class _GatherRows(torch.autograd.Function):
    """x[ids] for row ids that are unique per camera (packed mode): the backward is an index_add (atomics) instead
    of the sort-based index_put the generic advanced-indexing backward launches."""

    @staticmethod
    def forward(ctx, x, ids):
        ctx.save_for_backward(ids)
        ctx.n = x.shape[0]
        return x.index_select(0, ids)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        out = torch.zeros((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        return out.index_add_(0, ids, g), None


# ------------------------------------------------------------------------------------------------
# rasterization_2dgs_sdf: host orchestration of P1 -> P2 -> P3 -> P4 (neural_gaussian.cpp:129-271)
# ------------------------------------------------------------------------------------------------
def rasterization_2dgs_sdf(means, quats, scales, opacities, colors, viewmats, Ks, width, height, render_mode="RGB+ED",
                           near_plane=0.01, far_plane=1e10, radius_clip=0.0, sh_degree=None, packed=True, tile_size=16,
                           backgrounds=None, sparse_grad=False, absgrad=False, distloss=False, center_reg=False,
                           sample_seed=0, samples_gate=None):
    """Mirror of the reference's `rasterization_2dgs_sdf` -> (render_colors [C,H,W,4], render_alphas, meta).
    `samples_gate` (a trainer.GradGate, not in the reference): meta["samples"] passes through trainer.join_grad HERE, i.e.
    before the binning / compositing nodes are created, so that in the backward pass its gradient is consumed (and the
    gate's event waited for) only after the compositing backward has been issued."""
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise RuntimeError("Invalid render_mode")
    N, C = means.shape[0], viewmats.shape[0]
    if tuple(opacities.shape) != (N,):
        raise RuntimeError("Invalid opacities shape")
    (camera_ids, gaussian_ids, radii, means2d, depths, ray_transforms, normals, samples,
     samples_weights) = fully_fused_projection_2dgs(means, quats, scales, viewmats, Ks, width, height, near_plane,
                                                    far_plane, radius_clip, packed, sparse_grad,
                                                    0 if center_reg else sample_seed)
    if center_reg:
        samples, samples_weights = means.index_select(0, gaussian_ids), torch.ones_like(samples_weights)
    if samples_gate is not None:
        from .trainer import join_grad
        samples = join_grad(samples, samples_gate)
    pt_opacities = _GatherRows.apply(opacities, gaussian_ids)
    pt_colors = get_view_colors(viewmats, means, radii, colors, camera_ids, gaussian_ids, sh_degree)
    tiles_per_gauss, flatten_ids, isect_offsets = tile_encode(width, height, tile_size, means2d, radii, depths, packed,
                                                              C, camera_ids, gaussian_ids)
    means2d_absgrad = torch.zeros_like(means2d).requires_grad_(absgrad)
    densify = torch.zeros_like(means2d).requires_grad_(True)
    (render_colors, render_depths, render_alphas, render_normals, render_distort, render_median,
     visibilities) = rasterize_to_pixels_2dgs(means2d, ray_transforms, pt_colors, pt_opacities, normals, densify,
                                              width, height, tile_size, isect_offsets, flatten_ids, backgrounds, None,
                                              packed, means2d_absgrad, distloss)
    meta = {}
    if absgrad:
        meta["absgrad"] = means2d_absgrad
    # neural_gaussian.cpp:229-240 in one fused pass: expected depth, cat(colours, depth), normals to world space
    render_colors, render_normals, color3, depth1 = _RenderPost.apply(render_colors, render_depths, render_alphas,
                                                                      render_normals, viewmats, render_mode in ("ED", "RGB+ED"))
    meta.update(color=color3, depth=depth1)      # the slices renders[...,0:3] / [...,3:4] of neural_gaussian.cpp:533-543
    meta.update(render_normal=render_normals, render_median=render_median, normal=normals, gaussian_ids=gaussian_ids,
                radii=radii, gradient_2dgs=densify, width=torch.tensor([width]), height=torch.tensor([height]),
                n_cameras=torch.tensor([C]), samples=samples, samples_weights=samples_weights,
                samples_opacities=pt_opacities, visibilities=visibilities,
                # extras (not in the reference's meta): sizes that fix the roofline byte count
                tiles_per_gauss=tiles_per_gauss, flatten_ids=flatten_ids, isect_offsets=isect_offsets)
    return render_colors, render_alphas, meta
