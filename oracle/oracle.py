"""ctypes front-end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY — see oracle/README.md.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package never does.

Every function takes/returns numpy arrays.  `prec` selects the build: "f32" (bit-exact
integer parity target, CPU baseline) or "f64" (gradient checking).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIBS = {}
F32_BUILDS = ("f32", "f32fma", "f32acc", "f32fmaacc")


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    kinds = [k for k in ("splat", "sdf") if os.path.exists(os.path.join(_HERE, f"{k}_oracle.c"))]
    targets = [f"_build/liborc_{k}_{p}.so" for k in kinds for p in ("f32", "f64")] + [f"_build/liborc_splat_{p}.so" for p in ("f32fma", "f32acc", "f32fmaacc")]
    if os.path.exists(os.path.join(_HERE, "occ_oracle.c")):
        targets.append("_build/liborc_occ.so")
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + targets)


def _lib(kind, prec):
    key = (kind, prec)
    if key not in _LIBS:
        path = os.path.join(_BUILD, f"liborc_{kind}_{prec}.so")
        if not os.path.exists(path):
            build()
        _LIBS[key] = C.CDLL(path)
    return _LIBS[key]


def _dt(prec):
    return np.float32 if prec in F32_BUILDS else np.float64


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _r(x, prec):
    return C.c_float(x) if prec in F32_BUILDS else C.c_double(x)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------------------------
# P1 projection
# ----------------------------------------------------------------------------------------------
def projection_2dgs_fwd(means, quats, scales, viewmats, Ks, W, H, near=0.05, far=300.0, radius_clip=0.0,
                        seed=0, prec="f32"):
    dt = _dt(prec)
    means, quats, scales, viewmats, Ks = (_c(a, dt) for a in (means, quats, scales, viewmats, Ks))
    N, Cn = means.shape[0], viewmats.shape[0]
    cap = max(N * Cn, 1)
    cam = np.zeros(cap, np.int64); gid = np.zeros(cap, np.int64); radii = np.zeros(cap, np.int32)
    m2d = np.zeros((cap, 2), dt); dep = np.zeros(cap, dt); rt = np.zeros((cap, 3, 3), dt)
    nrm = np.zeros((cap, 3), dt); smp = np.zeros((cap, 3), dt); sw = np.zeros((cap, 1), dt)
    f = _lib("splat", prec).orc_projection_2dgs_fwd
    f.restype = C.c_int64
    M = f(C.c_int64(N), C.c_int64(Cn), _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks), C.c_int(W),
          C.c_int(H), _r(near, prec), _r(far, prec), _r(radius_clip, prec), C.c_uint64(seed), _p(cam), _p(gid),
          _p(radii), _p(m2d), _p(dep), _p(rt), _p(nrm), _p(smp), _p(sw))
    return dict(camera_ids=cam[:M].copy(), gaussian_ids=gid[:M].copy(), radii=radii[:M].copy(),
                means2d=m2d[:M].copy(), depths=dep[:M].copy(), ray_transforms=rt[:M].copy(),
                normals=nrm[:M].copy(), samples=smp[:M].copy(), samples_weights=sw[:M].copy())


def projection_2dgs_bwd(means, quats, scales, viewmats, Ks, W, H, camera_ids, gaussian_ids, v_means2d, v_depths,
                        v_ray_transforms, v_normals, v_samples=None, seed=0, prec="f32"):
    dt = _dt(prec)
    means, quats, scales, viewmats, Ks = (_c(a, dt) for a in (means, quats, scales, viewmats, Ks))
    N, Cn, M = means.shape[0], viewmats.shape[0], camera_ids.shape[0]
    cam, gid = _c(camera_ids, np.int64), _c(gaussian_ids, np.int64)
    v_means2d, v_depths, v_rt, v_normals, v_samples = (
        _c(a, dt) for a in (v_means2d, v_depths, v_ray_transforms, v_normals, v_samples))
    vm = np.zeros((N, 3), dt); vq = np.zeros((N, 4), dt); vs = np.zeros((N, 3), dt)
    _lib("splat", prec).orc_projection_2dgs_bwd(
        C.c_int64(N), C.c_int64(Cn), C.c_int64(M), _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks),
        C.c_int(W), C.c_int(H), C.c_uint64(seed), _p(cam), _p(gid), _p(v_means2d), _p(v_depths), _p(v_rt),
        _p(v_normals), _p(v_samples), _p(vm), _p(vq), _p(vs))
    return vm, vq, vs


def projection_2dgs_bwd_bound(means, quats, scales, viewmats, Ks, W, H, camera_ids, gaussian_ids, a_means2d, a_depths,
                              a_ray_transforms, a_normals, a_samples=None, seed=0):
    """Absolute shadow of projection_2dgs_bwd (fp64 build): sum of the absolute values of all terms entering every output, given the
    absolute upstream gradients (or their error bounds).  -> (b_means [N,3], b_quats [N,4], b_scales [N,3])"""
    dt = np.float64
    means, quats, scales, viewmats, Ks = (_c(a, dt) for a in (means, quats, scales, viewmats, Ks))
    N, Cn, M = means.shape[0], viewmats.shape[0], camera_ids.shape[0]
    cam, gid = _c(camera_ids, np.int64), _c(gaussian_ids, np.int64)
    a2, ad, art, an, asm = (_c(a, dt) for a in (a_means2d, a_depths, a_ray_transforms, a_normals, a_samples))
    bm = np.zeros((N, 3), dt); bq = np.zeros((N, 4), dt); bs = np.zeros((N, 3), dt)
    _lib("splat", "f64").orc_projection_2dgs_bwd_bound(
        C.c_int64(N), C.c_int64(Cn), C.c_int64(M), _p(means), _p(quats), _p(scales), _p(viewmats), _p(Ks),
        C.c_int(W), C.c_int(H), C.c_uint64(seed), _p(cam), _p(gid), _p(a2), _p(ad), _p(art), _p(an), _p(asm), _p(bm), _p(bq), _p(bs))
    return bm, bq, bs


# ----------------------------------------------------------------------------------------------
# P2 view colours
# ----------------------------------------------------------------------------------------------
def view_colors_fwd(viewmats, means, sh_coeffs, camera_ids, gaussian_ids, sh_degree, prec="f32"):
    dt = _dt(prec)
    viewmats, means, sh = _c(viewmats, dt), _c(means, dt), _c(sh_coeffs, dt)
    cam, gid = _c(camera_ids, np.int64), _c(gaussian_ids, np.int64)
    M, K = cam.shape[0], sh.shape[1]
    out = np.zeros((M, 3), dt)
    _lib("splat", prec).orc_view_colors_fwd(C.c_int64(M), C.c_int64(K), C.c_int(sh_degree), _p(viewmats),
                                            _p(means), _p(sh), _p(cam), _p(gid), _p(out))
    return out


def view_colors_bwd(viewmats, means, sh_coeffs, camera_ids, gaussian_ids, sh_degree, v_colors, prec="f32"):
    dt = _dt(prec)
    viewmats, means, sh, v_colors = _c(viewmats, dt), _c(means, dt), _c(sh_coeffs, dt), _c(v_colors, dt)
    cam, gid = _c(camera_ids, np.int64), _c(gaussian_ids, np.int64)
    M, K = cam.shape[0], sh.shape[1]
    v_sh = np.zeros_like(sh); v_means = np.zeros_like(means)
    _lib("splat", prec).orc_view_colors_bwd(C.c_int64(M), C.c_int64(K), C.c_int(sh_degree), _p(viewmats),
                                            _p(means), _p(sh), _p(cam), _p(gid), _p(v_colors), _p(v_sh),
                                            _p(v_means))
    return v_sh, v_means


def view_colors_bwd_bound(viewmats, means, sh_coeffs, camera_ids, gaussian_ids, sh_degree, a_colors):
    """absolute shadow of view_colors_bwd (fp64 build) -> (b_sh, b_means)"""
    dt = np.float64
    viewmats, means, sh, a_colors = _c(viewmats, dt), _c(means, dt), _c(sh_coeffs, dt), _c(a_colors, dt)
    cam, gid = _c(camera_ids, np.int64), _c(gaussian_ids, np.int64)
    M, K = cam.shape[0], sh.shape[1]
    b_sh = np.zeros_like(sh); b_means = np.zeros_like(means)
    _lib("splat", "f64").orc_view_colors_bwd_bound(C.c_int64(M), C.c_int64(K), C.c_int(sh_degree), _p(viewmats), _p(means), _p(sh), _p(cam),
                                                   _p(gid), _p(a_colors), _p(b_sh), _p(b_means))
    return b_sh, b_means


# ----------------------------------------------------------------------------------------------
# P3 tile binning
# ----------------------------------------------------------------------------------------------
def tile_encode(W, H, tile_size, means2d, radii, depths, camera_ids, n_cameras, prec="f32"):
    """Returns tiles_per_gauss int32[M], isect_ids int64[I], flatten_ids int32[I], isect_offsets int32[C,th,tw].
    `depths` is always reinterpreted from float32 bits (the key uses raw fp32 depth bits)."""
    dt = _dt(prec)
    m2d, radii = _c(means2d, dt), _c(radii, np.int32)
    cam = _c(camera_ids, np.int64)
    M = radii.shape[0]
    depth_bits = np.ascontiguousarray(depths, dtype=np.float32).view(np.uint32)
    tpg = np.zeros(max(M, 1), np.int32)
    lib = _lib("splat", prec)
    lib.orc_tile_count.restype = C.c_int64
    I = lib.orc_tile_count(C.c_int64(M), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d), _p(radii), _p(tpg))
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    ids = np.zeros(max(I, 1), np.int64); flat = np.zeros(max(I, 1), np.int32)
    offs = np.zeros((n_cameras, th, tw), np.int32)
    lib.orc_tile_encode(C.c_int64(M), C.c_int64(n_cameras), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d),
                        _p(radii), _p(depth_bits), _p(cam), C.c_int64(I), _p(ids), _p(flat), _p(offs))
    return tpg[:M].copy(), ids[:I].copy(), flat[:I].copy(), offs


# ----------------------------------------------------------------------------------------------
# P4 compositing
# ----------------------------------------------------------------------------------------------
def rasterize_2dgs_fwd(means2d, ray_transforms, colors, opacities, normals, W, H, tile_size, isect_offsets,
                       flatten_ids, backgrounds=None, masks=None, prec="f32"):
    dt = _dt(prec)
    m2d, rt, col, opa, nrm, bg = (_c(a, dt) for a in (means2d, ray_transforms, colors, opacities, normals,
                                                      backgrounds))
    offs, flat = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32)
    masks = _c(masks, np.uint8)
    Cn, M, I = offs.shape[0], opa.shape[0], flat.shape[0]
    rc = np.zeros((Cn, H, W, 3), dt); rd = np.zeros((Cn, H, W, 1), dt); ra = np.zeros((Cn, H, W, 1), dt)
    rn = np.zeros((Cn, H, W, 3), dt); rm = np.zeros((Cn, H, W, 1), dt)
    last = np.zeros((Cn, H, W), np.int32); med = np.zeros((Cn, H, W), np.int32)
    vis = np.zeros((max(M, 1), 1), dt)
    _lib("splat", prec).orc_rasterize_2dgs_fwd(
        C.c_int64(Cn), C.c_int64(M), C.c_int64(I), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d), _p(rt),
        _p(col), _p(opa), _p(nrm), _p(bg), _p(masks), _p(offs), _p(flat), _p(rc), _p(rd), _p(ra), _p(rn), _p(rm),
        _p(last), _p(med), _p(vis))
    return dict(render_colors=rc, render_depths=rd, render_alphas=ra, render_normals=rn, render_median=rm,
                last_ids=last, median_ids=med, visibilities=vis[:M])


def rasterize_2dgs_bwd(means2d, ray_transforms, colors, opacities, normals, W, H, tile_size, isect_offsets,
                       flatten_ids, render_alphas, last_ids, median_ids, v_render_colors, v_render_depths,
                       v_render_alphas, v_render_normals, v_render_median, backgrounds=None, masks=None,
                       absgrad=True, prec="f32", abs_sums=False):
    """Gradients are returned as float64 arrays irrespective of `prec`.  abs_sums=True adds `abs_ray_transforms` [M,3,3] and
    `abs_densify` [M,2]: the sum over pixels of |per-pixel contribution| to each element (the conditioning of the sums: an fp32
    accumulation of those terms cannot be more accurate than ~eps32 x this)."""
    dt = _dt(prec)
    m2d, rt, col, opa, nrm, bg = (_c(a, dt) for a in (means2d, ray_transforms, colors, opacities, normals,
                                                      backgrounds))
    offs, flat = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32)
    masks = _c(masks, np.uint8)
    ralpha, last, med = _c(render_alphas, dt), _c(last_ids, np.int32), _c(median_ids, np.int32)
    vc, vd, va, vn, vmed = (_c(a, dt) for a in (v_render_colors, v_render_depths, v_render_alphas,
                                                v_render_normals, v_render_median))
    Cn, M, I = offs.shape[0], opa.shape[0], flat.shape[0]
    Mz = max(M, 1)
    g = dict(v_means2d=np.zeros((Mz, 2)), v_ray_transforms=np.zeros((Mz, 3, 3)), v_colors=np.zeros((Mz, 3)),
             v_opacities=np.zeros(Mz), v_normals=np.zeros((Mz, 3)), v_densify=np.zeros((Mz, 2)),
             v_means2d_abs=np.zeros((Mz, 2)) if absgrad else None,
             abs_ray_transforms=np.zeros((Mz, 3, 3)) if abs_sums else None, abs_densify=np.zeros((Mz, 2)) if abs_sums else None)
    _lib("splat", prec).orc_rasterize_2dgs_bwd(
        C.c_int64(Cn), C.c_int64(M), C.c_int64(I), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d), _p(rt),
        _p(col), _p(opa), _p(nrm), _p(bg), _p(masks), _p(offs), _p(flat), _p(ralpha), _p(last), _p(med), _p(vc),
        _p(vd), _p(va), _p(vn), _p(vmed), _p(g["v_means2d"]), _p(g["v_ray_transforms"]), _p(g["v_colors"]),
        _p(g["v_opacities"]), _p(g["v_normals"]), _p(g["v_densify"]), _p(g["v_means2d_abs"]), _p(g["abs_ray_transforms"]),
        _p(g["abs_densify"]))
    return {k: (v[:M] if v is not None else None) for k, v in g.items() if v is not None or k == "v_means2d_abs"}


def rasterize_2dgs_fragility(means2d, ray_transforms, opacities, W, H, tile_size, isect_offsets, flatten_ids, masks=None,
                             kmargin=16.0, ulp_floor=2.4e-7, cond_abs=2e-6, kappa_max=8.0, prec="f64"):
    """Decision margins of the compositing operator (oracle/splat_oracle.c: orc_rasterize_2dgs_fragility).
    -> pix_flags uint8 [C,H,W], splat_flags uint8 [M], counts {pairs, valid}.  flag 0 = every decision the pixel (or any pixel
    that blends the splat) takes has a margin of more than `kmargin` x the fp32 evaluation error of the compared quantity and
    no blending weight is ill-conditioned (and, for a splat, it is nowhere blended edge-on: cancellation of z.z below
    `kappa_max`): the elements a 1e-4 element-wise comparison is meaningful on."""
    dt = _dt(prec)
    m2d, rt, opa = (_c(a, dt) for a in (means2d, ray_transforms, opacities))
    offs, flat = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32)
    masks = _c(masks, np.uint8)
    Cn, M, I = offs.shape[0], opa.shape[0], flat.shape[0]
    pf = np.zeros((Cn, H, W), np.uint8); sf = np.zeros(max(M, 1), np.uint8); counts = np.zeros(8, np.int64)
    _lib("splat", prec).orc_rasterize_2dgs_fragility(
        C.c_int64(Cn), C.c_int64(M), C.c_int64(I), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d), _p(rt), _p(opa),
        _p(masks), _p(offs), _p(flat), C.c_double(kmargin), C.c_double(ulp_floor), C.c_double(cond_abs), C.c_double(kappa_max),
        _p(pf), _p(sf), _p(counts))
    return pf, sf[:M], dict(pairs=int(counts[0]), valid=int(counts[1]))


# --- decision-matched evaluation (the parity gate of tests/util.py; see the block comment in splat_oracle.c) ----------------
ACC_DEPTH = 4.0
TRACE_BLEND, TRACE_BRANCH3D, TRACE_CLAMPED, TRACE_STOP, TRACE_MEDIAN = 1, 2, 4, 8, 16
FLIP_NAMES = ("alpha", "branch", "clamp", "termination", "median")
COND_SLICES = dict(v_means2d=slice(0, 2), v_ray_transforms=slice(2, 11), v_colors=slice(11, 14), v_opacities=slice(14, 15),
                   v_normals=slice(15, 18), v_densify=slice(18, 20), v_means2d_abs=slice(20, 22))
PIX_BOUND_COLS = dict(render_colors=0, render_depths=1, render_alphas=2, render_normals=3, render_median=4)


def trace_plan(pix_flags, isect_offsets, n_isects, tile_size=16):
    """trace_rows int32 [C,H,W] (row of the decision record, -1 = not traced) and the record stride (longest tile list among the
    traced pixels) for the pixels with a non-zero flag."""
    Cn, H, W = pix_flags.shape
    offs = np.asarray(isect_offsets, np.int64).reshape(-1)
    lens = np.diff(np.concatenate([offs, [n_isects]]))
    th, tw = (H + tile_size - 1) // tile_size, (W + tile_size - 1) // tile_size
    sel = np.flatnonzero(pix_flags.reshape(-1))
    rows = np.full(Cn * H * W, -1, np.int32)
    rows[sel] = np.arange(sel.size, dtype=np.int32)
    c, rem = np.divmod(sel, H * W)
    y, x = np.divmod(rem, W)
    tl = lens[(c * th + y // tile_size) * tw + x // tile_size] if sel.size else np.zeros(0, np.int64)
    stride = int(tl.max()) if sel.size else 1
    return rows.reshape(Cn, H, W), max(stride, 1), int(sel.size)


def rasterize_2dgs_trace(means2d, ray_transforms, opacities, W, H, tile_size, isect_offsets, flatten_ids, trace_rows, trace_stride,
                         masks=None, prec="f32"):
    """Decision record of this build's own evaluation for the traced pixels: uint8 [n_rows, trace_stride]."""
    dt = _dt(prec)
    m2d, rt, opa = (_c(a, dt) for a in (means2d, ray_transforms, opacities))
    offs, flat, rows = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32), _c(trace_rows, np.int32)
    masks = _c(masks, np.uint8)
    n_rows = int(rows.max()) + 1 if rows.size else 0
    bits = np.zeros((max(n_rows, 1), trace_stride), np.uint8)
    _lib("splat", prec).orc_rasterize_2dgs_trace(
        C.c_int64(offs.shape[0]), C.c_int64(opa.shape[0]), C.c_int64(flat.shape[0]), C.c_int(W), C.c_int(H), C.c_int(tile_size),
        _p(m2d), _p(rt), _p(opa), _p(masks), _p(offs), _p(flat), _p(rows), C.c_int64(trace_stride), _p(bits))
    return bits


def rasterize_2dgs_fwd_matched(means2d, ray_transforms, colors, opacities, normals, W, H, tile_size, isect_offsets, flatten_ids,
                               trace_rows=None, trace_bits=None, backgrounds=None, masks=None, ulp_floor=2.4e-7, prec="f64"):
    """Forward under the traced decisions (None: the build's own everywhere) + first-order fp32 error bounds in eps32 units
    (`pix_bound` [C,H,W,5] in PIX_BOUND_COLS order, `vis_bound` [M,1]) + the flips of the trace against the own decisions
    (`flips`: name -> (count, worst margin / fp32 evaluation error))."""
    dt = _dt(prec)
    m2d, rt, col, opa, nrm, bg = (_c(a, dt) for a in (means2d, ray_transforms, colors, opacities, normals, backgrounds))
    offs, flat = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32)
    masks = _c(masks, np.uint8)
    rows = _c(trace_rows, np.int32); bits = _c(trace_bits, np.uint8)
    stride = bits.shape[1] if bits is not None else 0
    Cn, M, I = offs.shape[0], opa.shape[0], flat.shape[0]
    rc = np.zeros((Cn, H, W, 3), dt); rd = np.zeros((Cn, H, W, 1), dt); ra = np.zeros((Cn, H, W, 1), dt)
    rn = np.zeros((Cn, H, W, 3), dt); rm = np.zeros((Cn, H, W, 1), dt)
    last = np.zeros((Cn, H, W), np.int32); med = np.zeros((Cn, H, W), np.int32)
    vis = np.zeros((max(M, 1), 1), dt)
    pb = np.zeros((Cn, H, W, 5)); vb = np.zeros((max(M, 1), 1)); fcnt = np.zeros(5, np.int64); fworst = np.zeros(5)
    _lib("splat", prec).orc_rasterize_2dgs_fwd_matched(
        C.c_int64(Cn), C.c_int64(M), C.c_int64(I), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d), _p(rt), _p(col), _p(opa),
        _p(nrm), _p(bg), _p(masks), _p(offs), _p(flat), _p(rows), C.c_int64(stride), _p(bits), C.c_double(ulp_floor), _p(rc), _p(rd),
        _p(ra), _p(rn), _p(rm), _p(last), _p(med), _p(vis), _p(pb), _p(vb), _p(fcnt), _p(fworst))
    return dict(render_colors=rc, render_depths=rd, render_alphas=ra, render_normals=rn, render_median=rm, last_ids=last,
                median_ids=med, visibilities=vis[:M], pix_bound=pb, vis_bound=vb[:M],
                flips={nm: (int(fcnt[i]), float(fworst[i])) for i, nm in enumerate(FLIP_NAMES)})


def rasterize_2dgs_bwd_matched(means2d, ray_transforms, colors, opacities, normals, W, H, tile_size, isect_offsets, flatten_ids,
                               render_alphas, last_ids, median_ids, v_render_colors, v_render_depths, v_render_alphas,
                               v_render_normals, v_render_median, trace_rows=None, trace_bits=None, backgrounds=None, masks=None,
                               prec="f64", recovers_final_T=False):
    """VJP under the traced decisions; float64 gradients + `cond` [M,22] (first-order fp32 error bound of every gradient element
    in eps32 units, COND_SLICES layout).  recovers_final_T: the implementation under test computes the final transmittance as
    1 - render_alphas (absolute error eps32, i.e. 1/T relative) instead of saving it; libgsdf_hip saves it."""
    dt = _dt(prec)
    m2d, rt, col, opa, nrm, bg = (_c(a, dt) for a in (means2d, ray_transforms, colors, opacities, normals, backgrounds))
    offs, flat = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32)
    masks = _c(masks, np.uint8)
    rows = _c(trace_rows, np.int32); bits = _c(trace_bits, np.uint8)
    stride = bits.shape[1] if bits is not None else 0
    ralpha, last, med = _c(render_alphas, dt), _c(last_ids, np.int32), _c(median_ids, np.int32)
    vc, vd, va, vn, vmed = (_c(a, dt) for a in (v_render_colors, v_render_depths, v_render_alphas, v_render_normals, v_render_median))
    Cn, M, I = offs.shape[0], opa.shape[0], flat.shape[0]
    Mz = max(M, 1)
    g = dict(v_means2d=np.zeros((Mz, 2)), v_ray_transforms=np.zeros((Mz, 3, 3)), v_colors=np.zeros((Mz, 3)), v_opacities=np.zeros(Mz),
             v_normals=np.zeros((Mz, 3)), v_densify=np.zeros((Mz, 2)), v_means2d_abs=np.zeros((Mz, 2)), cond=np.zeros((Mz, 44)))
    _lib("splat", prec).orc_rasterize_2dgs_bwd_matched(
        C.c_int64(Cn), C.c_int64(M), C.c_int64(I), C.c_int(W), C.c_int(H), C.c_int(tile_size), _p(m2d), _p(rt), _p(col), _p(opa),
        _p(nrm), _p(bg), _p(masks), _p(offs), _p(flat), _p(rows), C.c_int64(stride), _p(bits), _p(ralpha), _p(last), _p(med), _p(vc),
        _p(vd), _p(va), _p(vn), _p(vmed), _p(g["v_means2d"]), _p(g["v_ray_transforms"]), _p(g["v_colors"]), _p(g["v_opacities"]),
        _p(g["v_normals"]), _p(g["v_densify"]), _p(g["v_means2d_abs"]), _p(g["cond"]), C.c_double(1.0 if recovers_final_T else 0.0))
    out = {k: v[:M] for k, v in g.items()}
    # root-sum-square of the per-pixel evaluation errors + the fp32 accumulation of the terms (wave tree + a few atomics: ACC_DEPTH)
    out["cond"] = np.sqrt(out["cond"][:, :22]) + ACC_DEPTH * out["cond"][:, 22:]
    return out


def set_threads(n):
    """OpenMP thread count for the raster loops (cpu_baseline reports it as `cores`)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(n))
    except OSError:
        pass


# ----------------------------------------------------------------------------------------------
# S1 hash grid / S2 MLP / SDF head / K1 KNN   (oracle/sdf_oracle.c)
# ----------------------------------------------------------------------------------------------
GRID_DEFAULT = dict(n_levels=16, n_feat=2, log2_hashmap=19, base_res=32, per_level_scale=2.0)


def grid_offsets(cfg=GRID_DEFAULT):
    offs = np.zeros(cfg["n_levels"] + 1, np.int64)
    f = _lib("sdf", "f32").orc_grid_offsets
    f.restype = C.c_int64
    total = f(C.c_int(cfg["n_levels"]), C.c_int(cfg["n_feat"]), C.c_int(cfg["log2_hashmap"]), C.c_int(cfg["base_res"]),
              C.c_float(cfg["per_level_scale"]), _p(offs))
    return offs, int(total)


def _gargs(cfg):
    return (C.c_int(cfg["n_levels"]), C.c_int(cfg["n_feat"]), C.c_int(cfg["log2_hashmap"]), C.c_int(cfg["base_res"]),
            C.c_float(cfg["per_level_scale"]))


def grid_fwd(x, table, cfg=GRID_DEFAULT, want_jac=False, prec="f32"):
    dt = _dt(prec)
    x, table = _c(x, dt), _c(table, dt)
    offs, _ = grid_offsets(cfg)
    B, D = x.shape[0], cfg["n_levels"] * cfg["n_feat"]
    feat = np.zeros((B, D), dt)
    jac = np.zeros((B, D, 3), dt) if want_jac else None
    _lib("sdf", prec).orc_grid_fwd(C.c_int64(B), *_gargs(cfg), _p(offs), _p(x), _p(table), _p(feat), _p(jac))
    return (feat, jac) if want_jac else feat


def grid_bwd(x, table, v_feat, cfg=GRID_DEFAULT, prec="f32"):
    dt = _dt(prec)
    x, table, v_feat = _c(x, dt), _c(table, dt), _c(v_feat, dt)
    offs, _ = grid_offsets(cfg)
    v_table = np.zeros(table.shape, np.float64); v_x = np.zeros_like(x)
    _lib("sdf", prec).orc_grid_bwd(C.c_int64(x.shape[0]), *_gargs(cfg), _p(offs), _p(x), _p(table), _p(v_feat),
                                   _p(v_table), _p(v_x))
    return v_table, v_x


def grid_bwd_bwd(x, table, v_feat, vv_x, cfg=GRID_DEFAULT, prec="f32"):
    dt = _dt(prec)
    x, table, v_feat, vv_x = _c(x, dt), _c(table, dt), _c(v_feat, dt), _c(vv_x, dt)
    offs, _ = grid_offsets(cfg)
    g_vfeat = np.zeros_like(v_feat); g_table = np.zeros(table.shape, np.float64); g_x = np.zeros_like(x)
    _lib("sdf", prec).orc_grid_bwd_bwd(C.c_int64(x.shape[0]), *_gargs(cfg), _p(offs), _p(x), _p(table), _p(v_feat),
                                       _p(vv_x), _p(g_vfeat), _p(g_table), _p(g_x))
    return g_vfeat, g_table, g_x


def grid_bwd3(x, table, v_feat, vv_x, lam_x, mu=None, cfg=GRID_DEFAULT, prec="f32"):
    """third order: the backward of grid_bwd_bwd for lam_x (arriving at g_x) and mu (arriving at g_vfeat) -> (t_vfeat, t_table, t_vv, t_x)"""
    dt = _dt(prec)
    x, table, v_feat, vv_x, lam_x, mu = _c(x, dt), _c(table, dt), _c(v_feat, dt), _c(vv_x, dt), _c(lam_x, dt), _c(mu, dt)
    offs, _ = grid_offsets(cfg)
    t_vfeat = np.zeros_like(v_feat); t_table = np.zeros(table.shape, np.float64); t_vv = np.zeros_like(x); t_x = np.zeros_like(x)
    _lib("sdf", prec).orc_grid_bwd3(C.c_int64(x.shape[0]), *_gargs(cfg), _p(offs), _p(x), _p(table), _p(v_feat), _p(vv_x), _p(lam_x), _p(mu),
                                    _p(t_vfeat), _p(t_table), _p(t_vv), _p(t_x))
    return t_vfeat, t_table, t_vv, t_x


def mlp_fwd(x, dims, weights, biases=None, want_acts=False, prec="f32"):
    dt = _dt(prec)
    x, weights, biases = _c(x, dt), _c(weights, dt), _c(biases, dt)
    dims_a = np.asarray(dims, np.int32)
    B, nl = x.shape[0], len(dims) - 1
    out = np.zeros((B, dims[-1]), dt)
    acts = np.zeros((B, int(sum(dims[1:-1]))), dt) if want_acts else None
    _lib("sdf", prec).orc_mlp_fwd(C.c_int64(B), C.c_int(nl), _p(dims_a), _p(weights), _p(biases), _p(x), _p(out), _p(acts))
    return (out, acts) if want_acts else out


def mlp_bwd(x, dims, weights, biases, v_out, prec="f32"):
    dt = _dt(prec)
    x, weights, biases, v_out = _c(x, dt), _c(weights, dt), _c(biases, dt), _c(v_out, dt)
    dims_a = np.asarray(dims, np.int32)
    B, nl = x.shape[0], len(dims) - 1
    v_in = np.zeros_like(x); v_w = np.zeros(weights.shape, np.float64)
    v_b = np.zeros(biases.shape, np.float64) if biases is not None else None
    _lib("sdf", prec).orc_mlp_bwd(C.c_int64(B), C.c_int(nl), _p(dims_a), _p(weights), _p(biases), _p(x), _p(v_out),
                                  _p(v_in), _p(v_w), _p(v_b))
    return v_in, v_w, v_b


def mlp_bwd_bwd(x, dims, weights, biases, v_out, vv_in, prec="f32"):
    """Double backward of the decoder (oracle/sdf_oracle.c: orc_mlp_bwd_bwd): vv_in = dL/d(v_in) of the first backward ->
    (dL/d v_out [B, d_out], dL/d weights (float64))."""
    dt = _dt(prec)
    x, weights, biases, v_out, vv_in = _c(x, dt), _c(weights, dt), _c(biases, dt), _c(v_out, dt), _c(vv_in, dt)
    dims_a = np.asarray(dims, np.int32)
    B, nl = x.shape[0], len(dims) - 1
    g_vout = np.zeros((B, dims[-1]), dt); g_w = np.zeros(weights.shape, np.float64)
    _lib("sdf", prec).orc_mlp_bwd_bwd(C.c_int64(B), C.c_int(nl), _p(dims_a), _p(weights), _p(biases), _p(x), _p(v_out), _p(vv_in),
                                      _p(g_vout), _p(g_w))
    return g_vout, g_w


def sdf_head(out, inv_bce_sigma, prec="f32"):
    dt = _dt(prec)
    out = _c(out, dt)
    B = out.shape[0]
    sdf = np.zeros(B, dt); isig = np.zeros(B, dt)
    _lib("sdf", prec).orc_sdf_head(C.c_int64(B), _r(inv_bce_sigma, prec), _p(out), _p(sdf), _p(isig))
    return sdf, isig


def knn_mean_dist2(pts, prec="f32"):
    dt = _dt(prec)
    pts = _c(pts, dt)
    out = np.zeros(pts.shape[0], dt)
    _lib("sdf", prec).orc_knn_mean_dist2(C.c_int64(pts.shape[0]), _p(pts), _p(out))
    return out


# ----------------------------------------------------------------------------------------------
# A1 occupancy structure (oracle/occ_oracle.c; fp32 only)
# ----------------------------------------------------------------------------------------------
def _occ():
    if "occ" not in _LIBS:
        path = os.path.join(_BUILD, "liborc_occ.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.orc_occ_words.restype = C.c_int64
        lib.orc_occ_list.restype = C.c_int64
        _LIBS["occ"] = lib
    return _LIBS["occ"]


def occ_build(level, xyz, dilate27):
    lib = _occ()
    xyz = _c(xyz, np.float32).reshape(-1, 3)
    grid = np.zeros(lib.orc_occ_words(C.c_int(level)), np.uint32)
    lib.orc_occ_build(C.c_int(level), C.c_int64(xyz.shape[0]), _p(xyz), C.c_int(int(dilate27)), _p(grid))
    return grid


def occ_query(level, grid, xyz, query_level=-1):
    xyz = _c(xyz, np.float32).reshape(-1, 3)
    mask = np.zeros(xyz.shape[0], np.uint8)
    _occ().orc_occ_query(C.c_int(level), C.c_int(query_level), C.c_int64(xyz.shape[0]), _p(xyz), _p(grid), _p(mask))
    return mask


def occ_list(level, grid):
    lib = _occ()
    n = lib.orc_occ_list(C.c_int(level), _p(grid), None)
    out = np.zeros((n, 3), np.int16)
    lib.orc_occ_list(C.c_int(level), _p(grid), _p(out))
    return out


def occ_raymarch(level, grid, origins, dirs, num_samples=1):
    lib = _occ()
    o, d = _c(origins, np.float32).reshape(-1, 3), _c(dirs, np.float32).reshape(-1, 3)
    n = o.shape[0]
    counts = np.zeros(n, np.int32)
    lib.orc_occ_raymarch_count(C.c_int(level), C.c_int64(n), _p(o), _p(d), _p(grid), _p(counts))
    offs = np.concatenate([[0], np.cumsum(counts.astype(np.int64))[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
    S = int(counts.sum()) * num_samples
    ridx, samples, depth = np.zeros(S, np.int32), np.zeros((S, 3), np.float32), np.zeros((S, 1), np.float32)
    lib.orc_occ_raymarch_fill(C.c_int(level), C.c_int64(n), _p(o), _p(d), _p(grid), _p(offs), _p(counts), C.c_int(num_samples),
                              _p(ridx), _p(samples), _p(depth))
    return counts, ridx, samples, depth
