"""The 2x2 reach mask of the compositing kernels (gs-sdf_amd/csrc/reach_mask.h: reach_params once per splat + reach_mask2x2 once per (tile,
splat) pair: exact pixel-row intervals of the alpha >= 1/255 ellipse and of the low-pass disk, grown by a safety margin) decides which lane
quads evaluate a (tile, splat) pair at all — in the forward AND the backward — so it must never clear the bit of a 2x2 block that holds a pixel
the alpha test would keep.  The GPU parity tests see that through the decision-matched gate; here the product's own source is
compiled for the host and checked directly, on splats made to stress it: elongated (aspect up to 300), oblique, grazing, tiny, huge, near the
screen border, every opacity.  Brute force in fp64 with the reference's alpha test (z = (x M_w - M_u) x (y M_w - M_v), s = z.xy / z.z,
sigma = min(|s|^2, 2 |p - mean2d|^2) / 2, alpha = min(0.999, o e^-sigma) >= 1/255)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402


@pytest.fixture(scope="module")
def mask_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("reach_mask") / "libreach_mask_host.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "gs-sdf_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "reach_mask_host.cpp"), "-o", str(out)])
    lib = C.CDLL(str(out))
    lib.reach_masks2x2.restype = None
    return lib


def _scene(rng, n, W, H, kind):
    means = np.stack([rng.uniform(-3.2, 3.2, n), rng.uniform(-2.0, 2.0, n), rng.uniform(0.4, 9.0, n)], 1).astype(np.float32)
    q = rng.normal(size=(n, 4)).astype(np.float32)
    if kind == "elongated":
        s0 = np.exp(rng.uniform(np.log(2e-3), np.log(2.0), n))
        scales = np.stack([s0, s0 / np.exp(rng.uniform(0.0, np.log(300.0), n)), np.full(n, 1e-3)], 1)
    elif kind == "tiny":
        scales = np.exp(rng.uniform(np.log(1e-4), np.log(5e-3), (n, 3)))
    else:
        scales = np.exp(rng.uniform(np.log(3e-3), np.log(0.6), (n, 3)))
    if kind == "grazing":       # discs seen almost edge-on: the normal nearly perpendicular to the view ray
        d = means / np.linalg.norm(means, axis=1, keepdims=True)
        t = np.cross(d, rng.normal(size=(n, 3)))
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        nrm = t + rng.uniform(-0.03, 0.03, (n, 1)) * d            # third axis of the rotation = the disc normal
        a = np.cross(nrm, d); a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = np.cross(nrm, a)
        R = np.stack([a, b, nrm / np.linalg.norm(nrm, axis=1, keepdims=True)], 2)
        w = 0.5 * np.sqrt(np.maximum(1 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2], 1e-6))
        q = np.stack([w, (R[:, 2, 1] - R[:, 1, 2]) / (4 * w), (R[:, 0, 2] - R[:, 2, 0]) / (4 * w), (R[:, 1, 0] - R[:, 0, 1]) / (4 * w)], 1)
    viewmat = np.eye(4, dtype=np.float32)[None]
    K = np.array([[[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]]], np.float32)
    opac = np.where(rng.uniform(size=n) < 0.2, rng.uniform(0.004, 0.02, n), rng.uniform(0.02, 1.0, n)).astype(np.float32)
    return means, q.astype(np.float32), scales.astype(np.float32), viewmat, K, opac


@pytest.mark.parametrize("kind", ["generic", "elongated", "grazing", "tiny"])
def test_no_kept_pixel_outside_the_mask(mask_lib, kind):
    W, H, n = 208, 144, 6000
    rng = np.random.default_rng({"generic": 1, "elongated": 2, "grazing": 3, "tiny": 4}[kind])
    means, q, scales, viewmat, K, opac_all = _scene(rng, n, W, H, kind)
    pr = oracle.projection_2dgs_fwd(means, q, scales, viewmat, K, W, H)
    M = len(pr["gaussian_ids"])
    assert M > n // 4, M
    rt, m2d, radii = pr["ray_transforms"].astype(np.float32), pr["means2d"].astype(np.float32), pr["radii"]
    opac = opac_all[pr["gaussian_ids"]]
    # every (splat, tile) pair of the splat's tile rectangle (the binning's rectangle: radius box clipped to the image)
    pairs = []
    for i in range(M):
        rx = ry = int(radii[i]) if radii.ndim == 1 else None
        if radii.ndim == 2:
            rx, ry = int(radii[i, 0]), int(radii[i, 1])
        x0, x1 = int(np.floor((m2d[i, 0] - rx) / 16)), int(np.ceil((m2d[i, 0] + rx) / 16))
        y0, y1 = int(np.floor((m2d[i, 1] - ry) / 16)), int(np.ceil((m2d[i, 1] + ry) / 16))
        for ty in range(max(y0, 0), min(y1, (H + 15) // 16)):
            for tx in range(max(x0, 0), min(x1, (W + 15) // 16)):
                pairs.append((i, tx, ty))
    pairs = np.array(pairs, np.int64)
    P = len(pairs)
    assert P > 2 * M // 3
    idx = pairs[:, 0]
    txy = (pairs[:, 1:3] * 16).astype(np.float32)
    masks = np.zeros(P, np.uint64)
    a = [np.ascontiguousarray(v) for v in (rt[idx].reshape(P, 9), m2d[idx], opac[idx], txy)]
    mask_lib.reach_masks2x2(C.c_int64(P), *(v.ctypes.data_as(C.c_void_p) for v in a), masks.ctypes.data_as(C.c_void_p))
    # brute force, fp64: the 256 pixel centres of every pair
    px = np.arange(16)[None, :] + 0.5
    X = (txy[:, 0:1].astype(np.float64) + px)[:, None, :].repeat(16, 1)          # [P, y, x]
    Y = (txy[:, 1:2].astype(np.float64) + px)[:, :, None].repeat(16, 2)
    Mm = rt[idx].astype(np.float64)
    hu = X[..., None] * Mm[:, None, None, 2, :] - Mm[:, None, None, 0, :]
    hv = Y[..., None] * Mm[:, None, None, 2, :] - Mm[:, None, None, 1, :]
    z = np.cross(hu, hv)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = z[..., :2] / z[..., 2:3]
        g3 = (s ** 2).sum(-1)
    g2 = 2.0 * ((X - m2d[idx, 0:1, None].astype(np.float64)) ** 2 + (Y - m2d[idx, 1:2, None].astype(np.float64)) ** 2)
    sigma = 0.5 * np.where(np.isfinite(g3), np.minimum(g3, g2), g2)
    alpha = np.minimum(0.999, opac[idx].astype(np.float64)[:, None, None] * np.exp(-sigma))
    keep = (z[..., 2] != 0) & (alpha >= 1.0 / 255.0) & (X < W) & (Y < H)
    # block bit of a pixel: 8 (y >> 1) + (x >> 1)
    yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    bit = (8 * (yy >> 1) + (xx >> 1)).astype(np.uint64)
    allowed = ((masks[:, None, None] >> bit[None]) & np.uint64(1)).astype(bool)
    lost = keep & ~allowed
    assert not lost.any(), (f"{kind}: {int(lost.sum())} kept pixels of {int(lost.any(axis=(1, 2)).sum())} pairs lie in blocks the mask drops; first: "
                            f"pair {pairs[np.argmax(lost.any(axis=(1, 2)))]}")
    # and the mask is worth having: how many of the set bits hold a kept pixel
    blocks_kept = keep.reshape(P, 8, 2, 8, 2).any(axis=(2, 4)).reshape(P, 64)
    set_bits = ((masks[:, None] >> np.arange(64, dtype=np.uint64)[None]) & np.uint64(1)).astype(bool)
    tight = blocks_kept.sum() / max(set_bits.sum(), 1)
    print(f"{kind}: {P} pairs, {int(keep.sum())} kept pixels, {int(set_bits.sum())} blocks in the masks, {tight:.2f} of them hold a kept pixel")
    assert tight > (0.1 if kind == "grazing" else 0.3)      # edge-on discs: unbounded conics keep the full mask
