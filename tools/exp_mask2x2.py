"""Round 6: what the 2x2 reach mask (csrc/reach_mask.h: reach_params + reach_mask2x2, host build) gives the quad-list compositing kernels, on a
cfg3-like scene scaled to 480x272 (RM_WH=W,H for another size, RM_MARGIN for another safety margin) against an fp64 brute force of the alpha test (CPU only; termination not modelled): conservativeness, useful
lanes of the evaluated ones, wave iterations (max over a wave's 16 quad lists); round 5's row lists on its 4x4 mask: 0.32 useful, 158 856 iterations on the same scene.  python tools/exp_mask2x2.py"""
import sys, os, ctypes as C, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.synth as synth
from oracle import oracle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = '/tmp/libreach_mask_host.so'
MARGIN = os.environ.get("RM_MARGIN")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", ROOT + "/gs-sdf_amd/csrc", ROOT + "/tests/cpp/reach_mask_host.cpp", "-o", out]
                      + ([f"-DGSDF_RM_MARGIN={MARGIN}f"] if MARGIN else []))
lib = C.CDLL(out); lib.reach_masks2x2.restype = None
W, H = (int(v) for v in os.environ.get("RM_WH", "480,272").split(",")); N = int(1_000_000 * (W * H) / (1920 * 1080))
sc = synth.make_scene(N, W, H, 0, seed=0)
n = lambda t: t.detach().numpy()
means, quats = n(sc["means"]), n(sc["quats"]); scales = np.exp(n(sc["log_scales"])); opac_all = 1 / (1 + np.exp(-n(sc["logit_opacities"])))
vm = n(synth.make_views(1)); K = n(sc["K"])
pr = oracle.projection_2dgs_fwd(means, quats, scales, vm, K, W, H)
tpg, ids, flat, offs = oracle.tile_encode(W, H, 16, pr["means2d"], pr["radii"], pr["depths"], pr["camera_ids"], 1)
tw = (W + 15) // 16; th = (H + 15) // 16
offs = np.asarray(offs).reshape(-1); T = tw * th
tile_of = np.repeat(np.arange(T), np.diff(np.append(offs, len(flat))))
idx = np.asarray(flat)
P = len(idx); print('pairs', P, 'M', len(pr["gaussian_ids"]), 'L', P / T)
rt = pr["ray_transforms"].astype(np.float32); m2d = pr["means2d"].astype(np.float32); opac = opac_all[pr["gaussian_ids"]].astype(np.float32)
txy = np.stack([(tile_of % tw) * 16, (tile_of // tw) * 16], 1).astype(np.float32)
m2 = np.zeros(P, np.uint64)
a = [np.ascontiguousarray(v) for v in (rt[idx].reshape(P, 9), m2d[idx], opac[idx], txy)]
lib.reach_masks2x2(C.c_int64(P), *(v.ctypes.data_as(C.c_void_p) for v in a), m2.ctypes.data_as(C.c_void_p))
px = np.arange(16)[None, :] + 0.5
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing='ij')
tot_keep = tot_set2 = tot_exact2 = lost = 0
quad_len = np.zeros((T, 64), np.int64); quad_len_exact = np.zeros((T, 64), np.int64)
# quad id inside a wave: wave w = 8x8 quadrant, quad = 2x2 block of it (row-major)
by, bx = np.meshgrid(np.arange(8), np.arange(8), indexing='ij')
qid = (16 * (2 * (by >> 2) + (bx >> 2)) + 4 * (by & 3) + (bx & 3)).reshape(-1)
for s0 in range(0, P, 20000):
    sl = slice(s0, min(P, s0 + 20000)); ii = idx[sl]; t = txy[sl]
    X = (t[:, 0:1].astype(np.float64) + px)[:, None, :].repeat(16, 1); Y = (t[:, 1:2].astype(np.float64) + px)[:, :, None].repeat(16, 2)
    Mm = rt[ii].astype(np.float64)
    hu = X[..., None] * Mm[:, None, None, 2, :] - Mm[:, None, None, 0, :]; hv = Y[..., None] * Mm[:, None, None, 2, :] - Mm[:, None, None, 1, :]
    z = np.cross(hu, hv)
    with np.errstate(divide='ignore', invalid='ignore'):
        s = z[..., :2] / z[..., 2:3]; g3 = (s ** 2).sum(-1)
    g2 = 2.0 * ((X - m2d[ii, 0:1, None]) ** 2 + (Y - m2d[ii, 1:2, None]) ** 2)
    sig = 0.5 * np.where(np.isfinite(g3), np.minimum(g3, g2), g2)
    alpha = np.minimum(0.999, opac[ii].astype(np.float64)[:, None, None] * np.exp(-sig))
    keep = (z[..., 2] != 0) & (alpha >= 1 / 255) & (X < W) & (Y < H)
    k22 = keep.reshape(len(ii), 8, 2, 8, 2).any(axis=(2, 4)).reshape(len(ii), 64)
    set2 = ((m2[sl][:, None] >> np.arange(64, dtype=np.uint64)[None]) & np.uint64(1)).astype(bool)     # bit 8 by + bx
    lost += int((k22 & ~set2).sum())
    tot_keep += keep.sum(); tot_set2 += set2.sum(); tot_exact2 += k22.sum()
    tl = tile_of[sl]
    np.add.at(quad_len, (tl[:, None].repeat(64, 1), qid[None].repeat(len(ii), 0)), set2.astype(np.int64))
    np.add.at(quad_len_exact, (tl[:, None].repeat(64, 1), qid[None].repeat(len(ii), 0)), k22.astype(np.int64))
print('kept pixels', tot_keep, '| blocks with a kept pixel the 2x2 mask drops:', lost)
print('2x2 mask: bits', tot_set2, 'useful lanes', tot_keep / (4 * tot_set2), '| exact 2x2:', tot_exact2, tot_keep / (4 * tot_exact2))
quads = quad_len.reshape(T, 4, 16).max(axis=2).sum(); quads_exact = quad_len_exact.reshape(T, 4, 16).max(axis=2).sum()
print('wave iterations: quad lists, this 2x2 mask', quads, '| quad lists, exact 2x2', quads_exact)
print('mean list length per quad', quad_len.mean(), ' sum of quad entries per tile', quad_len.sum() / T)
