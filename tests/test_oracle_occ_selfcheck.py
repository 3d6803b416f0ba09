"""Pins oracle/occ_oracle.c (the occupancy structure's CPU restatement; parity unpinned, see its header) against an
independent brute-force numpy statement of the same semantics on small grids."""
import numpy as np
import pytest

from oracle import oracle as orc


def _quant(x, res):
    return np.clip(np.floor(res * (x.astype(np.float32) + np.float32(1.0)) / np.float32(2.0)), 0, res - 1).astype(np.int64)


def _bits(grid, L):
    """level-L occupancy as a dense bool [z,y,x] array from the packed pyramid"""
    off = sum(max(1, 8 ** k // 32) for k in range(L))
    words = grid[off: off + max(1, 8 ** L // 32)]
    b = ((words[:, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(-1)[: 8 ** L]
    return b.reshape(2 ** L, 2 ** L, 2 ** L)


@pytest.mark.parametrize("L,dilate", [(1, False), (3, True), (5, True), (6, False)])
def test_build_query_list_against_numpy(L, dilate):
    rng = np.random.default_rng(L)
    res = 2 ** L
    pts = (rng.random((300, 3)) * 2.4 - 1.2).astype(np.float32)        # some outside the cube: clamped by quantize
    grid = orc.occ_build(L, pts, dilate)
    want = np.zeros((res, res, res), bool)
    q = _quant(pts, res)
    r = 1 if dilate else 0
    for dz in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                c = np.clip(q + np.array([dx, dy, dz]), 0, res - 1)
                want[c[:, 2], c[:, 1], c[:, 0]] = True
    assert np.array_equal(_bits(grid, L), want)
    for l in range(L - 1, -1, -1):                                        # parent = OR of its 8 children
        f = 2 ** (L - l)
        coarse = want.reshape(2 ** l, f, 2 ** l, f, 2 ** l, f).any((1, 3, 5))
        assert np.array_equal(_bits(grid, l), coarse), l
    vox = orc.occ_list(L, grid)
    zyx = np.argwhere(want)
    assert np.array_equal(vox, zyx[:, ::-1].astype(np.int16))
    qp = (rng.random((500, 3)) * 2.6 - 1.3).astype(np.float32)
    inside = np.all((qp >= -1) & (qp <= 1), 1)
    for l in (L, max(L - 2, 0)):
        ql = _quant(qp, 2 ** l)
        f = 2 ** (L - l)
        occ = want.reshape(2 ** l, f, 2 ** l, f, 2 ** l, f).any((1, 3, 5))
        assert np.array_equal(orc.occ_query(L, grid, qp, l).astype(bool), inside & occ[ql[:, 2], ql[:, 1], ql[:, 0]])
    assert np.array_equal(orc.occ_query(L, grid, qp, -1), orc.occ_query(L, grid, qp, L))


@pytest.mark.parametrize("L,n_pts", [(4, 60), (6, 400), (7, 3000)])
def test_raymarch_against_brute_force_slabs(L, n_pts):
    rng = np.random.default_rng(10 + L)
    res = 2 ** L
    pts = (rng.random((n_pts, 3)) * 1.6 - 0.8).astype(np.float32)
    grid = orc.occ_build(L, pts, True)
    occ = _bits(grid, L)
    n_rays = 200
    o = (rng.random((n_rays, 3)) * 3.0 - 1.5).astype(np.float32)          # inside and outside the cube
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d[:20, 1] = 0.0                                                        # axis-parallel components
    d[:10, 2] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    counts, ridx, samples, depth = orc.occ_raymarch(L, grid, o, d, 2)
    assert counts.sum() > 50 and ridx.shape[0] == 2 * counts.sum()
    assert np.array_equal(ridx, np.repeat(np.arange(n_rays), 2 * counts).astype(np.int32))
    np.testing.assert_allclose(samples, o[ridx] + d[ridx] * depth, rtol=0, atol=1e-6)
    zyx = np.argwhere(occ).astype(np.float64)
    lo = zyx[:, ::-1] * (2.0 / res) - 1.0                                  # voxel min corners in [-1,1]
    hi = lo + 2.0 / res
    cell = 2.0 / res
    start = 0
    for r in range(n_rays):
        od, dd = o[r].astype(np.float64), d[r].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (lo - od) / dd, (hi - od) / dd
        par = dd == 0
        tin = np.where(par, -np.inf, np.minimum(t0, t1))
        tout = np.where(par, np.inf, np.maximum(t0, t1))
        miss = (par & ((od < lo) | (od >= hi))).any(1)
        tin, tout = np.maximum(tin.max(1), 0.0), tout.min(1)
        length = np.where(miss, -1.0, tout - tin)
        must = np.argsort(tin[length > 5e-3 * cell])                       # crossings longer than 5e-3 cells must be found
        want_in, want_out = tin[length > 5e-3 * cell][must], tout[length > 5e-3 * cell][must]
        got = depth[start: start + 2 * counts[r], 0].reshape(-1, 2).astype(np.float64)
        start += 2 * counts[r]
        # the oracle reports stratified midpoints t_in + (t_out - t_in) (k + 1/2) / 2 -> recover the interval
        g_in, g_out = got[:, 0] - (got[:, 1] - got[:, 0]) / 2, got[:, 1] + (got[:, 1] - got[:, 0]) / 2
        assert np.all(np.diff(g_in) > 0), r                                # front to back
        # every reported interval is a genuine crossing of an occupied voxel ...
        allowed_in, allowed_out = tin[length > 0], tout[length > 0]
        for a, b in zip(g_in, g_out):
            k = np.argmin(np.abs(allowed_in - a) + np.abs(allowed_out - b))
            assert abs(allowed_in[k] - a) < 2e-4 * cell * res and abs(allowed_out[k] - b) < 2e-4 * cell * res, (r, a, b)
        # ... and none of the substantial ones is missing
        for a, b in zip(want_in, want_out):
            assert np.min(np.abs(g_in - a) + np.abs(g_out - b)) < 4e-4 * cell * res if len(g_in) else False, (r, a, b)
