"""oracle/mc_oracle.py (marching cubes restatement) and the derived triangle table (tools/gen_mc_table.py) checked
against first principles: the vertex set is exactly the set of straddling grid edges, every vertex lies on the iso-surface
of the trilinear edge interpolant, the surface is a closed 2-manifold where it should be (every edge shared by exactly
two triangles with opposite directions), Euler characteristic / area / signed volume of a sphere."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gen_mc_table as gen
from oracle import mc_oracle as mco


def test_table_invariants():
    for m in range(256):
        tris = gen.triangulate(m)
        inside = [(m >> c) & 1 for c in range(8)]
        crossing = {i for i, (a, b) in enumerate(gen.EDGE) if inside[a] != inside[b]}
        assert {e for t in tris for e in t} == crossing and len(tris) <= 5
        # every directed edge of the cube-local patch appears at most once; interior edges pair up with their reverse
        d = {}
        for a, b, c in tris:
            for e in ((a, b), (b, c), (c, a)):
                assert e not in d
                d[e] = 1
    # the committed header is what the generator produces (one packed 64-bit word per mask, nibble 0xF = end)
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gs-sdf_amd", "csrc", "mc_table.h")).read()
    words = [int(w, 16) for w in re.findall(r"0x([0-9A-F]{16})ull", hdr)]
    assert len(words) == 256
    for m, w in enumerate(words):
        flat = [e for t in gen.triangulate(m) for e in t]
        nib = [(w >> (4 * k)) & 0xF for k in range(16)]
        assert nib == flat + [15] * (16 - len(flat))


def test_reference_table_golden_vector_and_packed_header():
    """The golden vector (the reference's triangle_table, tools/gen_mc_table_ref.py) is a valid marching-cubes table over the
    reference's corner / edge numbering: every configuration uses exactly its crossing edges, at most 5 triangles, complement
    configurations use the same edges; and gs-sdf_amd/csrc/mc_table_ref.h holds exactly this table (nibble packing)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = np.load(os.path.join(root, "tests", "golden", "mc_triangle_table_reference.npy"))
    assert ref.shape == (256, 16) and ref.dtype == np.int8 and (ref[:, 15] == -1).all()
    for m in range(256):
        row = ref[m]
        n = int((row >= 0).sum())
        assert n % 3 == 0 and n <= 15 and (row[n:] == -1).all()
        inside = [(m >> c) & 1 for c in range(8)]
        crossing = {i for i, (a, b) in enumerate(gen.EDGE) if inside[a] != inside[b]}
        assert set(int(e) for e in row[:n]) == crossing, m
        assert set(int(e) for e in ref[255 - m] if e >= 0) == crossing
    hdr = open(os.path.join(root, "gs-sdf_amd", "csrc", "mc_table_ref.h")).read()
    words = [int(w, 16) for w in re.findall(r"0x([0-9A-F]{16})ull", hdr)]
    assert len(words) == 256
    for m, w in enumerate(words):
        nib = [(w >> (4 * k)) & 0xF for k in range(16)]
        assert nib == [15 if e < 0 else int(e) for e in ref[m]]
    if os.path.exists("/root/reference/include/mesher/cumcubes/include/utils.cuh"):     # build container: re-derive from the source
        sys.path.insert(0, os.path.join(root, "tools"))
        import gen_mc_table_ref as gr
        assert np.array_equal(gr.parse("/root/reference/include/mesher/cumcubes/include/utils.cuh"), ref)


def test_reference_table_mesh_of_a_sphere_is_closed():
    """With the reference's table the oracle's sphere (no ambiguous faces on a smooth surface) is a closed manifold too."""
    res, r = 24, 0.6
    ax = np.linspace(-1.0, 1.0, res, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    grid = (r - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)
    v, f = mco.marching_cubes(grid, 0.0, [-1.0] * 3, [1.0 + 2.0 / (res - 1)] * 3, "reference")
    v2, f2 = mco.marching_cubes(grid, 0.0, [-1.0] * 3, [1.0 + 2.0 / (res - 1)] * 3, "watertight")
    assert np.array_equal(v, v2) and len(f) == len(f2)                          # same vertices; same triangle count here
    assert _edges_manifold(f).all() and len(v) - 3 * len(f) // 2 + len(f) == 2


def _edges_manifold(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    key = e[:, 0].astype(np.int64) * (faces.max() + 1) + e[:, 1]
    rev = e[:, 1].astype(np.int64) * (faces.max() + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key)                       # no directed edge twice: consistent orientation
    return np.isin(key, rev)                                     # which directed edges have their partner


@pytest.mark.parametrize("res,thresh", [(24, 0.0), (33, 0.15)])
def test_sphere_is_a_closed_oriented_manifold(res, thresh):
    ax = np.linspace(-1.0, 1.0, res, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = 0.6
    grid = (r - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)            # > thresh inside the ball
    lower, upper = [-1.0] * 3, [1.0 + 2.0 / (res - 1)] * 3                    # so that v * (upper-lower)/res + lower = grid coords
    v, f = mco.marching_cubes(grid, thresh, lower, upper, "watertight")
    assert len(f) > 100 and f.min() == 0 and f.max() == len(v) - 1
    assert _edges_manifold(f).all()                                           # closed
    n_edges = 3 * len(f) // 2
    assert len(v) - n_edges + len(f) == 2                                     # Euler characteristic of a sphere
    rr = r - thresh
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), rr, atol=0.6 * (2.0 / (res - 1)) ** 2 / rr + 1e-3)
    a, b, c = v[f[:, 0]].astype(np.float64), v[f[:, 1]].astype(np.float64), v[f[:, 2]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    vol = (a * np.cross(b, c)).sum() / 6.0
    assert abs(area / (4 * np.pi * rr * rr) - 1) < 0.03
    assert abs(vol / (4 / 3 * np.pi * rr ** 3) - 1) < 0.03                    # positive: normals point out of the inside region


def test_random_field_vertices_and_watertightness():
    rng = np.random.default_rng(0)
    g = rng.standard_normal((13, 11, 9)).astype(np.float32)                   # every ambiguous configuration occurs
    v, f = mco.marching_cubes(g, 0.1, [0, 0, 0], [13, 11, 9], "watertight")   # identity mapping to index space
    inside = g > 0.1
    n_cross = (inside[:-1] != inside[1:]).sum() + (inside[:, :-1] != inside[:, 1:]).sum() + (inside[:, :, :-1] != inside[:, :, 1:]).sum()
    assert len(v) == n_cross
    frac = v - np.floor(v)
    assert ((frac > 0).sum(1) <= 1).all()                                     # each vertex moves along ONE axis from a grid node
    # value of the linear interpolant along the owning edge equals the threshold
    base = np.floor(v).astype(int)
    axis = np.argmax(frac, 1)
    nb = base.copy(); nb[np.arange(len(v)), axis] += 1
    nb = np.minimum(nb, np.array(g.shape) - 1)
    t = frac[np.arange(len(v)), axis]
    val = g[base[:, 0], base[:, 1], base[:, 2]] * (1 - t) + g[nb[:, 0], nb[:, 1], nb[:, 2]] * t
    ok = t > 0
    np.testing.assert_allclose(val[ok], 0.1, atol=2e-5)
    # watertight: a directed edge lacks its partner only on the boundary of the volume
    has = _edges_manifold(f)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])[~has]
    pts = v[e.reshape(-1)]
    on_boundary = ((pts < 1e-6) | (pts > np.array(g.shape) - 1 - 1e-6)).any(1)
    assert on_boundary.all()
