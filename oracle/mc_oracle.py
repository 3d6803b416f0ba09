"""CPU restatement (numpy) of marching cubes as the reference's in-tree mesher defines it — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/include/mesher/cumcubes/src/cumcubes_kernel.cu: inside <=> value > thresh (:24,:52-59); one vertex
per straddling grid edge, owned by the edge's lower cell, at index + (thresh - d0)/(d1 - d0) (:97-139); corner bits and
the 12 edge -> (owner cell, axis) pairs (:169-193); world mapping vertices * (upper - lower)/res + lower (:260-274).
One thing is NOT the reference's: the ordering of vertices and faces (the reference's comes out of global atomics and is
not reproducible; here: by owning cell, then axis / table order).  Triangle table: `table="reference"` is the reference's
own in-tree table (include/mesher/cumcubes/include/utils.cuh:31-289), held as the golden vector
tests/golden/mc_triangle_table_reference.npy (written by tools/gen_mc_table_ref.py from the header where it lies): with it
the face SET of every cell is the reference's — the one place on this path where parity is PINNED by reference-held data.
`table="watertight"` regenerates the derived table of tools/gen_mc_table.py from first principles.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from gen_mc_table import triangulate     # noqa: E402  (regenerates the table from first principles, not from mc_table.h)

EDGE_OWNER = [(0, 0, 0, 0), (1, 0, 0, 1), (0, 1, 0, 0), (0, 0, 0, 1), (0, 0, 1, 0), (1, 0, 1, 1),
              (0, 1, 1, 0), (0, 0, 1, 1), (0, 0, 0, 2), (1, 0, 0, 2), (1, 1, 0, 2), (0, 1, 0, 2)]
_TABLES = {}
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mc_triangle_table_reference.npy")


def table(kind="reference"):
    if kind not in _TABLES:
        if kind == "watertight":
            _TABLES[kind] = [triangulate(m) for m in range(256)]
        elif kind == "reference":
            ref = np.load(GOLDEN)
            _TABLES[kind] = [[tuple(int(e) for e in ref[m][k:k + 3]) for k in range(0, 15, 3) if ref[m][k] >= 0] for m in range(256)]
        else:
            raise ValueError(kind)
    return _TABLES[kind]


def marching_cubes(grid, thresh, lower, upper, table_kind="reference"):
    g = np.ascontiguousarray(grid, np.float32)
    X, Y, Z = g.shape
    thresh = np.float32(thresh)
    inside = g > thresh
    cross = np.zeros((X, Y, Z, 3), bool)
    cross[:-1, :, :, 0] = inside[:-1] != inside[1:]
    cross[:, :-1, :, 1] = inside[:, :-1] != inside[:, 1:]
    cross[:, :, :-1, 2] = inside[:, :, :-1] != inside[:, :, 1:]
    n_vert = cross.sum(-1).reshape(-1)
    v_off = np.concatenate([[0], np.cumsum(n_vert)[:-1]]).reshape(X, Y, Z)
    # vertices: cell-major, then axis
    cells, axes = np.nonzero(cross.reshape(-1, 3))
    x, y, z = np.unravel_index(cells, (X, Y, Z))
    d0 = g[x, y, z]
    nb = np.stack([x, y, z], 1)
    nb[np.arange(len(axes)), axes] += 1
    d1 = g[nb[:, 0], nb[:, 1], nb[:, 2]]
    dt = (thresh - d0) / (d1 - d0)
    pos = np.stack([x, y, z], 1).astype(np.float32)
    pos[np.arange(len(axes)), axes] = pos[np.arange(len(axes)), axes] + dt
    scale = ((np.asarray(upper, np.float32) - np.asarray(lower, np.float32)) / np.array([X, Y, Z], np.float32)).astype(np.float32)
    vertices = (pos * scale).astype(np.float32) + np.asarray(lower, np.float32)
    # faces: cell-major, then table order
    if min(X, Y, Z) < 2:
        return vertices.astype(np.float32), np.zeros((0, 3), np.int32)
    i = inside.astype(np.int32)
    mask = (i[:-1, :-1, :-1] | i[1:, :-1, :-1] << 1 | i[1:, 1:, :-1] << 2 | i[:-1, 1:, :-1] << 3 | i[:-1, :-1, 1:] << 4
            | i[1:, :-1, 1:] << 5 | i[1:, 1:, 1:] << 6 | i[:-1, 1:, 1:] << 7)
    rank = np.zeros((X, Y, Z, 3), np.int64)
    rank[..., 1] = cross[..., 0]
    rank[..., 2] = cross[..., 0].astype(np.int64) + cross[..., 1]
    out_cells, out_k, out_faces = [], [], []
    T = table(table_kind)
    for m in np.unique(mask):
        tris = T[int(m)]
        if not tris:
            continue
        cx, cy, cz = np.nonzero(mask == m)
        lin = (cx * Y + cy) * Z + cz
        for k, tri in enumerate(tris):
            ids = []
            for e in tri:
                ox, oy, oz, a = EDGE_OWNER[e]
                ids.append(v_off[cx + ox, cy + oy, cz + oz] + rank[cx + ox, cy + oy, cz + oz, a])
            out_cells.append(lin); out_k.append(np.full(len(lin), k)); out_faces.append(np.stack(ids, 1))
    if not out_faces:
        return vertices.astype(np.float32), np.zeros((0, 3), np.int32)
    cells_all, k_all, faces = np.concatenate(out_cells), np.concatenate(out_k), np.concatenate(out_faces)
    order = np.lexsort((k_all, cells_all))
    return vertices.astype(np.float32), faces[order].astype(np.int32)
