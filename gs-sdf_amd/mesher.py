"""Host-side mirror of the reference's in-tree mesher interface on top of the C ABI (include/gsdf_hip.h, section M1):
`mc::marching_cubes(density_grid, thresh, lower, upper) -> [vertices, faces]` and `mc::save_mesh_as_ply`
(/root/reference/include/mesher/cumcubes/include/cumcubes.hpp:10-21, src/cumcubes.cpp)."""
import ctypes as C

import torch

from . import capi
from .capi import f32, ptr


TABLES = {"reference": 0, "watertight": 1}


def marching_cubes(density_grid, thresh, lower, upper, table="reference"):
    """density_grid [X,Y,Z] float32 on the device -> [vertices [V,3] float32, faces [F,3] int32].  Vertices are ordered by
    owning cell then axis, faces by cell then table order (deterministic; the reference's order comes from atomics).
    table = "reference" (default): the reference's in-tree triangle table -> the reference's triangles in every cell;
    "watertight": the derived table (same vertices, no cracks across ambiguous faces)."""
    if density_grid.dim() != 3:
        raise RuntimeError("marching_cubes: expected a [X,Y,Z] grid")
    if table not in TABLES:
        raise RuntimeError(f"marching_cubes: table must be one of {sorted(TABLES)}")
    tb = TABLES[table]
    L = capi.lib()
    g = density_grid.contiguous().float()
    rx, ry, rz = g.shape
    n = g.numel()
    dev = g.device
    n_vert = torch.empty(n, dtype=torch.int32, device=dev)
    n_tri = torch.empty(n, dtype=torch.int32, device=dev)
    capi.check(L.gsdf_mc_count(rx, ry, rz, tb, f32(g), float(thresh), ptr(n_vert), ptr(n_tri), capi.stream()), "mc_count")
    v_incl, t_incl = torch.cumsum(n_vert, 0, dtype=torch.int64), torch.cumsum(n_tri, 0, dtype=torch.int64)
    V, F = (int(v) for v in torch.stack([v_incl[-1], t_incl[-1]]).tolist())        # the one host sync
    vertices = torch.empty(V, 3, dtype=torch.float32, device=dev)
    faces = torch.empty(F, 3, dtype=torch.int32, device=dev)
    if V:
        lo, up = (C.c_float * 3)(*[float(v) for v in lower]), (C.c_float * 3)(*[float(v) for v in upper])
        v_off, t_off = (v_incl - n_vert).contiguous(), (t_incl - n_tri).contiguous()    # exclusive scans (kept alive here)
        capi.check(L.gsdf_mc_emit(rx, ry, rz, tb, f32(g), float(thresh), ptr(v_off), ptr(t_off), lo, up, f32(vertices),
                                  ptr(faces), capi.stream()), "mc_emit")
    return [vertices, faces]


def save_mesh_as_ply(path, vertices, faces, colors=None):
    """The file mc::save_mesh_as_ply writes (cumcubes.cpp:29-79): binary little-endian PLY, vertices float x y z + uchar
    red green blue (white when `colors` is None), faces as `property list int int vertex_index` (count 3 as int32)."""
    import numpy as np
    v = vertices.detach().cpu().float().numpy().astype("<f4")
    f = faces.detach().cpu().numpy().astype("<i4")
    c = np.full((v.shape[0], 3), 255, np.uint8) if colors is None else colors.detach().cpu().numpy().astype(np.uint8)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {v.shape[0]}", "property float x", "property float y",
              "property float z", "property uchar red", "property uchar green", "property uchar blue",
              f"element face {f.shape[0]}", "property list int int vertex_index", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        rec = np.zeros(v.shape[0], dtype=[("p", "<f4", 3), ("c", "u1", 3)])
        rec["p"], rec["c"] = v, c
        fh.write(rec.tobytes())
        fh.write(np.concatenate([np.full((f.shape[0], 1), 3, "<i4"), f], 1).astype("<i4").tobytes())
