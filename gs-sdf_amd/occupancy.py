"""Host-side mirror of the occupancy acceleration structure the reference takes from its (absent) kaolin_wisp_cpp
submodule, on top of the C ABI (include/gsdf_hip.h, section A1):

  spc_ops.quantize_points / points_to_neighbors / quantized_points_to_fpoints   sub_map.cpp:26-31, neural_mapping.cpp:757-758
  from_quantized_points(qpts, level) -> OctreeAS                                sub_map.cpp:33-34
  OctreeAS.query(xyz, level).pidx, .raymarch(origin, dir, "voxel", n), .get_quantized_points()
                                                  sub_map.cpp:79, local_map.cpp:467-476,514, neural_mapping.cpp:755

`OctreeAS` here is a dense bit pyramid in HBM (csrc/occupancy.hip), not a pointer octree; `pidx` carries only what the
reference reads from it (`> -1`): 0 for occupied, -1 for free.  Semantics: DESIGN.md SPEC A.9.
"""
import collections

import torch

from . import capi
from .capi import f32, ptr

QueryResult = collections.namedtuple("QueryResult", "pidx")
RaymarchResult = collections.namedtuple("RaymarchResult", "ridx samples depth_samples")


class spc_ops:
    @staticmethod
    def quantize_points(x, level):
        """[-1,1] floats -> int16 voxel coordinates at `level`."""
        res = 2 ** level
        return torch.floor(res * (x + 1.0) / 2.0).clamp(0, res - 1).to(torch.int16)

    @staticmethod
    def points_to_neighbors(qpts):
        """[n,3] -> [n,27,3]: the 3x3x3 neighbourhood (x fastest), NOT clamped (the caller clamps, sub_map.cpp:31)."""
        r = torch.arange(-1, 2, device=qpts.device, dtype=qpts.dtype)
        offs = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(27, 3).flip(-1)
        return qpts[:, None, :] + offs[None]

    @staticmethod
    def points_to_corners(qpts):
        """[n,3] -> [n,8,3]: the 8 corners of each voxel (x fastest)."""
        r = torch.arange(0, 2, device=qpts.device, dtype=qpts.dtype)
        offs = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(8, 3).flip(-1)
        return qpts[:, None, :] + offs[None]

    @staticmethod
    def quantized_points_to_fpoints(qpts, level):
        """voxel coordinates -> the voxel's minimum corner in [-1,1] (kaolin's convention)."""
        return qpts.to(torch.float32) * (2.0 / 2 ** level) - 1.0


class OctreeAS:
    def __init__(self, level, grid):
        self.level, self.grid = int(level), grid

    @property
    def max_level(self):
        return self.level

    @classmethod
    def from_points(cls, xyz_m1p1, level, dilate27=False):
        """quantize_points -> unique -> (points_to_neighbors, clamp) -> from_quantized_points in one build
        (SubMap::update_octree_as, sub_map.cpp:22-35)."""
        L = capi.lib()
        nbytes = L.gsdf_occ_bytes(int(level))
        if nbytes == 0:
            raise RuntimeError(f"OctreeAS: level {level} outside [1,12]")
        xyz = xyz_m1p1.contiguous().float()
        grid = torch.empty(nbytes // 4, dtype=torch.int32, device=xyz.device)
        capi.check(L.gsdf_occ_build(int(level), xyz.shape[0], f32(xyz), int(bool(dilate27)), ptr(grid), capi.stream()),
                   "occ_build")
        return cls(level, grid)

    def query(self, xyz_m1p1, level=-1):
        """-> QueryResult(pidx int64 [n]): 0 where the cell at `level` (default: finest) is occupied, else -1."""
        xyz = xyz_m1p1.contiguous().float()
        mask = torch.empty(xyz.shape[0], dtype=torch.uint8, device=xyz.device)
        capi.check(capi.lib().gsdf_occ_query(self.level, -1 if level is None else int(level), xyz.shape[0], f32(xyz),
                                             ptr(self.grid), ptr(mask), capi.stream()), "occ_query")
        return QueryResult(mask.to(torch.int64) - 1)

    def query_world_mask(self, xyz_world, origin, map_size_inv, level=-1):
        """bool [n]: `query(((xyz - origin) * 2) * map_size_inv, level).pidx > -1` in one launch (SubMap::get_valid_mask)."""
        import ctypes as C
        xyz = xyz_world.contiguous().float()
        mask = torch.empty(xyz.shape[0], dtype=torch.bool, device=xyz.device)
        capi.check(capi.lib().gsdf_occ_query_world(self.level, -1 if level is None else int(level), xyz.shape[0], f32(xyz),
                                                   (C.c_float * 3)(*origin), float(map_size_inv), ptr(self.grid),
                                                   ptr(mask), capi.stream()), "occ_query_world")
        return mask

    def visible_set(self, samples_world, visibilities, samples_weights, vis_thresh, origin, map_size_inv, level=-1):
        """The visible, occupancy-valid splat samples of the joint iteration (neural_mapping.cpp:423-437) in three launches and one
        size read-back: -> (ids int64 [n_valid] increasing = nonzero(get_valid_mask(samples) & (visibilities > thr)),
        w_all [M] = samples_weights * visibilities)."""
        import ctypes as C
        xyz = samples_world.detach().contiguous().float()
        n = xyz.shape[0]
        vis, sw = visibilities.detach().reshape(-1).contiguous().float(), samples_weights.detach().reshape(-1).contiguous().float()
        w_all = torch.empty(n, device=xyz.device)
        ids = torch.empty(n, dtype=torch.int64, device=xyz.device)
        L = capi.lib()
        ws = torch.empty(L.gsdf_visible_set_ws_bytes(n), dtype=torch.uint8, device=xyz.device)
        n_valid = capi.count_via_host_word(lambda count: capi.check(
            L.gsdf_visible_set(self.level, -1 if level is None else int(level), n, f32(xyz), (C.c_float * 3)(*origin), float(map_size_inv), ptr(self.grid),
                               f32(vis), f32(sw), float(vis_thresh), f32(w_all), ptr(ids), count, ptr(ws), capi.stream()), "visible_set"), xyz.device, upper=n)
        return ids[:n_valid], w_all

    def get_quantized_points(self):
        """-> int16 [V,3] occupied finest-level voxels (x-fastest linear order)."""
        L = capi.lib()
        n_words = max(1, 8 ** self.level // 32)
        counts = torch.empty(n_words, dtype=torch.int32, device=self.grid.device)
        capi.check(L.gsdf_occ_voxel_counts(self.level, ptr(self.grid), ptr(counts), capi.stream()), "occ_voxel_counts")
        incl = torch.cumsum(counts, 0, dtype=torch.int64)
        total = int(incl[-1].item())
        offs = (incl - counts).contiguous()
        vox = torch.empty(total, 3, dtype=torch.int16, device=self.grid.device)
        if total:
            capi.check(L.gsdf_occ_voxel_list(self.level, ptr(self.grid), ptr(offs), ptr(vox), capi.stream()), "occ_voxel_list")
        return vox

    def raymarch(self, origins_m1p1, dirs, raymarch_type="voxel", num_samples=1):
        """-> RaymarchResult(ridx int64 [S], samples [S,3], depth_samples [S,1]): `num_samples` stratified midpoints in
        every occupied finest-level voxel a ray crosses, rays in order, front to back (LocalMap::sample, :467-476)."""
        if raymarch_type != "voxel":
            raise RuntimeError("OctreeAS.raymarch: only the 'voxel' mode the reference uses is implemented")
        L = capi.lib()
        o, d = origins_m1p1.contiguous().float(), dirs.contiguous().float()
        n = o.shape[0]
        counts = torch.empty(n, dtype=torch.int32, device=o.device)
        capi.check(L.gsdf_occ_raymarch_count(self.level, n, f32(o), f32(d), ptr(self.grid), ptr(counts), capi.stream()),
                   "occ_raymarch_count")
        incl = torch.cumsum(counts, 0, dtype=torch.int64)
        total = int(incl[-1].item()) * int(num_samples) if n else 0
        offs = (incl - counts).contiguous()
        ridx = torch.empty(total, dtype=torch.int32, device=o.device)
        samples = torch.empty(total, 3, dtype=torch.float32, device=o.device)
        depth = torch.empty(total, 1, dtype=torch.float32, device=o.device)
        if total:
            capi.check(L.gsdf_occ_raymarch_fill(self.level, n, f32(o), f32(d), ptr(self.grid), ptr(offs), int(num_samples),
                                                ptr(ridx), f32(samples), f32(depth), capi.stream()), "occ_raymarch_fill")
        return RaymarchResult(ridx.to(torch.int64), samples, depth)


def from_quantized_points(qpts, level):
    """kaolin_wisp_cpp's factory (sub_map.cpp:33-34): builds from int voxel coordinates (their centres are re-quantised)."""
    centres = (qpts.to(torch.float32) + 0.5) * (2.0 / 2 ** level) - 1.0
    return OctreeAS.from_points(centres, level, dilate27=False)


def write_points_ply(path, xyz):
    """ply_utils::export_to_ply of a bare point set: binary little-endian PLY, float x y z."""
    import numpy as np
    pts = xyz.detach().cpu().float().numpy().astype("<f4")
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nend_header\n" % pts.shape[0]).encode("ascii"))
        f.write(np.ascontiguousarray(pts).tobytes())


def read_points_ply(path, device="cpu"):
    """-> {"xyz": [n,3]} from a binary little-endian PLY whose vertex element starts with float x y z
    (ply_utils::read_ply_file_to_map_tensor as neural_mapping.cpp:1369-1373 uses it)."""
    import numpy as np
    with open(path, "rb") as f:
        names, n, fmt = [], 0, None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                n = int(line.split()[2])
            elif line.startswith("property"):
                parts = line.split()
                if parts[1] != "float":
                    raise RuntimeError("read_points_ply: only float properties are supported")
                names.append(parts[2])
            elif line == "end_header":
                break
        if fmt != "binary_little_endian" or names[:3] != ["x", "y", "z"]:
            raise RuntimeError("read_points_ply: expected binary_little_endian with leading x y z")
        data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    return {"xyz": torch.from_numpy(np.array(data[:, :3], dtype=np.float32)).to(device)}
