"""M1 marching cubes (csrc/marching_cubes.hip) through the C ABI against oracle/mc_oracle.py: same deterministic ordering,
so vertices (fp32, no contraction on either side) and faces (int32) are compared exactly."""
import numpy as np
import pytest
import torch

from oracle import mc_oracle as mco

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def _sphere(res, r=0.6):
    ax = np.linspace(-1.0, 1.0, res, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (r - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)


@pytest.mark.parametrize("shape,thresh,kind", [((13, 11, 9), 0.1, "random"), ((40, 40, 40), 0.0, "sphere"), ((2, 2, 2), 0.5, "random"),
                                              ((65, 33, 17), -0.2, "random"), ((128, 128, 128), 0.05, "sphere"), ((5, 1, 7), 0.0, "random")])
@pytest.mark.parametrize("table", ["reference", "watertight"])
def test_marching_cubes_bit_exact(shape, thresh, kind, table):
    """table="reference": the oracle triangulates with the REFERENCE'S OWN triangle table (golden vector
    tests/golden/mc_triangle_table_reference.npy, from include/mesher/cumcubes/include/utils.cuh:31-289): vertex array equal
    and the face list equal cell by cell (hence the per-cell face SET equals the reference mesher's, whose order is whatever
    its atomics produce)."""
    from gs_sdf_amd.mesher import marching_cubes
    g = _sphere(shape[0]) if kind == "sphere" else np.random.default_rng(sum(shape)).standard_normal(shape).astype(np.float32)
    lower, upper = [-1.5, 0.25, 3.0], [2.5, 4.25, 11.0]
    v_ref, f_ref = mco.marching_cubes(g, thresh, lower, upper, table)
    v, f = marching_cubes(torch.from_numpy(g).to(dev), thresh, lower, upper, table)
    assert v.dtype == torch.float32 and f.dtype == torch.int32
    assert v.shape == v_ref.shape and f.shape == f_ref.shape and (kind != "sphere" or f.shape[0] > 1000)
    assert np.array_equal(v.cpu().numpy(), v_ref)
    assert np.array_equal(f.cpu().numpy(), f_ref)


def test_empty_surface_and_ply_export(tmp_path):
    from gs_sdf_amd.mesher import marching_cubes, save_mesh_as_ply
    v, f = marching_cubes(torch.ones(8, 8, 8, device=dev), 2.0, [0, 0, 0], [1, 1, 1])
    assert v.shape == (0, 3) and f.shape == (0, 3)
    g = torch.from_numpy(_sphere(24)).to(dev)
    v, f = marching_cubes(g, 0.0, [-1] * 3, [1] * 3)
    col = (torch.rand(v.shape[0], 3, device=dev) * 255).to(torch.uint8)
    path = str(tmp_path / "mesh.ply")
    save_mesh_as_ply(path, v, f, col)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex %d" % v.shape[0] in head and b"element face %d" % f.shape[0] in head
    assert len(body) == v.shape[0] * 15 + f.shape[0] * 16 and b"property list int int vertex_index" in head
    with pytest.raises(RuntimeError):
        marching_cubes(torch.ones(8, 8, device=dev), 0.0, [0] * 3, [1] * 3)
    with pytest.raises(RuntimeError):
        marching_cubes(torch.ones(4, 4, 4), 0.0, [0] * 3, [1] * 3)            # CPU tensor: no fallback


def test_sdf_network_meshing_slice():
    """LocalMap-style use (local_map.cpp:258-300): evaluate the SDF network on a grid, extract the zero level set."""
    import gs_sdf_amd.sdf as sdfm
    from gs_sdf_amd.mesher import marching_cubes
    lm = sdfm.LocalMap([0.0, 0.0, 0.0], 4.0, decoder_implementation=1, device=dev, seed=3)
    res = 48
    ax = torch.linspace(-1.5, 1.5, res, device=dev)
    pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    with torch.no_grad():
        # an untrained network has no surface: add the analytic sphere so that the path is exercised end to end
        sdf = lm.get_sdf(pts)[0].reshape(res, res, res) + (pts.norm(dim=1) - 1.0).reshape(res, res, res)
    v, f = marching_cubes(sdf, 0.0, [-1.5] * 3, [1.5 + 3.0 / (res - 1)] * 3)
    assert f.shape[0] > 2000 and float((v.norm(dim=1) - 1.0).abs().max()) < 0.05


def test_cpp_marching_cubes_wrapper_matches_python_mirror():
    """mc::marching_cubes_wrapper of the C++ operator layer (what the reference's cumcubes.cpp:9-27 calls)."""
    import gs_sdf_amd.hostlib as h
    from gs_sdf_amd.mesher import marching_cubes
    host = h.load()
    g = torch.from_numpy(_sphere(40)).to(dev)
    v, f = marching_cubes(g, 0.0, [-1, -1, -1], [1, 1, 1], "reference")          # the C++ wrapper uses the reference's table
    v2, f2 = host.marching_cubes(g, 0.0, [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0])
    assert torch.equal(v, v2) and torch.equal(f, f2) and f2.dtype == torch.int32
