"""Launch shapes of the SDF path's kernels that a caller selects through the C ABI (no environment switches).

gsdf_hashgrid_fwd_stencil_resident(w) (a per-thread hint) launches the stencil hash-grid forward as a resident grid — w workgroups per CU that walk
the chunks — instead of one workgroup per chunk; gsdf_extras::JointIteration uses w = 3 while the other leg shares the chip.  The
walk changes which workgroup computes a chunk, never a chunk's arithmetic: features and Jacobians stay bit-identical to the row-major kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("resident", [1, 2, 3])
@pytest.mark.parametrize("n", [20011, 150000])
def test_resident_grid_is_bit_identical(resident, n):
    import gs_sdf_amd.capi as capi
    import gs_sdf_amd.sdf as sdf
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(29)
    delta = 1e-3
    base = torch.rand(n, 3, generator=g) * 0.6 + 0.2
    offs = torch.tensor([[0, 0, 0], [delta, 0, 0], [-delta, 0, 0], [0, delta, 0], [0, -delta, 0], [0, 0, delta], [0, 0, -delta]])
    x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous().to(dev)
    B = x.shape[0]
    c = (16, 2, 19, 32, 2.0)
    enc = sdf.TCNNEncoding(3, None, "enc", dev, seed=3)
    table = (enc.params_.detach() * 1e3).contiguous()
    L = capi.lib()
    ref, got = torch.empty(B, 32, device=dev), torch.zeros(B, 32, device=dev)
    jref, jgot = torch.empty(n, 32, 3, device=dev), torch.zeros(n, 32, 3, device=dev)
    capi.check(L.gsdf_hashgrid_fwd_jac_rows(B, n, *c, capi.f32(x), capi.f32(table), capi.f32(ref), capi.f32(jref), capi.stream()), "rows")
    before = L.gsdf_hashgrid_fwd_stencil_resident(resident)
    try:
        capi.check(L.gsdf_hashgrid_fwd_stencil(B, n, n, *c, capi.f32(x), capi.f32(table), capi.f32(got), capi.f32(jgot), capi.stream()), "stencil")
    finally:
        L.gsdf_hashgrid_fwd_stencil_resident(before)
    torch.cuda.synchronize()
    assert torch.equal(ref, got) and torch.equal(jref, jgot)


@pytest.mark.parametrize("resident", [0, 3])
@pytest.mark.parametrize("n_a,m_b,n_ids", [(30000, 60000, 41000), (500, 700, 0), (0, 30000, 20011), (4096, 0, 0)])
def test_stencil_forward_from_world_points_is_the_two_launches(resident, n_a, m_b, n_ids):
    """gsdf_hashgrid_fwd_stencil_points = gsdf_sdf_query_points2 + gsdf_hashgrid_fwd_stencil in one launch: the 7 encoder rows of a point are made
    inside the encoder's kernel from the world point (per-ray rows, then rows ids[j] of the splat samples).  Same x01 rows, same features, same
    Jacobians, bit for bit — full grid and resident grid, the XCD level map (7 n >= 65536) and the single-map launch below it, with and without ids."""
    import ctypes as C
    import gs_sdf_amd.capi as capi
    import gs_sdf_amd.sdf as sdf
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    a = ((torch.rand(max(n_a, 1), 3, generator=g) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
    b = ((torch.rand(max(m_b, 1), 3, generator=g) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
    ids = torch.randperm(m_b, generator=g)[:n_ids].sort().values.to(dev) if n_ids else None
    n_b = n_ids if n_ids else m_b
    n = n_a + n_b
    c = (16, 2, 19, 32, 2.0)
    enc = sdf.TCNNEncoding(3, None, "enc", dev, seed=3)
    table = (enc.params_.detach() * 1e3).contiguous()
    L = capi.lib()
    org, inv, delta = (C.c_float * 3)(0.0, 0.0, 5.5), 1.0 / 16.0, 0.02
    p = lambda t: None if t is None else t.data_ptr()
    x_ref, x_got = torch.empty(7 * n, 3, device=dev), torch.zeros(7 * n, 3, device=dev)
    f_ref, f_got = torch.empty(7 * n, 32, device=dev), torch.zeros(7 * n, 32, device=dev)
    j_ref, j_got = torch.empty(n, 32, 3, device=dev), torch.zeros(n, 32, 3, device=dev)
    before = L.gsdf_hashgrid_fwd_stencil_resident(resident)
    try:
        capi.check(L.gsdf_sdf_query_points2(n_a, p(a), n_b, p(b), p(ids), 1, delta, org, inv, p(x_ref), capi.stream()), "query_points2")
        capi.check(L.gsdf_hashgrid_fwd_stencil(7 * n, n, n, *c, p(x_ref), p(table), p(f_ref), p(j_ref), capi.stream()), "stencil")
        capi.check(L.gsdf_hashgrid_fwd_stencil_points(n_a, p(a), n_b, p(b), p(ids), delta, org, inv, 1, *c, p(table), p(x_got), p(f_got), p(j_got),
                                                      capi.stream()), "stencil_points")
    finally:
        L.gsdf_hashgrid_fwd_stencil_resident(before)
    torch.cuda.synchronize()
    assert torch.equal(x_ref, x_got), "query rows"
    assert torch.equal(f_ref, f_got), "features"
    assert torch.equal(j_ref, j_got), "Jacobians"
    assert float(f_got.abs().sum()) > 0 and float(j_got.abs().sum()) > 0


def test_stencil_forward_from_world_points_with_64_bit_row_offsets():
    """3.3 M base points: 7 n x 16 levels x 6 floats pass 2^31, the kernels take their 64-bit row-offset instantiations — same bits from both entry points."""
    import ctypes as C
    import gs_sdf_amd.capi as capi
    import gs_sdf_amd.sdf as sdf
    dev = torch.device("cuda:0")
    n = 3_300_000
    g = torch.Generator().manual_seed(41)
    a = ((torch.rand(n, 3, generator=g) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
    c = (16, 2, 19, 32, 2.0)
    table = (sdf.TCNNEncoding(3, None, "enc", dev, seed=3).params_.detach() * 1e3).contiguous()
    L = capi.lib()
    org, inv, delta = (C.c_float * 3)(0.0, 0.0, 5.5), 1.0 / 16.0, 0.02
    p = lambda t: None if t is None else t.data_ptr()
    x_ref, x_got = torch.empty(7 * n, 3, device=dev), torch.zeros(7 * n, 3, device=dev)
    f_ref, f_got = torch.empty(7 * n, 32, device=dev), torch.zeros(7 * n, 32, device=dev)
    j_ref, j_got = torch.empty(n, 32, 3, device=dev), torch.zeros(n, 32, 3, device=dev)
    capi.check(L.gsdf_sdf_query_points2(n, p(a), 0, None, None, 1, delta, org, inv, p(x_ref), capi.stream()), "query_points2")
    capi.check(L.gsdf_hashgrid_fwd_stencil(7 * n, n, n, *c, p(x_ref), p(table), p(f_ref), p(j_ref), capi.stream()), "stencil")
    capi.check(L.gsdf_hashgrid_fwd_stencil_points(n, p(a), 0, None, None, delta, org, inv, 1, *c, p(table), p(x_got), p(f_got), p(j_got), capi.stream()),
               "stencil_points")
    torch.cuda.synchronize()
    assert torch.equal(x_ref, x_got) and torch.equal(f_ref, f_got) and torch.equal(j_ref, j_got)
    assert float(f_got[-1].abs().sum()) > 0
