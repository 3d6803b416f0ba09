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
