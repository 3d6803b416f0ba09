"""Experiment (run on the GPU box): the analytic configuration's table scatter (gsdf_hashgrid_bwd_binned2: first + second order contributions,
no stencil structure) at the joint iteration's batch size, alone.  Usage: python tools/exp_scatter2.py [n ...]   (per-kernel times: run it under
rocprofv3 --kernel-trace --stats)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.capi as capi  # noqa: E402

dev = torch.device("cuda:0")
L = capi.lib()
cfg = (16, 2, 19, 32, 2.0)
total = 15269888
ns = [int(a) for a in sys.argv[1:]] or [436000]
for n in ns:
    g = torch.Generator().manual_seed(0)
    # a frustum-like cloud: a cone from a corner of the unit cube (the bench's visible splats), not a uniform box
    t = torch.rand(n, 1, generator=g) ** (1 / 3)
    d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.8, 0.6]]) + 0.45 * torch.randn(n, 3, generator=g), dim=1)
    x = (0.08 + 0.8 * t * d.abs()).clamp(0.02, 0.98).contiguous().to(dev)
    v = (torch.randn(n, 32, generator=g) * 1e-3).to(dev)
    v2 = (torch.randn(n, 32, generator=g) * 1e-3).to(dev)
    vv = torch.randn(n, 3, generator=g).to(dev)
    out = torch.zeros(total, 2, device=dev)
    nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(n, *cfg)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def run():
        capi.check(L.gsdf_hashgrid_bwd_binned2(n, *cfg, capi.f32(x), capi.f32(v), capi.f32(v2), capi.f32(vv), capi.f32(out), capi.ptr(ws), nbytes,
                                               capi.stream()), "binned2")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    out.zero_()
    run()
    torch.cuda.synchronize()
    print(f"n={n}: binned2 {a.elapsed_time(b) / reps:.3f} ms per call; checksum {float(out.double().abs().sum()):.9e}", flush=True)
