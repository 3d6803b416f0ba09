"""Experiment (run on the GPU box): hash-grid table-gradient scatter, binned (no global atomics) vs atomic kernel,
at the batch sizes of the joint iteration.  Usage: python tools/exp_scatter.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.capi as capi  # noqa: E402

dev = torch.device("cuda:0")
L = capi.lib()
cfg = (16, 2, 19, 32, 2.0)
total = 15269888
Bs = [int(a) for a in sys.argv[1:]] or [229376, 458752, 2973000]
for B in Bs:
    g = torch.Generator().manual_seed(0)
    n = B // 7
    base = torch.rand(n, 3, generator=g) * 0.8 + 0.1
    d = 0.02 / 16.0
    offs = torch.tensor([[0, 0, 0], [d, 0, 0], [-d, 0, 0], [0, d, 0], [0, -d, 0], [0, 0, d], [0, 0, -d]])
    x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous().to(dev)
    Bq = x.shape[0]
    v = torch.randn(Bq, 32, generator=g).to(dev)
    table = torch.zeros(total, 2, device=dev)
    out_a, out_b = torch.zeros(total, 2, device=dev), torch.zeros(total, 2, device=dev)
    nbytes = L.gsdf_hashgrid_bwd_binned_ws_bytes(Bq, *cfg)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def timeit(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    t_at = timeit(lambda: capi.check(L.gsdf_hashgrid_bwd(Bq, *cfg, capi.f32(x), capi.f32(table), capi.f32(v), capi.f32(out_a), None, capi.stream()), "a"))
    t_bin = timeit(lambda: capi.check(L.gsdf_hashgrid_bwd_binned(Bq, *cfg, capi.f32(x), capi.f32(v), capi.f32(out_b), capi.ptr(ws), nbytes, capi.stream()), "b"))
    out_c = torch.zeros(total, 2, device=dev)
    t_st = timeit(lambda: capi.check(L.gsdf_hashgrid_bwd_binned_stencil(Bq, n, 5, *cfg, capi.f32(x), capi.f32(v), capi.f32(out_c), capi.ptr(ws), nbytes, capi.stream()), "c"))
    err_c = float((out_c - out_b).abs().max() / out_b.abs().mean())
    err = float((out_a - out_b).abs().max() / out_a.abs().mean())
    print(f"B={Bq}: atomic {t_at:.3f} ms, binned {t_bin:.3f} ms, binned+stencil merge {t_st:.3f} ms (vs binned {err_c:.1e}) (ws {nbytes / 2**30:.2f} GiB), max|diff|/mean|ref| {err:.2e}", flush=True)
