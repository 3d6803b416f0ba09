"""Decoder kernels of the analytic configuration against the batch size: fixed cost (weight staging, partial-buffer exit) vs. per-row cost.
python tools/exp_mlp_sweep.py"""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gs_sdf_amd.capi as capi
from gs_sdf_amd.capi import f32, ptr
L = capi.lib()
dev = torch.device("cuda:0")
dims = [int(v) for v in os.environ.get("DIMS", "32,64,64,64,64,2").split(",")]
BIAS = os.environ.get("BIAS", "1") == "1"
SIZES = [int(v) for v in os.environ.get("SIZES", "1024,8192,40000,160000,327000,654000,1308000").split(",")]
nl, dims_c = len(dims) - 1, (C.c_int * len(dims))(*dims)
g = torch.Generator().manual_seed(0)
nw = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
W = (torch.randn(nw, generator=g) * 0.15).to(dev); bias = (torch.randn(sum(dims[1:]), generator=g) * 0.05).to(dev)
out_all = {}


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record(); fn()
    ev[reps].record(); torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return round(t[reps // 2] * 1e3, 1)


for n in SIZES:
    x = torch.randn(n, 32, generator=g).to(dev)
    out = torch.empty(n, dims[-1], device=dev); acts = torch.empty(L.gsdf_mlp_acts_floats(n, nl), device=dev)
    e0 = torch.zeros(n, dims[-1], device=dev); e0[:, 0] = 1
    vo = torch.randn(n, dims[-1], generator=g).to(dev)
    vv = torch.randn(n, 32, generator=g).to(dev)
    bws = torch.empty(L.gsdf_mlp_bwd_ws_bytes(n, nl), dtype=torch.uint8, device=dev)
    ws2 = torch.empty(L.gsdf_mlp_bwd_bwd_ws_bytes(n, nl), dtype=torch.uint8, device=dev)
    g0, gv, gw, gb = torch.empty(n, 32, device=dev), torch.empty(n, dims[-1], device=dev), torch.zeros_like(W), torch.zeros_like(bias)
    r = {}
    r["fwd_saving_acts_us"] = timed(lambda: capi.check(L.gsdf_mlp_fwd(n, nl, dims_c, f32(W), (f32(bias) if BIAS else None), f32(x), f32(out), f32(acts), capi.stream()), "fwd"))
    r["fwd_only_us"] = timed(lambda: capi.check(L.gsdf_mlp_fwd(n, nl, dims_c, f32(W), (f32(bias) if BIAS else None), f32(x), f32(out), None, capi.stream()), "fwd"))
    r["e0_backward_lean_us"] = timed(lambda: capi.check(L.gsdf_mlp_bwd(n, nl, dims_c, f32(W), (f32(bias) if BIAS else None), f32(x), f32(acts), f32(e0), f32(g0), None, None, None, capi.stream()), "bwd lean"))
    r["backward_full_us"] = timed(lambda: capi.check(L.gsdf_mlp_bwd(n, nl, dims_c, f32(W), (f32(bias) if BIAS else None), f32(x), f32(acts), f32(vo), f32(g0), f32(gw), (f32(gb) if BIAS else None), ptr(bws), capi.stream()), "bwd full"))
    r["double_backward_lean_us"] = timed(lambda: capi.check(L.gsdf_mlp_bwd_bwd(n, nl, dims_c, f32(W), f32(acts), f32(e0), None, f32(vv), f32(gv), f32(gw), ptr(ws2), capi.stream()), "bwd_bwd lean"))
    out_all[n] = r
    print(n, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out_all, open(os.path.join(ROOT, "gpurun_out", "mlp_sweep.json"), "w"), indent=1)
