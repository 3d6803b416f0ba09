"""The e0 backward and the decoder double backward at the headline's size: fp32-pipe pair with saved v_pre images against the lean pair on the
bf16 pipe (chain recomputed from the ReLU masks).  Times (HIP events, median of 20) and the largest differences of the results.
python tools/exp_mlp_lean.py [n_points]"""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gs_sdf_amd.capi as capi
from gs_sdf_amd.capi import f32, ptr
L = capi.lib()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 380_000
dims = [32, 64, 64, 64, 64, 2]
nl, dims_c = len(dims) - 1, (C.c_int * len(dims))(*dims)
g = torch.Generator().manual_seed(0)
nw = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
W = (torch.randn(nw, generator=g) * 0.15).to(dev); bias = (torch.randn(sum(dims[1:]), generator=g) * 0.05).to(dev)
x = torch.randn(n, 32, generator=g).to(dev)
out = torch.empty(n, 2, device=dev); acts = torch.empty(L.gsdf_mlp_acts_floats(n, nl), device=dev)
capi.check(L.gsdf_mlp_fwd(n, nl, dims_c, f32(W), f32(bias), f32(x), f32(out), f32(acts), capi.stream()), "fwd")
e0 = torch.zeros(n, 2, device=dev); e0[:, 0] = 1
vv = torch.randn(n, 32, generator=g).to(dev)
bws = torch.empty(L.gsdf_mlp_bwd_ws_bytes(n, nl), dtype=torch.uint8, device=dev)
ws2 = torch.empty(L.gsdf_mlp_bwd_bwd_ws_bytes(n, nl), dtype=torch.uint8, device=dev)
res = {}


def timed(name, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record(); fn()
    ev[reps].record(); torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    res[name] = t[reps // 2]


g0a, g0b = torch.empty(n, 32, device=dev), torch.empty(n, 32, device=dev)
timed("e0_backward_fp32_pipe_saving_v_pre", lambda: capi.check(L.gsdf_mlp_bwd(n, nl, dims_c, f32(W), f32(bias), f32(x), f32(acts), f32(e0), f32(g0a), None, None, ptr(bws), capi.stream()), "bwd"))
timed("e0_backward_lean_bf16_pipe", lambda: capi.check(L.gsdf_mlp_bwd(n, nl, dims_c, f32(W), f32(bias), f32(x), f32(acts), f32(e0), f32(g0b), None, None, None, capi.stream()), "bwd lean"))
gva, gvb = torch.empty(n, 2, device=dev), torch.empty(n, 2, device=dev)
gwa, gwb = torch.zeros_like(W), torch.zeros_like(W)
timed("double_backward_fp32_pipe_from_v_pre", lambda: capi.check(L.gsdf_mlp_bwd_bwd(n, nl, dims_c, f32(W), f32(acts), f32(e0), ptr(bws), f32(vv), f32(gva), f32(gwa), ptr(ws2), capi.stream()), "bwd_bwd"))
timed("double_backward_lean_recompute", lambda: capi.check(L.gsdf_mlp_bwd_bwd(n, nl, dims_c, f32(W), f32(acts), f32(e0), None, f32(vv), f32(gvb), f32(gwb), ptr(ws2), capi.stream()), "bwd_bwd lean"))
gwa.zero_(); gwb.zero_()
capi.check(L.gsdf_mlp_bwd_bwd(n, nl, dims_c, f32(W), f32(acts), f32(e0), ptr(bws), f32(vv), f32(gva), f32(gwa), ptr(ws2), capi.stream()), "bwd_bwd")
capi.check(L.gsdf_mlp_bwd_bwd(n, nl, dims_c, f32(W), f32(acts), f32(e0), None, f32(vv), f32(gvb), f32(gwb), ptr(ws2), capi.stream()), "bwd_bwd lean")
torch.cuda.synchronize()
rel = lambda a, b: float((a - b).abs().max() / (b.abs().mean() + 1e-30))
res["max_diff_over_mean"] = dict(g0=rel(g0b, g0a), g_vout=rel(gvb, gva), g_weights=rel(gwb, gwa))
res["n_points"] = n
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "mlp_lean.json"), "w"), indent=1)
