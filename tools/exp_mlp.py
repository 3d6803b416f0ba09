"""Experiment (GPU box): the three decoder kernels at the joint iteration's batch size.  Usage: python tools/exp_mlp.py [B]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.capi as capi
dev = torch.device("cuda:0"); L = capi.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2744000
dims = [32, 64, 64, 64, 64, 2] if os.environ.get("EXP_TORCH_TOPOLOGY", "1") == "1" else [32, 64, 64, 64, 2]
nl = len(dims) - 1; dims_c = (C.c_int * len(dims))(*dims)
g = torch.Generator().manual_seed(0)
W = (torch.randn(sum(i * o for i, o in zip(dims[:-1], dims[1:])), generator=g) * 0.2).to(dev)
x = torch.randn(B, 32, generator=g).to(dev); out = torch.empty(B, 2, device=dev)
bias = (torch.randn(sum(dims[1:]), generator=g) * 0.1).to(dev) if os.environ.get("EXP_BIAS", "0") == "1" else None
v_b2 = torch.zeros_like(bias) if bias is not None else None
fb_ = lambda t: capi.f32(t) if t is not None else None
acts = torch.empty(L.gsdf_mlp_acts_floats(B, nl), device=dev); ws = torch.empty(L.gsdf_mlp_bwd_ws_bytes(B, nl), dtype=torch.uint8, device=dev)
v_out = torch.randn(B, 2, generator=g).to(dev); v_in = torch.empty_like(x); v_w = torch.zeros_like(W)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
f = t(lambda: capi.check(L.gsdf_mlp_fwd(B, nl, dims_c, capi.f32(W), None, capi.f32(x), capi.f32(out), capi.f32(acts), capi.stream()), "f"))
fi = t(lambda: capi.check(L.gsdf_mlp_fwd(B, nl, dims_c, capi.f32(W), None, capi.f32(x), capi.f32(out), None, capi.stream()), "f"))
bd = t(lambda: capi.check(L.gsdf_mlp_bwd(B, nl, dims_c, capi.f32(W), None, capi.f32(x), capi.f32(acts), capi.f32(v_out), capi.f32(v_in), None, None, capi.ptr(ws), capi.stream()), "b"))
bw = t(lambda: capi.check(L.gsdf_mlp_bwd_weights(B, nl, dims_c, 0, capi.f32(x), capi.f32(acts), capi.f32(v_out), capi.ptr(ws), capi.f32(v_w), None, capi.stream()), "w"))
fl = 2 * sum(i * o for i, o in zip(dims[:-1], dims[1:])) * B / 1e9
print(f"B={B} dims={dims} MFMA={os.environ.get('GSDF_MLP_MFMA','bf16x3')}: fwd {f:.3f} ms ({fl/f:.0f} TF/s) fwd(no acts) {fi:.3f} bwd_data {bd:.3f} ({fl/bd:.0f}) bwd_weights {bw:.3f} ({fl/bw:.0f})", flush=True)
# one-pass backward (both gradients) against the two fp32-pipe kernels on the same saved activations
v_w2 = torch.zeros_like(W); v_in2 = torch.empty_like(x)
capi.check(L.gsdf_mlp_fwd(B, nl, dims_c, capi.f32(W), fb_(bias), capi.f32(x), capi.f32(out), capi.f32(acts), capi.stream()), "f")
fb = t(lambda: capi.check(L.gsdf_mlp_bwd(B, nl, dims_c, capi.f32(W), fb_(bias), capi.f32(x), capi.f32(acts), capi.f32(v_out), capi.f32(v_in2), capi.f32(v_w2), fb_(v_b2), capi.ptr(ws), capi.stream()), "b"))
if bias is not None: print(f"(with biases) one-pass backward {fb:.3f} ms")
v_w2.zero_(); v_w.zero_()
capi.check(L.gsdf_mlp_bwd(B, nl, dims_c, capi.f32(W), None, capi.f32(x), capi.f32(acts), capi.f32(v_out), capi.f32(v_in2), capi.f32(v_w2), None, capi.ptr(ws), capi.stream()), "b")
capi.check(L.gsdf_mlp_bwd(B, nl, dims_c, capi.f32(W), None, capi.f32(x), capi.f32(acts), capi.f32(v_out), capi.f32(v_in), None, None, capi.ptr(ws), capi.stream()), "b")
capi.check(L.gsdf_mlp_bwd_weights(B, nl, dims_c, 0, capi.f32(x), capi.f32(acts), capi.f32(v_out), capi.ptr(ws), capi.f32(v_w), None, capi.stream()), "w")
torch.cuda.synchronize()
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
ew = (v_w2.double() - v_w.double()).abs() / (v_w.double().abs().mean())
print(f"one-pass backward {fb:.3f} ms ({3 * fl / fb:.0f} TF/s incl. dW); vs two-kernel fp32: v_in rel-L2 {rel(v_in2, v_in):.2e}, v_w rel-L2 {rel(v_w2, v_w):.2e}, "
      f"v_w max |diff| / mean|ref| {float(ew.max()):.2e} at {int(ew.argmax())}, elements above 1e-4: {int((ew > 1e-4).sum())}", flush=True)
if int((ew > 1e-4).sum()):
    idx = torch.nonzero(ew > 1e-4).flatten()[:40].tolist(); off = 0; lay = []
    for l, (i, o) in enumerate(zip(dims[:-1], dims[1:])):
        lay.append((off, i, o)); off += i * o
    for k in idx:
        l = max(j for j, (o0, _, _) in enumerate(lay) if o0 <= k); o0, i, o = lay[l]
        print("  layer", l, "o", (k - o0) // i, "i", (k - o0) % i, "got", float(v_w2[k]), "ref", float(v_w[k]))
