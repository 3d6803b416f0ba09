"""Experiment (GPU box): hash-grid forward at the joint iteration's batch size.  Usage: python tools/exp_fwd.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.capi as capi
dev = torch.device("cuda:0"); L = capi.lib(); cfg = (16, 2, 19, 32, 2.0); total = 15269888
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2744000
g = torch.Generator().manual_seed(0)
n = B // 7; base = torch.rand(n, 3, generator=g) * 0.8 + 0.1; d = 0.02 / 16.0
offs = torch.tensor([[0, 0, 0], [d, 0, 0], [-d, 0, 0], [0, d, 0], [0, -d, 0], [0, 0, d], [0, 0, -d]])
x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous().to(dev); Bq = x.shape[0]
table = (torch.rand(total, 2, generator=g) * 2 - 1).to(dev)
feat = torch.empty(Bq, 32, device=dev); jac = torch.empty(n, 32, 3, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
t_plain = t(lambda: capi.check(L.gsdf_hashgrid_fwd(Bq, *cfg, capi.f32(x), capi.f32(table), capi.f32(feat), capi.stream()), "f"))
ref = feat.clone()
t_jac = t(lambda: capi.check(L.gsdf_hashgrid_fwd_jac_rows(Bq, n, *cfg, capi.f32(x), capi.f32(table), capi.f32(feat), capi.f32(jac), capi.stream()), "f"))
jref = jac.clone(); feat.zero_(); jac.zero_()
t_st = t(lambda: capi.check(L.gsdf_hashgrid_fwd_stencil(Bq, n, n, *cfg, capi.f32(x), capi.f32(table), capi.f32(feat), capi.f32(jac), capi.stream()), "f"))
print(f"stencil kernel: fwd+jac(n) {t_st:.3f} ms, feat equal {bool(torch.equal(ref, feat))}, jac equal {bool(torch.equal(jref, jac))}")
t_st0 = t(lambda: capi.check(L.gsdf_hashgrid_fwd_stencil(Bq, n, 0, *cfg, capi.f32(x), capi.f32(table), capi.f32(feat), None, capi.stream()), "f"))
print(f"stencil kernel, no jac: {t_st0:.3f} ms, feat equal {bool(torch.equal(ref, feat))}")
print(f"B={Bq} phases={os.environ.get('GSDF_HASHGRID_PHASES','1')}: fwd {t_plain:.3f} ms, fwd+jac(n) {t_jac:.3f} ms, equal {bool(torch.equal(ref, feat))}", flush=True)
