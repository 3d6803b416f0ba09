"""Experiment (GPU box): the stencil hash-grid forward (7 rows per point, Jacobian of the base rows) at the joint iteration's batch
size, points either uniform in the map or on a wall-like surface.  Usage: python tools/exp_fwd_stencil.py [n_points] [surface]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.capi as capi
dev = torch.device("cuda:0"); L = capi.lib(); cfg = (16, 2, 19, 32, 2.0); total = 15269888
n = int(sys.argv[1]) if len(sys.argv) > 1 else 494000
g = torch.Generator().manual_seed(0)
base = torch.rand(n, 3, generator=g) * 0.8 + 0.1
if len(sys.argv) > 2:       # a curved surface: what the visible splats' samples look like
    base[:, 2] = 0.5 + 0.1 * torch.sin(6 * base[:, 0]) * torch.cos(5 * base[:, 1])
d = 0.02 / 16.0
offs = torch.tensor([[0, 0, 0], [d, 0, 0], [-d, 0, 0], [0, d, 0], [0, -d, 0], [0, 0, d], [0, 0, -d]])
x = (base[None] + offs[:, None]).reshape(-1, 3).contiguous().to(dev); Bq = x.shape[0]
table = (torch.rand(total, 2, generator=g) * 2 - 1).to(dev)
feat = torch.empty(Bq, 32, device=dev); jac = torch.empty(n, 32, 3, device=dev)
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
t_st = t(lambda: capi.check(L.gsdf_hashgrid_fwd_stencil(Bq, n, n, *cfg, capi.f32(x), capi.f32(table), capi.f32(feat), capi.f32(jac), capi.stream()), "f"))
print(f"n={n} seq={os.environ.get('GSDF_HASHGRID_SEQ','0')} nt={os.environ.get('GSDF_HASHGRID_NT','0')}: stencil fwd+jac {t_st:.3f} ms  checksum {float(feat.double().sum()):.6f} {float(jac.double().sum()):.6f}", flush=True)
