"""Experiment (run on the GPU box): how much of the compositing kernels' evaluated (pixel, splat) work is useful.
Instrumented launches (gsdf_rasterize_2dgs_fwd_instr / _bwd_instr with counters) at the bench workload's shape.  Usage: python tools/exp_raster_pairs.py [workload]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gs_sdf_amd.capi as capi, gs_sdf_amd.ops as ops, gs_sdf_amd.synth as synth
sys.path.insert(0, ROOT)
from bench import WORKLOADS, WORKLOAD_SIGMA_PX
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_1M_1080p"
N, W, H, deg, replica = WORKLOADS[name]
sc = synth.make_scene(N, W, H, sh_degree=deg, seed=0, replica=replica, sigma_px=WORKLOAD_SIGMA_PX.get(name, (0.5, 4.0)))
vm = synth.make_views(2, seed=1)[1:2].to(dev)
cnt = torch.zeros(16, dtype=torch.int64, device=dev)
d = lambda t: t.to(dev)
with torch.no_grad():
    cam, gid, radii, m2d, dep, rt, nrm, smp, sw = ops.fully_fused_projection_2dgs(d(sc["means"]), d(sc["quats"]), d(sc["log_scales"].exp()), vm,
                                                                                  d(sc["K"]), W, H, 0.05, 300.0, 0.0)
    col = ops.get_view_colors(vm, d(sc["means"]), radii, d(sc["sh"]), cam, gid, deg)
    opa = torch.sigmoid(d(sc["logit_opacities"]))[gid].contiguous()
    tpg, flat, offs = ops.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid)
    fwd = ops.rasterize_fwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, counters=cnt)
    ug = {k: v.to(dev) for k, v in synth.upstream_grads(H, W, seed=2).items()}
    ops.rasterize_bwd_instr(m2d, rt, col, opa, nrm, W, H, offs, flat, fwd, ug, cnt)
torch.cuda.synchronize()
meta = dict(flatten_ids=flat, gaussian_ids=gid)
c = cnt.cpu().tolist()
I, M, P = int(meta["flatten_ids"].shape[0]), int(meta["gaussian_ids"].shape[0]), W * H
out = dict(workload=name, M=M, I=I, P=P, staged_tile_splat_pairs=I, all_pairs_without_culling=I * 256,
           fwd=dict(wave_iterations=c[0], lanes_evaluated=c[0] * 64, lanes_live=c[1], lanes_alpha_ok=c[2], lanes_blended=c[3],
                    useful_of_evaluated=c[3] / max(1, c[0] * 64), alpha_ok_of_live=c[2] / max(1, c[1]),
                    visits_with_no_useful_lane=c[7], fraction_of_visits_with_no_useful_lane=c[7] / max(1, c[0]),
                    mean_useful_lanes_in_a_useful_visit=c[2] / max(1, c[0] - c[7])),
           bwd=dict(wave_iterations=c[4], lanes_evaluated=c[4] * 64, lanes_replaying=c[5], lanes_blended=c[6],
                    useful_of_evaluated=c[6] / max(1, c[4] * 64)))
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"raster_pairs_{name}.json"), "w"), indent=1)
