"""Experiment (GPU box): times the tile binning (hand-written depth-first radix sort; the rocPRIM path it was compared with in round 2
has been removed).  Usage: python tools/exp_binning.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.ops as ops, gs_sdf_amd.synth as synth
dev = torch.device("cuda:0")
for (N, W, H) in ((1_000_000, 1920, 1080), (300_000, 1200, 680), (3_000_000, 640, 512)):
    sc = synth.make_scene(N, W, H, sh_degree=0, seed=0)
    vm = synth.make_views(2, seed=1)[1:].to(dev)
    cam, gid, radii, m2d, dep, rt, nrm, smp, sw = ops.fully_fused_projection_2dgs(sc["means"].to(dev), sc["quats"].to(dev), sc["log_scales"].exp().to(dev),
                                                                                 vm, sc["K"].to(dev), W, H, 0.05, 300.0, 0.0)
    def run():
        return ops.tile_encode(W, H, 16, m2d, radii, dep, True, 1, cam, gid, return_isect_ids=True)
    out = run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('GSDF_BINNING_SORT','own')}: N={N} {W}x{H} M={cam.numel()} I={out[1].numel()}: {a.elapsed_time(b)/10:.3f} ms (incl. 1 host sync), "
          f"checksum {int(out[1].long().sum())} {int(out[3].sum() % 1000003)} {int(out[2].long().sum())}", flush=True)
