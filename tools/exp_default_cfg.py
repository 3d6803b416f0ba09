"""Experiment (run on the GPU box): per-kernel times of the fused analytic SDF batch (the reference's default configuration) at the
joint iteration's two batch sizes, one stream, nothing else on the chip.  Usage: python tools/exp_default_cfg.py [n ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gs_sdf_amd.ops as ops, gs_sdf_amd.sdf as sdfm

dev = torch.device("cuda:0")
lm = sdfm.LocalMap([0.0, 0.0, 5.5], 16.0, bce_sigma=0.02, decoder_implementation=0, device=dev, seed=5)
grp = lm.flatten(accumulate_table_grad_in_place=True)
g = torch.Generator().manual_seed(0)
for n in [int(a) for a in sys.argv[1:]] or [32768, 340000]:
    pts = ((torch.rand(n, 3, generator=g) - 0.5) * 14.0 + torch.tensor([0.0, 0.0, 5.5])).to(dev)
    gt = (torch.randn(n, 1, generator=g) * 0.02).to(dev)
    for rep in range(6):
        if rep == 2:
            ops.TIMERS.enable()
        with sdfm.grad_sinks_armed():
            lm.ray_loss_analytic(pts, gt, 0.02, 1.0, 0.1, 0.1).backward()
    t = ops.TIMERS.summary_ms("median")
    ops.TIMERS.disable()
    print(f"n={n}: " + ", ".join(f"{k} {v:.3f}" for k, v in t.items()) + f" | sum {sum(t.values()):.3f} ms", flush=True)
