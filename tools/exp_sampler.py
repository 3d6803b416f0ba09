"""Experiment (GPU box): the fused per-ray SDF batch sampler (gsdf_model::sample_rays: rand, randn, count, scan, fill) against the composed op chain
(sample_rays_composed: the reference's NeuralSLAM::sample on libtorch + OctreeAS::raymarch) at the bench's scene, for the step's throttled batch
(~294 rays -> ~32.7 k points) and for a full batch of 32768 rays.  Kernel time = HIP events around the call on an otherwise idle chip (the composed
chain's figure therefore includes its ~60 launches' gaps: it is latency, as in the step).
Usage: python tools/exp_sampler.py  ->  gpurun_out/r05_ray_sampler.json"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gs_sdf_amd.hostlib as hostlib
import gs_sdf_amd.synth as synth
host = hostlib.load()
dev = torch.device("cuda:0")
N, W, H = 1_000_000, 1920, 1080
sc = synth.make_scene(N, W, H, sh_degree=0, seed=0)
cfg = host.MapConfig()
cfg.leaf_size, cfg.inner_map_size = 0.0625, 16.0 - 2 * 0.0625
lm = host.LocalMap(torch.tensor([0.0, 0.0, 5.5]), cfg)
lm.update_octree_as(sc["means"].to(dev), False)
views = synth.make_views(200, seed=1)
c2w = torch.linalg.inv(views.double())
g = torch.Generator().manual_seed(7)
out = {}
for n in (294, 4096, 32768):
    idx = torch.randint(0, N, (n,), generator=g)
    org = c2w[torch.randint(0, 200, (n,), generator=g), :3, 3].float()
    end = sc["means"][idx]
    d = end - org
    depth = d.norm(dim=1, keepdim=True)
    rays = dict(origin=org.to(dev), direction=(d / depth).to(dev), depth=depth.to(dev), xyz=end.to(dev))
    res = {}
    for name, fn in (("fused", host.sample_rays), ("composed", host.sample_rays_composed)):
        torch.manual_seed(1)
        b = fn(lm, rays, 0.02, 0.1875, 3, True)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            torch.manual_seed(1)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record(); b = fn(lm, rays, 0.02, 0.1875, 3, True); e.record()
            torch.cuda.synchronize()
            ts.append((a.elapsed_time(e), (time.perf_counter() - t0) * 1e3))
        ts.sort()
        res[name] = {"gpu_ms_median": round(ts[2][0], 4), "wall_ms_median": round(sorted(t[1] for t in ts)[2], 4), "rows": int(b["xyz"].shape[0])}
        if name == "fused":
            keep = b
        else:
            res["identical_rows"] = bool(torch.equal(keep["ridx"], b["ridx"]) and torch.equal(keep["xyz"], b["xyz"]) and torch.equal(keep["ray_sdf"], b["ray_sdf"]))
    out[f"{n}_rays"] = res
    print(n, res, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": __doc__.split("Usage")[0].strip(), "scene": "cfg3: 1 M random splats, 1/16 m leaves in a 16 m cube (level 8), rays camera centre -> splat centre", **out},
          open(os.path.join(ROOT, "gpurun_out", "r05_ray_sampler.json"), "w"), indent=1)
