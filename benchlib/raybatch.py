"""The per-iteration SDF ray batch of the reference, built inside the timed step."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class RayBatcher:
    """The reference's per-iteration SDF ray batch, built INSIDE the timed step:
      NeuralSLAM::sdf_train_batch_iter (neural_mapping.cpp:138-164): k_batch_num random indices into the HOST-side depth pack
        (train_depth_pack_ lives on the CPU, :145-156), gather, copy to the device;
      NeuralSLAM::sample (:73-104) = gsdf_model::sample_rays (host/src/local_map.cpp): LocalMap::sample — octree ray march, one sample per
        occupied voxel crossed (gsdf occ_raymarch kernels), + free_sample_num stratified samples, those in front of the surface kept
        (local_map.cpp:449-509) — + surface_sample_num samples at depth - N(0, sample_std), targets truncated at +-truncated_dis, + the ray end
        points, in-range filter;
      the throttle of the training loop (:324-330): k_batch_num = min(batch_pt_num / EMA(points per ray), batch_pt_num), so that a batch
        holds ~batch_pt_num = 32768 points.
    The batch depends on the occupancy structure and the rays only, never on the parameters: it is issued ONE STEP AHEAD on a stream of its
    own (a data-loader prefetch), so its size read-backs (nonzero) wait for its own small kernels, not for the training step in flight.
    Synthetic depth pack: rays from the 200 camera centres to splat centres (every ray ends in an occupied leaf), 10000 per view."""

    def __init__(self, host, sc, views, dev, batch_pt_num=32768, rays_per_view=10000, leaf=0.0625, map_size=16.0, seed=7):
        self.host, self.dev, self.batch_pt_num = host, dev, batch_pt_num
        cfg = host.MapConfig()
        cfg.leaf_size, cfg.inner_map_size = leaf, map_size - 2 * leaf
        self.lm = host.LocalMap(torch.tensor([0.0, 0.0, 5.5]), cfg)
        self.lm.update_octree_as(sc["means"].to(dev), False)
        g = torch.Generator().manual_seed(7)            # the depth pack is the data set: the same on every rank; `seed` drives the draws
        c2w = torch.linalg.inv(views.cpu().double())
        centres = c2w[:, :3, 3].float()                                               # camera centres in the world
        V, N = centres.shape[0], sc["means"].shape[0]
        idx = torch.randint(0, N, (V, rays_per_view), generator=g)
        end = sc["means"][idx.reshape(-1)]
        org = centres[:, None, :].expand(V, rays_per_view, 3).reshape(-1, 3)
        d = end - org
        depth = d.norm(dim=1, keepdim=True)
        pin = lambda t: t.contiguous().pin_memory()
        self.pack = dict(origin=pin(org), direction=pin(d / depth), depth=pin(depth), xyz=pin(end))   # the host-side depth pack
        self.n_rays = org.shape[0]
        self.k_batch_num, self.pts_per_ray = batch_pt_num, 1.0                        # nsdf_train: k_batch_num = k_batch_ray_num (= batch_pt_num)
        self.stream = torch.cuda.Stream(device=dev)
        self.gen = torch.Generator().manual_seed(seed + 1)
        self.sample_std, self.truncated_dis = 0.02, 3 * leaf                          # base.yaml: sample_std; truncated at 3 leaves
        self.ready = None
        self.hist = []
        # the throttle starts from its steady state (the reference reaches it after ~50 iterations of :324-330; a batch of 32768 RAYS in this
        # scene would be 3.6 M points): a calibration batch of 256 rays measures the points per ray
        self.k_batch_num = 256
        self.issue()
        n0, p0 = self.hist[-1]
        self.pts_per_ray = max(p0 / max(n0, 1), 1e-3)
        self.k_batch_num = max(1, min(int(self.batch_pt_num / self.pts_per_ray), self.batch_pt_num))
        self.ready, self.hist = None, []

    def issue(self):
        """queues the next batch on the prefetch stream -> nothing; `take()` hands it to the step"""
        n = int(self.k_batch_num)
        indices = (torch.rand(n, generator=self.gen) * self.n_rays).long().clamp_(0, self.n_rays - 1)      # :141-149
        with torch.cuda.stream(self.stream):
            rays = {k: v.index_select(0, indices).to(self.dev, non_blocking=True) for k, v in self.pack.items()}   # :151-156
            b = self.host.sample_rays(self.lm, rays, self.sample_std, self.truncated_dis, 3, True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        pt_n = int(b["xyz"].shape[0])
        self.pts_per_ray = self.pts_per_ray * 0.9 + (pt_n / max(n, 1)) * 0.1           # :324-327
        self.k_batch_num = max(1, min(int(self.batch_pt_num / self.pts_per_ray), self.batch_pt_num))
        self.hist.append((n, pt_n))
        self.ready = (b["xyz"].contiguous(), b["ray_sdf"].contiguous(), ev)

    def take(self, consumer_stream):
        if self.ready is None:
            self.issue()
        xyz, rsdf, ev = self.ready
        consumer_stream.wait_event(ev)
        xyz.record_stream(consumer_stream); rsdf.record_stream(consumer_stream)
        self.ready = None
        return xyz, rsdf

