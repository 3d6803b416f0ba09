"""Deterministic synthetic scenes for parity tests and bench.py.

Everything is generated on the CPU from fixed seeds so that the oracle and the HIP path see
identical bits.  Shapes follow BASELINE.json's configs; values follow the reference's own
initialisers where they exist (random quaternion: include/neural_gaussian/gauss_utils.hpp:31-46).
"""
import math

import torch


def intrinsics(W, H, replica=False):
    """K [1,3,3].  Replica room intrinsics (include/data_loader/data_parsers/replica_parser.hpp:75-80)
    for the 1200x680 configs, fx=fy=0.8W otherwise."""
    if replica:
        fx = fy = 600.0
        cx, cy = 599.5, 339.5
    else:
        fx = fy = 0.8 * W
        cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    return torch.tensor([[[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]], dtype=torch.float32)


def random_quat(n, gen):
    u, v, w = (torch.rand(n, generator=gen) for _ in range(3))
    return torch.stack([torch.sqrt(1 - u) * torch.sin(2 * math.pi * v), torch.sqrt(1 - u) * torch.cos(2 * math.pi * v),
                        torch.sqrt(u) * torch.sin(2 * math.pi * w), torch.sqrt(u) * torch.cos(2 * math.pi * w)], -1)


def make_scene(N, W, H, sh_degree=0, seed=0, replica=False, sigma_px=(0.5, 4.0)):
    """Random-Gaussian scene: every centre in-frustum at depth U(1,10); projected 1-sigma
    log-uniform in `sigma_px` pixels (fronto-parallel).  Returns raw (pre-activation) params in
    the reference's parameterisation (neural_gaussian.cpp:463-492): log-scales, logit-opacity."""
    g = torch.Generator().manual_seed(seed)
    K = intrinsics(W, H, replica)
    fx, fy, cx, cy = K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2]
    u = torch.rand(N, generator=g) * W
    v = torch.rand(N, generator=g) * H
    z = 1.0 + 9.0 * torch.rand(N, generator=g)
    means = torch.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], -1)
    lo, hi = math.log(sigma_px[0]), math.log(sigma_px[1])
    s_uv = (z / fx)[:, None] * torch.exp(lo + (hi - lo) * torch.rand(N, 2, generator=g))
    scales = torch.cat([s_uv, s_uv[:, :1]], -1)
    quats = random_quat(N, g)
    opac = 0.05 + 0.9 * torch.rand(N, generator=g)
    Kb = (sh_degree + 1) ** 2
    sh = torch.cat([torch.rand(N, 1, 3, generator=g), 0.05 * torch.randn(N, Kb - 1, 3, generator=g)], 1)
    return dict(means=means.contiguous(), quats=quats.contiguous(), log_scales=torch.log(scales).contiguous(),
                logit_opacities=torch.logit(opac).contiguous(), sh=sh.contiguous(), K=K, W=W, H=H,
                sh_degree=sh_degree)


def make_views(n_views, seed=1):
    """World->camera matrices [V,4,4]: view 0 is the identity, the others rotate by U(-10,10) deg
    about each axis and translate by U(-0.5,0.5)^3."""
    g = torch.Generator().manual_seed(seed)
    vms = [torch.eye(4)]
    for _ in range(1, n_views):
        a = (torch.rand(3, generator=g) * 2 - 1) * math.radians(10.0)
        t = torch.rand(3, generator=g) - 0.5
        cx, sx, cy, sy, cz, sz = math.cos(a[0]), math.sin(a[0]), math.cos(a[1]), math.sin(a[1]), math.cos(a[2]), math.sin(a[2])
        Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float32)
        Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float32)
        Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float32)
        m = torch.eye(4)
        m[:3, :3] = Rz @ Ry @ Rx
        m[:3, 3] = t
        vms.append(m)
    return torch.stack(vms).contiguous()


def upstream_grads(H, W, seed=2, C=1):
    """Op-level upstream gradients N(0,1) for every rasteriser output."""
    g = torch.Generator().manual_seed(seed)
    return dict(v_render_colors=torch.randn(C, H, W, 3, generator=g), v_render_depths=torch.randn(C, H, W, 1, generator=g),
                v_render_alphas=torch.randn(C, H, W, 1, generator=g), v_render_normals=torch.randn(C, H, W, 3, generator=g),
                v_render_median=torch.randn(C, H, W, 1, generator=g))
