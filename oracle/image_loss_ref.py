"""Restatement of the reference's photometric loss in plain torch (TEST INFRASTRUCTURE ONLY, see oracle/README.md).

The reference computes this loss with libtorch ops, so the restatement is a line-by-line transcription of
  loss_utils::gaussian / create_window / ssim   /root/reference/include/optimizer/loss_utils/loss_utils.cpp:6-21, 71-117
  loss::rgb_loss, loss::dssim_loss              /root/reference/include/optimizer/loss/loss.cpp:22-47
  k_rgb_weight * rgb + k_dssim_weight * dssim   /root/reference/include/neural_mapping/neural_mapping.cpp:237-240
evaluated in float64 on the CPU; gradients come from torch.autograd."""
import math

import torch


def gaussian(window_size=11, sigma=1.5, dtype=torch.float64):
    g = torch.tensor([math.exp(-(math.floor((x - window_size) / 2.0) ** 2) / (2.0 * sigma * sigma)) for x in range(window_size)], dtype=dtype)
    return g / g.sum()


def ssim(img1, img2, window_size=11, channel=3):
    w1 = gaussian(window_size, 1.5, img1.dtype).unsqueeze(1)
    window = (w1 @ w1.t())[None, None].expand(channel, 1, window_size, window_size).contiguous()
    conv = lambda t: torch.nn.functional.conv2d(t, window, padding=window_size // 2, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 * 0.01, 0.03 * 0.03
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


def l1_dssim_loss(render, gt, rgb_weight=0.8, dssim_weight=0.2):
    rgb = (render - gt).abs().mean()
    dssim = 1.0 - ssim(render.permute(2, 0, 1)[None], gt.permute(2, 0, 1)[None])
    return rgb_weight * rgb + dssim_weight * dssim


def depth_to_normal(fx, fy, cx, cy, pose, depth):
    """sensor::depth_to_normal, /root/reference/include/utils/sensor_utils/cameras.hpp:15-29 (zdir, pixel offset 0.5),
    :176-226 (back-projection, central differences, cross product, zero border)."""
    H, W = depth.shape[0], depth.shape[1]
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype) + 0.5, torch.arange(W, dtype=depth.dtype) + 0.5, indexing="ij")
    zdir = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], -1)
    rot, pos = pose[:, :3], pose[:, 3]
    pts = zdir @ rot.t() * depth + pos
    out = torch.zeros_like(pts)
    dx = pts[2:, 1:-1] - pts[:-2, 1:-1]
    dy = pts[1:-1, 2:] - pts[1:-1, :-2]
    out[1:-1, 1:-1] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out


def normal_consistency_loss(fx, fy, cx, cy, pose, depth, alpha, render_normal):
    """/root/reference/include/neural_mapping/neural_mapping.cpp:243-266"""
    a = alpha.detach()
    dn = depth_to_normal(fx, fy, cx, cy, pose, depth) * a
    return (a.square().squeeze(-1) - (dn * render_normal).sum(-1).nan_to_num()).mean()
