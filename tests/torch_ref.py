"""Independent torch (fp64, autograd) restatement of the splat operators.

Used ONLY to validate the hand-derived VJPs of the C oracle (tests/test_oracle_selfcheck.py):
the forward is written in the most direct vectorised form and the gradients come from
torch.autograd, so an error in the oracle's backward formulas cannot hide here.
Formulas: DESIGN.md section SPEC (A.1-A.5).
"""
import torch


def quat_to_rotmat(qn):
    w, x, y, z = qn.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def project(means, quats, scales, viewmat, K, eps_uv=None):
    """Single camera, no culling: returns means2d, depths, ray_transforms, normals, samples."""
    R, t = viewmat[:3, :3], viewmat[:3, 3]
    mc = means @ R.T + t
    qn = quats / quats.norm(dim=-1, keepdim=True)
    Rq = quat_to_rotmat(qn)
    Rc = R @ Rq
    Hm = torch.stack([scales[:, 0:1] * Rc[:, :, 0], scales[:, 1:2] * Rc[:, :, 1], mc], -1)
    Wm = K @ Hm
    Mu, Mv, Mw = Wm[:, 0], Wm[:, 1], Wm[:, 2]
    tt = torch.tensor([1.0, 1.0, -1.0], dtype=means.dtype)
    d = (tt * Mw * Mw).sum(-1, keepdim=True)
    f = tt / d
    m2d = torch.stack([(f * Mu * Mw).sum(-1), (f * Mv * Mw).sum(-1)], -1)
    normal = Rc[:, :, 2]
    flip = torch.where((-(normal * mc).sum(-1)) > 0, 1.0, -1.0).to(means.dtype)
    normal = normal * flip[:, None]
    samples = means
    if eps_uv is not None:
        samples = means + (scales[:, 0:1] * eps_uv[:, 0:1]) * Rq[:, :, 0] + (scales[:, 1:2] * eps_uv[:, 1:2]) * Rq[:, :, 1]
    return m2d, mc[:, 2], Wm, normal, samples


_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]


def sh_colors(deg, dirs, coeffs):
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    b = [torch.full_like(x, _C0)]
    if deg >= 1:
        b += [-_C1 * y, _C1 * z, -_C1 * x]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_C2[0] * xy, _C2[1] * yz, _C2[2] * (2 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy)]
    if deg >= 3:
        b += [_C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy),
              _C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy),
              _C3[6] * x * (xx - 3 * yy)]
    B = torch.stack(b, -1)
    return torch.clamp_min((B[:, :, None] * coeffs[:, :B.shape[1]]).sum(1) + 0.5, 0.0)


def rasterize(means2d, ray_transforms, colors, opacities, normals, W, H, tile, offsets, flatten_ids, background=None):
    """Single camera.  offsets [th,tw] int, flatten_ids [I] int (from the binning stage)."""
    dt = means2d.dtype
    th, tw = offsets.shape
    I = flatten_ids.shape[0]
    offs = offsets.reshape(-1).tolist() + [I]
    rc = torch.zeros(H, W, 3, dtype=dt); rn = torch.zeros(H, W, 3, dtype=dt)
    rd = torch.zeros(H, W, 1, dtype=dt); ra = torch.zeros(H, W, 1, dtype=dt); rm = torch.zeros(H, W, 1, dtype=dt)
    out = []
    for t in range(th * tw):
        ty, tx = divmod(t, tw)
        ids = flatten_ids[offs[t]:offs[t + 1]].long()
        ys = torch.arange(ty * tile, min((ty + 1) * tile, H))
        xs = torch.arange(tx * tile, min((tx + 1) * tile, W))
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        px = (gx.reshape(-1).to(dt) + 0.5)[:, None]
        py = (gy.reshape(-1).to(dt) + 0.5)[:, None]
        P = px.shape[0]
        if ids.numel() == 0:
            col = torch.zeros(P, 3, dtype=dt); nrm = torch.zeros(P, 3, dtype=dt)
            dep = torch.zeros(P, 1, dtype=dt); alp = torch.zeros(P, 1, dtype=dt); med = torch.zeros(P, 1, dtype=dt)
        else:
            M9 = ray_transforms[ids]
            Mu, Mv, Mw = M9[:, 0][None], M9[:, 1][None], M9[:, 2][None]       # [1,L,3]
            hu = px[:, :, None] * Mw - Mu
            hv = py[:, :, None] * Mw - Mv
            z = torch.cross(hu, hv, dim=-1)
            zz = torch.where(z[..., 2] == 0, torch.ones_like(z[..., 2]), z[..., 2])
            sx, sy = z[..., 0] / zz, z[..., 1] / zz
            g3 = sx * sx + sy * sy
            dx = means2d[ids][None, :, 0] - px
            dy = means2d[ids][None, :, 1] - py
            g2 = 2.0 * (dx * dx + dy * dy)
            b3 = g3 <= g2
            sigma = 0.5 * torch.where(b3, g3, g2)
            alpha = torch.clamp_max(opacities[ids][None] * torch.exp(-sigma), 0.999)
            valid = (z[..., 2] != 0) & (sigma >= 0) & (alpha >= 1.0 / 255.0)
            a = torch.where(valid, alpha, torch.zeros_like(alpha))
            Tb = torch.cumprod(torch.cat([torch.ones(P, 1, dtype=dt), (1 - a)[:, :-1]], 1), 1)   # T before
            nT = Tb * (1 - a)
            stop = (torch.cummax(((nT <= 1e-4) & valid).to(torch.int8), 1).values > 0)
            contrib = valid & ~stop
            w = torch.where(contrib, a * Tb, torch.zeros_like(a))
            dpt = torch.where(b3, sx * Mw[..., 0] + sy * Mw[..., 1] + Mw[..., 2], Mw[..., 2].expand_as(sx))
            col = w @ colors[ids]; nrm = w @ normals[ids]
            dep = (w * dpt).sum(1, keepdim=True)
            alp = w.sum(1, keepdim=True)
            mm = contrib & (Tb > 0.5)
            L = ids.numel()
            pos = torch.where(mm, torch.arange(L)[None].expand(P, L), torch.full((P, L), -1))
            midx = pos.max(1).values
            med = torch.where(midx >= 0, dpt.gather(1, midx.clamp_min(0)[:, None])[:, 0], torch.zeros(P, dtype=dt))[:, None]
            if background is not None:
                col = col + (1 - alp) * background[None]
        out.append((gy.reshape(-1), gx.reshape(-1), col, nrm, dep, alp, med))
    ys = torch.cat([o[0] for o in out]); xs = torch.cat([o[1] for o in out])
    rc = rc.index_put((ys, xs), torch.cat([o[2] for o in out]))
    rn = rn.index_put((ys, xs), torch.cat([o[3] for o in out]))
    rd = rd.index_put((ys, xs), torch.cat([o[4] for o in out]))
    ra = ra.index_put((ys, xs), torch.cat([o[5] for o in out]))
    rm = rm.index_put((ys, xs), torch.cat([o[6] for o in out]))
    return rc, rd, ra, rn, rm
