"""fp64 autograd restatement of the 2DGS forward -- TEST INFRASTRUCTURE ONLY.

Purpose: an *independent* check of (a) the C oracle's forward arithmetic and (b) its hand-derived
backward (surfel_oracle.c, ``oracle_backward``).  The forward below is written once, in torch
float64, sequential over a tile's sorted splat list and vectorised over the tile's pixels; its
gradients come from ``torch.autograd``, not from any formula.

It consumes the integer stages (tile ranges, sorted lists) from the C oracle so that both walk the
same lists.  Two places where the published backward is *not* the analytic derivative are handled
explicitly (see DESIGN.md "gradient semantics"):
  * alpha = min(0.99, o*G): the published backward passes the gradient through the clamp.  Here
    ``clamp_passthrough=True`` does the same (straight-through), so scenes with o*G > 0.99 still
    agree; with False the clamp has zero gradient.
  * the screen-space low-pass branch's depth gradient quirk is *not* reproduced here: compare
    against ``oracle.backward(..., lowpass_depth_quirk=False)``.
Only small scenes (P <~ 1e3, images <~ 64x64): seconds.
"""
from __future__ import annotations

import torch

NEAR_N, FAR_N = 0.2, 100.0
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
      0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _rotmat(q):
    q = q / q.norm(dim=-1, keepdim=True).detach()  # published VJP ignores the normalisation
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def _sh_rgb(deg, shs, means, campos):
    d = means - campos
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * shs[:, 0]
    if deg > 0:
        r = r - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6] \
            + C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8]
        if deg > 2:
            r = r + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10] \
                + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] \
                + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14] \
                + C3[6] * x * (xx - 3 * yy) * shs[:, 15]
    return torch.clamp_min(r + 0.5, 0.0)


def render(view, means3D, opacities, shs, scales, rotations, ranges, point_list,
           clamp_passthrough: bool = True, colors_precomp=None, tiles=None, lowpass_depth_quirk: bool = False):
    """Differentiable fp64 forward.  Returns (color [3,H,W], allmap [7,H,W]).
    `tiles`: render only these tile ids (the other pixels stay zero) -- the full-size arbitration of
    tools/grad_arbiter.py walks a few dozen tiles of a 512 x 512 frame, not all 1024.
    `lowpass_depth_quirk`: give the screen-space low-pass branch the PUBLISHED backward (dL/dTw += dL_dz * (s.x, s.y, 1)
    although its forward depth is Tw.z) through a straight-through term, so that the result can be held against
    `oracle.backward(..., lowpass_depth_quirk=True)` and the HIP path, which both reproduce the published code."""
    dt = torch.float64
    H, W = int(view.image_height), int(view.image_width)
    vm = torch.as_tensor(view.viewmatrix, dtype=dt).reshape(4, 4)
    pm = torch.as_tensor(view.projmatrix, dtype=dt).reshape(4, 4)
    campos = torch.as_tensor(view.campos, dtype=dt).reshape(3)
    bg = torch.as_tensor(view.bg, dtype=dt).reshape(3)
    P = means3D.shape[0]
    R = _rotmat(rotations)
    sm = float(view.scale_modifier)
    L0 = R[:, :, 0] * (sm * scales[:, 0:1])
    L1 = R[:, :, 1] * (sm * scales[:, 1:2])
    ones = torch.ones(P, 1, dtype=dt)
    zeros = torch.zeros(P, 1, dtype=dt)
    rows = torch.stack([torch.cat([L0, zeros], 1), torch.cat([L1, zeros], 1),
                        torch.cat([means3D, ones], 1)], dim=1)          # [P,3,4]
    hom = rows @ pm                                                       # [P,3,4]  (row-vector)
    Tx = hom[..., 0] * (W / 2.0) + hom[..., 3] * ((W - 1) / 2.0)           # [P,3] = Tu
    Ty = hom[..., 1] * (H / 2.0) + hom[..., 3] * ((H - 1) / 2.0)           # Tv
    Tw = hom[..., 3]                                                      # Tw
    p_view = means3D @ vm[:3, :3] + vm[3, :3]
    normal = R[:, :, 2] @ vm[:3, :3]
    cosv = -(p_view * normal).sum(-1, keepdim=True)
    normal = normal * torch.where(cosv > 0, 1.0, -1.0).detach()
    # centre of the 3-sigma box (feeds the screen-space low-pass)
    t = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
    dist = (t * Tw * Tw).sum(-1, keepdim=True)
    f = t / dist
    cx = (f * Tx * Tw).sum(-1)
    cy = (f * Ty * Tw).sum(-1)
    if colors_precomp is None:
        rgb = _sh_rgb(int(view.sh_degree), shs, means3D, campos)
    else:
        rgb = colors_precomp
    opac = opacities.reshape(-1)

    color = torch.zeros(3, H, W, dtype=dt)
    allmap = torch.zeros(7, H, W, dtype=dt)
    gx = (W + 15) // 16
    gy = (H + 15) // 16
    outs_c, outs_a, coords = [], [], []
    for tile in (range(gx * gy) if tiles is None else tiles):
        tile = int(tile)
        tx, ty = tile % gx, tile // gx
        ys, xs = torch.meshgrid(torch.arange(ty * 16, min(ty * 16 + 16, H)),
                                torch.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
        px = xs.reshape(-1).to(dt)
        py = ys.reshape(-1).to(dt)
        n = px.numel()
        T = torch.ones(n, dtype=dt)
        done = torch.zeros(n, dtype=torch.bool)
        C = torch.zeros(n, 3, dtype=dt)
        N = torch.zeros(n, 3, dtype=dt)
        Dd = torch.zeros(n, dtype=dt)
        M1 = torch.zeros(n, dtype=dt)
        M2 = torch.zeros(n, dtype=dt)
        dist_acc = torch.zeros(n, dtype=dt)
        med = torch.zeros(n, dtype=dt)
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        for i in range(r0, r1):
            g = int(point_list[i])
            k = px[:, None] * Tw[g] - Tx[g]
            l = py[:, None] * Tw[g] - Ty[g]
            p = torch.cross(k, l, dim=-1)
            ok = p[:, 2] != 0
            pz = torch.where(ok, p[:, 2], torch.ones_like(p[:, 2]))
            sx, sy = p[:, 0] / pz, p[:, 1] / pz
            rho3d = sx * sx + sy * sy
            dx, dy = cx[g] - px, cy[g] - py
            rho2d = 2.0 * (dx * dx + dy * dy)
            use3d = rho3d <= rho2d
            rho = torch.where(use3d, rho3d, rho2d)
            flat_depth = Tw[g, 2].expand(n)
            if lowpass_depth_quirk:      # value Tw.z, gradient dL_dz * (s.x, s.y, 1) into Tw (the published backward)
                q = sx.detach() * Tw[g, 0] + sy.detach() * Tw[g, 1]
                flat_depth = flat_depth + (q - q.detach())
            depth = torch.where(use3d, sx * Tw[g, 0] + sy * Tw[g, 1] + Tw[g, 2], flat_depth)
            ok = ok & (depth >= NEAR_N) & (-0.5 * rho <= 0)
            G = torch.exp(-0.5 * rho)
            a_raw = opac[g] * G
            if clamp_passthrough:
                alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
            else:
                alpha = torch.clamp(a_raw, max=0.99)
            ok = ok & (alpha >= 1.0 / 255.0)
            test_T = T * (1 - alpha)
            newly_done = ok & (~done) & (test_T < 1e-4)
            act = ok & (~done) & (~newly_done)
            done = done | newly_done
            if not bool(act.any()):
                continue
            w = torch.where(act, alpha * T, torch.zeros_like(T))
            A = 1 - T
            safe_depth = torch.where(act, depth, torch.ones_like(depth))
            m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / safe_depth)
            dist_acc = dist_acc + (m * m * A + M2 - 2 * m * M1) * w
            Dd = Dd + safe_depth * w
            M1 = M1 + m * w
            M2 = M2 + m * m * w
            med = torch.where(act & (T > 0.5), depth, med)
            N = N + normal[g] * w[:, None]
            C = C + rgb[g] * w[:, None]
            T = torch.where(act, test_T, T)
        outs_c.append(C + T[:, None] * bg)
        outs_a.append(torch.stack([Dd, 1 - T, N[:, 0], N[:, 1], N[:, 2], med, dist_acc], dim=-1))
        coords.append((ys.reshape(-1), xs.reshape(-1)))
    yy = torch.cat([c[0] for c in coords])
    xx = torch.cat([c[1] for c in coords])
    cc = torch.cat(outs_c)   # [HW,3]
    aa = torch.cat(outs_a)   # [HW,7]
    color[:, yy, xx] = cc.t()
    allmap[:, yy, xx] = aa.t()
    return color, allmap
