"""TEST INFRASTRUCTURE (oracle): numpy restatement of the per-voxel TSDF update that Open3D's `ScalableTSDFVolume`
applies (UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier), as used by the reference's
tools/meshExtractor.py:67-110.  Open3D is absent from /root/reference and from this image: PARITY UNPINNED -- this
restates the published algorithm [RECALLED], on a dense grid.  fp32 arithmetic in the same operation order as the kernel."""
import numpy as np


def integrate(res, origin, voxel_length, sdf_trunc, depth, color, intrinsics, extrinsics, depth_trunc, tsdf=None, weight=None, rgb=None,
              touched=None):
    """`touched` [n_views, nb, nb, nb] bool (from `touched_blocks`): view v only updates voxels of blocks it touched --
    ScalableTSDFVolume's semantics; None: every voxel (UniformTSDFVolume)."""
    f = np.float32
    n_views, H, W = depth.shape
    if tsdf is None:
        tsdf, weight, rgb = np.zeros(res ** 3, f), np.zeros(res ** 3, f), np.zeros((res ** 3, 3), f)
    idx = np.arange(res ** 3)
    z, y, x = idx % res, (idx // res) % res, idx // (res * res)
    vl = f(voxel_length)
    px = f(origin[0]) + vl * (f(0.5) + x.astype(f))
    py = f(origin[1]) + vl * (f(0.5) + y.astype(f))
    pz = f(origin[2]) + vl * (f(0.5) + z.astype(f))
    trunc, inv_trunc = f(sdf_trunc), f(1.0) / f(sdf_trunc)
    for v in range(n_views):
        e, k = extrinsics[v].astype(f).reshape(16), intrinsics[v].astype(f)
        cz = ((e[8] * px + e[9] * py) + e[10] * pz) + e[11]
        cx = ((e[0] * px + e[1] * py) + e[2] * pz) + e[3]
        cy = ((e[4] * px + e[5] * py) + e[6] * pz) + e[7]
        ok = cz > 0
        if touched is not None:
            ok &= touched[v][x // 16, y // 16, z // 16]
        with np.errstate(divide="ignore", invalid="ignore"):
            uf = (cx * k[0] / cz + k[2]) + f(0.5)
            vf = (cy * k[1] / cz + k[3]) + f(0.5)
        ok &= (uf >= f(0.0001)) & (uf < f(W) - f(0.0001)) & (vf >= f(0.0001)) & (vf < f(H) - f(0.0001))
        u = np.where(ok, uf, 0).astype(np.int64)
        vv = np.where(ok, vf, 0).astype(np.int64)
        d = depth[v][vv, u].astype(f)
        ok &= (d > 0) & ~(d > f(depth_trunc[v]))
        rx, ry = (u.astype(f) - k[2]) / k[0], (vv.astype(f) - k[3]) / k[1]
        sdf = (d - cz) * np.sqrt((rx * rx + ry * ry) + f(1.0)).astype(f)
        ok &= sdf > -trunc
        tv = np.minimum(f(1.0), sdf * inv_trunc)
        inv = f(1.0) / (weight + f(1.0))
        tsdf = np.where(ok, (tsdf * weight + tv) * inv, tsdf).astype(f)
        col = color[v][vv, u].astype(f)
        rgb = np.where(ok[:, None], (rgb * weight[:, None] + col) * inv[:, None], rgb).astype(f)
        weight = np.where(ok, weight + f(1.0), weight).astype(f)
    return tsdf, weight, rgb


def touched_blocks(res, origin, voxel_length, sdf_trunc, depth, intrinsics, extrinsics, depth_trunc, stride=4):
    """ScalableTSDFVolume::Integrate's unit selection [RECALLED]: every `stride`-th pixel with a valid depth is back-projected
    (PointCloud::CreateFromDepthImage with the depth sampling stride) and the 16^3 volume units within +- sdf_trunc of the
    point, per axis, are opened / marked for this view.  fp32, the kernel's operation order."""
    f = np.float32
    n_views, H, W = depth.shape
    nb = res // 16
    out = np.zeros((n_views, nb, nb, nb), dtype=bool)
    inv = f(1.0) / (f(voxel_length) * f(16.0))
    for v in range(n_views):
        k = intrinsics[v].astype(f)
        m = np.linalg.inv(extrinsics[v].astype(np.float64)).astype(f)
        ii, jj = np.meshgrid(np.arange(0, H, stride), np.arange(0, W, stride), indexing="ij")
        d = depth[v][ii, jj].astype(f)
        ok = (d > 0) & ~(d > f(depth_trunc[v]))
        cx = (jj.astype(f) - k[2]) * d / k[0]
        cy = (ii.astype(f) - k[3]) * d / k[1]
        w = [((m[a, 0] * cx + m[a, 1] * cy) + m[a, 2] * d) + m[a, 3] - f(origin[a]) for a in range(3)]
        lo = [np.floor((w[a] - f(sdf_trunc)) * inv).astype(np.int64) for a in range(3)]
        hi = [np.floor((w[a] + f(sdf_trunc)) * inv).astype(np.int64) for a in range(3)]
        for a in range(3):
            ok &= (hi[a] >= 0) & (lo[a] < nb)
        for idx in np.argwhere(ok):
            sl = tuple(slice(max(int(lo[a][tuple(idx)]), 0), min(int(hi[a][tuple(idx)]), nb - 1) + 1) for a in range(3))
            out[v][sl] = True
    return out


def extract_mesh(res, origin, voxel_length, tsdf, weight, rgb):
    """Marching cubes as include/lara_tsdf.h describes it, cell by cell in Python (small volumes only): returns
    (vertices [T,3,3], colors [T,3,3], edge_keys [T,3]) in cell order (x, then y, then z), the kernel's order."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_mc_tables as g
    table = g.build()
    f = np.float32
    T3, W3, C3 = tsdf.reshape(res, res, res), weight.reshape(res, res, res), rgb.reshape(res, res, res, 3)
    verts, cols, keys = [], [], []
    for x in range(res - 1):
        for y in range(res - 1):
            for z in range(res - 1):
                corner = [(x + dx, y + dy, z + dz) for dx, dy, dz in g.CORNERS]
                if any(not W3[c] > 0 for c in corner):
                    continue
                fv = [T3[c] for c in corner]
                case = sum((1 << i) for i in range(8) if fv[i] < 0)
                for tri in table[case]:
                    tv, tc, tk = [], [], []
                    for e in tri:
                        a, b = g.EDGE_CORNERS[e]
                        w = f(fv[a]) / (f(fv[a]) - f(fv[b]))
                        pa, pb = np.array(corner[a]), np.array(corner[b])
                        tv.append([f(origin[i]) + f(voxel_length) * (f(0.5) + f(pa[i]) + w * f(pb[i] - pa[i])) for i in range(3)])
                        tc.append(((f(1.0) - w) * C3[corner[a]] + w * C3[corner[b]]) * (f(1.0) / f(255.0)))
                        tk.append(((pa[0] * res + pa[1]) * res + pa[2]) * 3 + int(np.argmax(pb - pa)))
                    verts.append(tv); cols.append(tc); keys.append(tk)
    return (np.array(verts, f).reshape(-1, 3, 3), np.array(cols, f).reshape(-1, 3, 3), np.array(keys, np.int64).reshape(-1, 3))
