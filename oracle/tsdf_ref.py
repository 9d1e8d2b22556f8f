"""TEST INFRASTRUCTURE (oracle): numpy restatement of the per-voxel TSDF update that Open3D's `ScalableTSDFVolume`
applies (UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier), as used by the reference's
tools/meshExtractor.py:67-110.  Open3D is absent from /root/reference and from this image: PARITY UNPINNED -- this
restates the published algorithm [RECALLED], on a dense grid.  fp32 arithmetic in the same operation order as the kernel."""
import numpy as np


def integrate(res, origin, voxel_length, sdf_trunc, depth, color, intrinsics, extrinsics, depth_trunc, tsdf=None, weight=None, rgb=None):
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
