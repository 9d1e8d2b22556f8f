// tsdf.hip -- dense TSDF fusion of rendered views (include/lara_tsdf.h): thread = voxel, all views of a call folded
// in registers before the voxel's 20 bytes are written back once (a per-view launch would move the volume n_views times).
#include "common.h"
#include "../../include/lara_tsdf.h"

namespace {

__global__ void __launch_bounds__(256)
tsdf_integrate_kernel(const int res, const float ox, const float oy, const float oz, const float vl, const float trunc,
                      const int n_views, const int H, const int W, const float *__restrict__ depth,
                      const float *__restrict__ color, const float *__restrict__ K, const float *__restrict__ E,
                      const float *__restrict__ dtrunc, float *__restrict__ tsdf, float *__restrict__ weight,
                      float *__restrict__ rgb) {
    const int64_t nvox = (int64_t)res * res * res;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nvox) return;
    // consecutive threads walk z (the fastest axis of Open3D's layout): coalesced volume traffic
    const int z = (int)(idx % res), y = (int)((idx / res) % res), x = (int)(idx / ((int64_t)res * res));
    const float px = ox + vl * (0.5f + (float)x), py = oy + vl * (0.5f + (float)y), pz = oz + vl * (0.5f + (float)z);
    float t = tsdf[idx], w = weight[idx], c0 = rgb[3 * idx], c1 = rgb[3 * idx + 1], c2 = rgb[3 * idx + 2];
    const float inv_trunc = 1.0f / trunc, safe_w = (float)W - 0.0001f, safe_h = (float)H - 0.0001f;
    for (int v = 0; v < n_views; v++) {
        const float *e = E + 16 * v, *k = K + 4 * v;   // uniform addresses: scalar loads
        const float cz = e[8] * px + e[9] * py + e[10] * pz + e[11];
        if (!(cz > 0.f)) continue;
        const float cx = e[0] * px + e[1] * py + e[2] * pz + e[3];
        const float cy = e[4] * px + e[5] * py + e[6] * pz + e[7];
        const float uf = cx * k[0] / cz + k[2] + 0.5f, vf = cy * k[1] / cz + k[3] + 0.5f;
        if (!(uf >= 0.0001f && uf < safe_w && vf >= 0.0001f && vf < safe_h)) continue;
        const int u = (int)uf, vv = (int)vf;
        const size_t pix = ((size_t)v * H + vv) * W + u;
        const float d = depth[pix];
        if (!(d > 0.f) || d > dtrunc[v]) continue;
        const float rx = ((float)u - k[2]) / k[0], ry = ((float)vv - k[3]) / k[1];
        const float sdf = (d - cz) * sqrtf(rx * rx + ry * ry + 1.0f);
        if (!(sdf > -trunc)) continue;
        const float tv = fminf(1.0f, sdf * inv_trunc);
        const float inv = 1.0f / (w + 1.0f);
        t = (t * w + tv) * inv;
        c0 = (c0 * w + color[3 * pix]) * inv;
        c1 = (c1 * w + color[3 * pix + 1]) * inv;
        c2 = (c2 * w + color[3 * pix + 2]) * inv;
        w += 1.0f;
    }
    tsdf[idx] = t; weight[idx] = w;
    rgb[3 * idx] = c0; rgb[3 * idx + 1] = c1; rgb[3 * idx + 2] = c2;
}

}  // namespace

extern "C" int lara_tsdf_integrate(int32_t res, const float *origin, float voxel_length, float sdf_trunc, int32_t n_views,
                                   int32_t H, int32_t W, const float *depth, const float *color, const float *intrinsics,
                                   const float *extrinsics, const float *depth_trunc, float *tsdf, float *weight, float *rgb,
                                   void *stream) {
    if (res <= 0 || res > 2048 || n_views < 0 || H <= 0 || W <= 0 || !(voxel_length > 0.f) || !(sdf_trunc > 0.f)) return LARA2DGS_E_INVALID;
    if (!origin || !tsdf || !weight || !rgb) return LARA2DGS_E_INVALID;
    if (n_views == 0) return LARA2DGS_OK;
    if (!depth || !color || !intrinsics || !extrinsics || !depth_trunc) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvox = (int64_t)res * res * res;
    {
        L2D_PROF("tsdf_integrate", s);
        hipLaunchKernelGGL(tsdf_integrate_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, s, res, origin[0], origin[1],
                           origin[2], voxel_length, sdf_trunc, n_views, H, W, depth, color, intrinsics, extrinsics, depth_trunc,
                           tsdf, weight, rgb);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
