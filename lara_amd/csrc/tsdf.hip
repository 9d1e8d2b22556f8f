// tsdf.hip -- dense TSDF fusion of rendered views (include/lara_tsdf.h): thread = voxel, all views of a call folded
// in registers before the voxel's 20 bytes are written back once (a per-view launch would move the volume n_views times).
#include "common.h"
#include "../../include/lara_tsdf.h"

namespace {

struct TsdfP {
    int res;
    float ox, oy, oz, vl, trunc;
    int n_views, H, W;
    const float *depth, *color, *K, *E, *dtrunc;
    float *tsdf, *weight, *rgb;
};

// one voxel through the views of a call, in order (include/lara_tsdf.h); `mask`: bit v clear = view v skips this voxel
// (the block-sparse path: the view never touched the voxel's block).  The same code serves the dense and the block-sparse
// kernel, so where both integrate a view they produce the same bits.
__device__ __forceinline__ void integrate_voxel(const TsdfP &p, const int x, const int y, const int z, const uint64_t mask) {
    const int64_t idx = ((int64_t)x * p.res + y) * p.res + z;
    const float px = p.ox + p.vl * (0.5f + (float)x), py = p.oy + p.vl * (0.5f + (float)y), pz = p.oz + p.vl * (0.5f + (float)z);
    float t = p.tsdf[idx], w = p.weight[idx], c0 = p.rgb[3 * idx], c1 = p.rgb[3 * idx + 1], c2 = p.rgb[3 * idx + 2];
    const float inv_trunc = 1.0f / p.trunc, safe_w = (float)p.W - 0.0001f, safe_h = (float)p.H - 0.0001f;
    for (int v = 0; v < p.n_views; v++) {
        if (!((mask >> v) & 1ull)) continue;
        const float *e = p.E + 16 * v, *k = p.K + 4 * v;   // uniform addresses: scalar loads
        const float cz = e[8] * px + e[9] * py + e[10] * pz + e[11];
        if (!(cz > 0.f)) continue;
        const float cx = e[0] * px + e[1] * py + e[2] * pz + e[3];
        const float cy = e[4] * px + e[5] * py + e[6] * pz + e[7];
        const float uf = cx * k[0] / cz + k[2] + 0.5f, vf = cy * k[1] / cz + k[3] + 0.5f;
        if (!(uf >= 0.0001f && uf < safe_w && vf >= 0.0001f && vf < safe_h)) continue;
        const int u = (int)uf, vv = (int)vf;
        const size_t pix = ((size_t)v * p.H + vv) * p.W + u;
        const float d = p.depth[pix];
        if (!(d > 0.f) || d > p.dtrunc[v]) continue;
        const float rx = ((float)u - k[2]) / k[0], ry = ((float)vv - k[3]) / k[1];
        const float sdf = (d - cz) * sqrtf(rx * rx + ry * ry + 1.0f);
        if (!(sdf > -p.trunc)) continue;
        const float tv = fminf(1.0f, sdf * inv_trunc);
        const float inv = 1.0f / (w + 1.0f);
        t = (t * w + tv) * inv;
        c0 = (c0 * w + p.color[3 * pix]) * inv;
        c1 = (c1 * w + p.color[3 * pix + 1]) * inv;
        c2 = (c2 * w + p.color[3 * pix + 2]) * inv;
        w += 1.0f;
    }
    p.tsdf[idx] = t; p.weight[idx] = w;
    p.rgb[3 * idx] = c0; p.rgb[3 * idx + 1] = c1; p.rgb[3 * idx + 2] = c2;
}

__global__ void __launch_bounds__(256) tsdf_integrate_kernel(const TsdfP p) {
    const int64_t nvox = (int64_t)p.res * p.res * p.res;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nvox) return;
    // consecutive threads walk z (the fastest axis of Open3D's layout): coalesced volume traffic
    integrate_voxel(p, (int)(idx / ((int64_t)p.res * p.res)), (int)((idx / p.res) % p.res), (int)(idx % p.res), ~0ull);
}

// ---- block-sparse integration: Open3D's ScalableTSDFVolume (what tools/meshExtractor.py:67 instantiates) -------------------
constexpr int TB = 16;   // voxels per block edge: ScalableTSDFVolume's default volume_unit_resolution

// (1) which 16^3 blocks does each view touch?  ScalableTSDFVolume::Integrate back-projects every `stride`-th pixel (depth
// sampling stride, default 4) and opens the volume units within +- sdf_trunc of the point; only those units integrate the view.
__global__ void __launch_bounds__(256)
tsdf_touch_kernel(const TsdfP p, const int stride, const float *__restrict__ c2w, uint8_t *__restrict__ touched) {
    const int v = blockIdx.y, ny = (p.H + stride - 1) / stride, nx = (p.W + stride - 1) / stride;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= ny * nx) return;
    const int i = (s / nx) * stride, j = (s % nx) * stride;
    const float d = p.depth[((size_t)v * p.H + i) * p.W + j];
    if (!(d > 0.f) || d > p.dtrunc[v]) return;
    const float *k = p.K + 4 * v, *m = c2w + 16 * v;
    const float cx = ((float)j - k[2]) * d / k[0], cy = ((float)i - k[3]) * d / k[1];
    const float wx = m[0] * cx + m[1] * cy + m[2] * d + m[3], wy = m[4] * cx + m[5] * cy + m[6] * d + m[7],
                wz = m[8] * cx + m[9] * cy + m[10] * d + m[11];
    const int nb = p.res / TB;
    const float inv = 1.0f / (p.vl * (float)TB);
    int lo[3], hi[3];
    const float w3[3] = {wx - p.ox, wy - p.oy, wz - p.oz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = (int)floorf((w3[a] - p.trunc) * inv);
        hi[a] = (int)floorf((w3[a] + p.trunc) * inv);
        if (hi[a] < 0 || lo[a] >= nb) return;       // outside the volume's domain
        lo[a] = lo[a] < 0 ? 0 : lo[a];
        hi[a] = hi[a] >= nb ? nb - 1 : hi[a];
    }
    uint8_t *t = touched + (size_t)v * nb * nb * nb;
    for (int bx = lo[0]; bx <= hi[0]; bx++)
        for (int by = lo[1]; by <= hi[1]; by++)
            for (int bz = lo[2]; bz <= hi[2]; bz++) t[((size_t)bx * nb + by) * nb + bz] = 1;   // (racing stores of the same value)
}

// (2) every touched block folds the views that touched it; one workgroup = 256 of a block's 4096 voxels, z fastest
__global__ void __launch_bounds__(256)
tsdf_integrate_blocks_kernel(const TsdfP p, const uint8_t *__restrict__ touched, uint8_t *__restrict__ allocated) {
    const int nb = p.res / TB, block = blockIdx.x / (TB * TB * TB / 256), part = blockIdx.x % (TB * TB * TB / 256);
    const size_t nb3 = (size_t)nb * nb * nb;
    uint64_t mask = 0;
    for (int v = 0; v < p.n_views; v++) mask |= (uint64_t)(touched[(size_t)v * nb3 + block] != 0) << v;   // uniform: scalar loads
    if (mask == 0) return;
    if (part == 0 && threadIdx.x == 0) allocated[block] = 1;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    const int l = part * 256 + threadIdx.x, lz = l % TB, ly = (l / TB) % TB, lx = l / (TB * TB);
    integrate_voxel(p, bx * TB + lx, by * TB + ly, bz * TB + lz, mask);
}

// ---- (3) mesh extraction: marching cubes over the cells whose 8 corners were all observed (weight > 0), as
// ScalableTSDFVolume::ExtractTriangleMesh does; corner = voxel centre; vertex = linear zero crossing on a cell edge --------
#include "mc_tables.h"
__constant__ uint8_t c_mc_ntri[256];
__constant__ uint8_t c_mc_tri[256][15];
__constant__ uint8_t c_mc_edge[12][2];

struct McP {
    int res;
    float ox, oy, oz, vl;
    const float *tsdf, *weight, *rgb;
};

__device__ __forceinline__ int mc_case(const McP &p, const int x, const int y, const int z, float f[8]) {
    if (x >= p.res - 1 || y >= p.res - 1 || z >= p.res - 1) return 0;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int64_t idx = ((int64_t)(x + (i & 1)) * p.res + (y + ((i >> 1) & 1))) * p.res + (z + ((i >> 2) & 1));
        if (!(p.weight[idx] > 0.f)) return 0;
        f[i] = p.tsdf[idx];
        c |= (f[i] < 0.f) << i;
    }
    return c;
}

__global__ void __launch_bounds__(256)
mc_count_kernel(const McP p, const uint8_t *__restrict__ allocated, int32_t *__restrict__ counts) {
    const int nb = p.res / TB, block = blockIdx.x / (TB * TB * TB / 256), part = blockIdx.x % (TB * TB * TB / 256);
    if (allocated && !allocated[block]) return;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    const int l = part * 256 + threadIdx.x, lz = l % TB, ly = (l / TB) % TB, lx = l / (TB * TB);
    const int x = bx * TB + lx, y = by * TB + ly, z = bz * TB + lz;
    float f[8];
    counts[((int64_t)x * p.res + y) * p.res + z] = c_mc_ntri[mc_case(p, x, y, z, f)];
}

__global__ void __launch_bounds__(256)
mc_emit_kernel(const McP p, const uint8_t *__restrict__ allocated, const int32_t *__restrict__ counts,
               const int64_t *__restrict__ ends, float *__restrict__ verts, float *__restrict__ colors, int64_t *__restrict__ keys) {
    const int nb = p.res / TB, block = blockIdx.x / (TB * TB * TB / 256), part = blockIdx.x % (TB * TB * TB / 256);
    if (allocated && !allocated[block]) return;
    const int bz = block % nb, by = (block / nb) % nb, bx = block / (nb * nb);
    const int l = part * 256 + threadIdx.x, lz = l % TB, ly = (l / TB) % TB, lx = l / (TB * TB);
    const int x = bx * TB + lx, y = by * TB + ly, z = bz * TB + lz;
    const int64_t cell = ((int64_t)x * p.res + y) * p.res + z;
    const int n = counts[cell];
    if (n == 0) return;
    float f[8];
    const int c = mc_case(p, x, y, z, f);
    int64_t tri = ends[cell] - n;      // `ends` = inclusive prefix sum of counts
    for (int t = 0; t < n; t++, tri++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = c_mc_tri[c][3 * t + k], a = c_mc_edge[e][0], b = c_mc_edge[e][1];
            const float w = f[a] / (f[a] - f[b]);       // f[a], f[b] on opposite sides of zero
            const int ax = x + (a & 1), ay = y + ((a >> 1) & 1), az = z + ((a >> 2) & 1);
            const int bxx = x + (b & 1), byy = y + ((b >> 1) & 1), bzz = z + ((b >> 2) & 1);
            float *o = verts + (tri * 3 + k) * 3;
            o[0] = p.ox + p.vl * (0.5f + (float)ax + w * (float)(bxx - ax));
            o[1] = p.oy + p.vl * (0.5f + (float)ay + w * (float)(byy - ay));
            o[2] = p.oz + p.vl * (0.5f + (float)az + w * (float)(bzz - az));
            const int64_t ia = ((int64_t)ax * p.res + ay) * p.res + az, ib = ((int64_t)bxx * p.res + byy) * p.res + bzz;
            float *col = colors + (tri * 3 + k) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) col[ch] = ((1.0f - w) * p.rgb[3 * ia + ch] + w * p.rgb[3 * ib + ch]) * (1.0f / 255.0f);
            // the grid edge the vertex sits on (a < b: a is its lower corner): equal keys = the same vertex
            keys[tri * 3 + k] = ia * 3 + (bxx != ax ? 0 : (byy != ay ? 1 : 2));
        }
    }
}

}  // namespace

static bool mc_tables_loaded[64] = {};   // per device: constant memory is per device

extern "C" {

int lara_tsdf_integrate(int32_t res, const float *origin, float voxel_length, float sdf_trunc, int32_t n_views,
                        int32_t H, int32_t W, const float *depth, const float *color, const float *intrinsics,
                        const float *extrinsics, const float *depth_trunc, float *tsdf, float *weight, float *rgb,
                        void *stream) {
    if (res <= 0 || res > 2048 || n_views < 0 || H <= 0 || W <= 0 || !(voxel_length > 0.f) || !(sdf_trunc > 0.f)) return LARA2DGS_E_INVALID;
    if (!origin || !tsdf || !weight || !rgb) return LARA2DGS_E_INVALID;
    if (n_views == 0) return LARA2DGS_OK;
    if (n_views > 64 || !depth || !color || !intrinsics || !extrinsics || !depth_trunc) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvox = (int64_t)res * res * res;
    const TsdfP p{res, origin[0], origin[1], origin[2], voxel_length, sdf_trunc, n_views, H, W, depth, color, intrinsics, extrinsics,
                  depth_trunc, tsdf, weight, rgb};
    {
        L2D_PROF("tsdf_integrate", s);
        hipLaunchKernelGGL(tsdf_integrate_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_tsdf_integrate_blocks(int32_t res, const float *origin, float voxel_length, float sdf_trunc, int32_t n_views,
                               int32_t H, int32_t W, int32_t depth_sampling_stride, const float *depth, const float *color,
                               const float *intrinsics, const float *extrinsics, const float *cam_to_world,
                               const float *depth_trunc, float *tsdf, float *weight, float *rgb, uint8_t *touched,
                               uint8_t *allocated, void *stream) {
    if (res <= 0 || res > 2048 || res % TB || n_views < 0 || H <= 0 || W <= 0 || depth_sampling_stride <= 0 || !(voxel_length > 0.f) ||
        !(sdf_trunc > 0.f))
        return LARA2DGS_E_INVALID;
    if (!origin || !tsdf || !weight || !rgb || !touched || !allocated) return LARA2DGS_E_INVALID;
    if (n_views == 0) return LARA2DGS_OK;
    if (n_views > 64 || !depth || !color || !intrinsics || !extrinsics || !cam_to_world || !depth_trunc) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int nb = res / TB;
    const size_t nb3 = (size_t)nb * nb * nb;
    const TsdfP p{res, origin[0], origin[1], origin[2], voxel_length, sdf_trunc, n_views, H, W, depth, color, intrinsics, extrinsics,
                  depth_trunc, tsdf, weight, rgb};
    if (hipMemsetAsync(touched, 0, nb3 * n_views, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    {
        L2D_PROF("tsdf_touch", s);
        const int samples = ((H + depth_sampling_stride - 1) / depth_sampling_stride) * ((W + depth_sampling_stride - 1) / depth_sampling_stride);
        hipLaunchKernelGGL(tsdf_touch_kernel, dim3((unsigned)((samples + 255) / 256), (unsigned)n_views), dim3(256), 0, s, p,
                           depth_sampling_stride, cam_to_world, touched);
    }
    {
        L2D_PROF("tsdf_integrate_blocks", s);
        hipLaunchKernelGGL(tsdf_integrate_blocks_kernel, dim3((unsigned)(nb3 * (TB * TB * TB / 256))), dim3(256), 0, s, p, touched, allocated);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

static int mc_params(int32_t res, const float *origin, float voxel_length, const float *tsdf, const float *weight, const float *rgb,
                     McP *p) {
    if (res <= 0 || res > 2048 || res % TB || !(voxel_length > 0.f) || !origin || !tsdf || !weight || !rgb) return LARA2DGS_E_INVALID;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LARA2DGS_E_LAUNCH;
    if (!mc_tables_loaded[dev]) {
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_mc_ntri), MC_NTRI, sizeof(MC_NTRI)) != hipSuccess ||
            hipMemcpyToSymbol(HIP_SYMBOL(c_mc_tri), MC_TRI, sizeof(MC_TRI)) != hipSuccess ||
            hipMemcpyToSymbol(HIP_SYMBOL(c_mc_edge), MC_EDGE_CORNERS, sizeof(MC_EDGE_CORNERS)) != hipSuccess)
            return LARA2DGS_E_LAUNCH;
        mc_tables_loaded[dev] = true;
    }
    *p = McP{res, origin[0], origin[1], origin[2], voxel_length, tsdf, weight, rgb};
    return LARA2DGS_OK;
}

int lara_tsdf_mesh_count(int32_t res, const float *origin, float voxel_length, const float *tsdf, const float *weight,
                         const float *rgb, const uint8_t *allocated, int32_t *counts, void *stream) {
    McP p;
    const int rc = mc_params(res, origin, voxel_length, tsdf, weight, rgb, &p);
    if (rc) return rc;
    if (!counts) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int nb = res / TB;
    {
        L2D_PROF("tsdf_mesh_count", s);
        hipLaunchKernelGGL(mc_count_kernel, dim3((unsigned)((size_t)nb * nb * nb * (TB * TB * TB / 256))), dim3(256), 0, s, p, allocated, counts);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_tsdf_mesh_emit(int32_t res, const float *origin, float voxel_length, const float *tsdf, const float *weight,
                        const float *rgb, const uint8_t *allocated, const int32_t *counts, const int64_t *ends, float *vertices,
                        float *colors, int64_t *edge_keys, void *stream) {
    McP p;
    const int rc = mc_params(res, origin, voxel_length, tsdf, weight, rgb, &p);
    if (rc) return rc;
    if (!counts || !ends || !vertices || !colors || !edge_keys) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int nb = res / TB;
    {
        L2D_PROF("tsdf_mesh_emit", s);
        hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)((size_t)nb * nb * nb * (TB * TB * TB / 256))), dim3(256), 0, s, p, allocated, counts,
                           ends, vertices, colors, edge_keys);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
