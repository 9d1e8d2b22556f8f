// pointfeat.hip -- LaRa's fine-stage point sampler (lightning/network.py:390-411 with projection :182-187):
// projection of the Gaussian centres into the input views + bilinear gather of the 8-channel image stack + depth
// residual, one thread per (point, view) over a packed channel-last copy of the maps; the backward scatters into a
// stack of the same shape with float atomics and sums the coordinate gradient over the views with lane shuffles.
// include/lara_pointfeat.h has the formulas and the contract.
#include "common.h"
#include "../../include/lara_pointfeat.h"

namespace {

constexpr int PF_MAX_VIEWS = 8;

struct PfP {
    int n, V, h, w;
    const float *points, *w2cs, *ixts, *img_ref, *image, *acc, *depth;
    int vtot;   // 0: image / acc / depth are [V,h,w,c]; > 0: [h, vtot*w, c], the views side by side (`render_views(concat=True)`)
};

// pixel i = (v, y, x) of the packed stack -> its pixel index in the render-derived maps
__device__ __forceinline__ size_t map_pixel(const PfP &p, const size_t i) {
    if (p.vtot == 0) return i;
    const size_t hw = (size_t)p.h * p.w, v = i / hw, pix = i - v * hw, y = pix / p.w, x = pix - y * p.w;
    return (y * p.vtot + v) * p.w + x;
}

struct Tap {  // the four bilinear taps of a sample position; weight 0 and a safe index outside the image
    int idx[4];
    float wgt[4];
    float wx, wy;   // fractional offsets
    bool in[4];
};

__device__ __forceinline__ Tap taps(const float x, const float y, const int h, const int w) {
    Tap t;
    const float fx = floorf(x), fy = floorf(y);
    t.wx = x - fx; t.wy = y - fy;
    // (positions far outside the image, inf or nan give taps outside it: every `in` is false)
    const int x0 = (fx >= -2.f && fx <= (float)w + 1.f) ? (int)fx : -2, y0 = (fy >= -2.f && fy <= (float)h + 1.f) ? (int)fy : -2;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
        t.in[k] = xi >= 0 && xi < w && yi >= 0 && yi < h;
        t.idx[k] = t.in[k] ? yi * w + xi : 0;
        t.wgt[k] = ((k & 1) ? t.wx : 1.f - t.wx) * ((k >> 1) ? t.wy : 1.f - t.wy);
    }
    return t;
}

// The four tensors are first packed into one channel-last stack [V, h, w, 8] (32 bytes per pixel: a bilinear tap
// is then two 16-byte loads from one cache line instead of eight 4-byte loads from six different lines -- the
// sampler is bound by the lines it touches, not by arithmetic), and the backward scatters into a stack of the
// same shape that `unpack_grad_kernel` folds into the three gradient tensors.
__global__ void __launch_bounds__(256)
pack_stack_kernel(const PfP p, float4 *__restrict__ packed) {
    const size_t hw = (size_t)p.h * p.w, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)p.V * hw) return;
    const size_t v = i / hw, pix = i - v * hw, m = map_pixel(p, i);
    const float *im = p.image + m * 3;
    packed[2 * i] = make_float4(p.img_ref[(v * 3 + 0) * hw + pix], p.img_ref[(v * 3 + 1) * hw + pix],
                                p.img_ref[(v * 3 + 2) * hw + pix], im[0]);
    packed[2 * i + 1] = make_float4(im[1], im[2], p.acc[m], p.depth[m]);
}

__global__ void __launch_bounds__(256)
unpack_grad_kernel(const PfP p, const float4 *__restrict__ dpacked, const size_t n, float *d_image, float *d_acc, float *d_depth) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = dpacked[2 * i], b = dpacked[2 * i + 1];
    const size_t m = map_pixel(p, i);
    if (d_image) { d_image[3 * m] += a.w; d_image[3 * m + 1] += b.x; d_image[3 * m + 2] += b.y; }
    if (d_acc) d_acc[m] += b.z;
    if (d_depth) d_depth[m] += b.w;
}

// the 8 channels of view v at pixel index `pix`
__device__ __forceinline__ void chan8(const float4 *__restrict__ packed, const size_t hw, const int v, const int pix, float (&c)[8]) {
    const float4 a = packed[2 * ((size_t)v * hw + pix)], b = packed[2 * ((size_t)v * hw + pix) + 1];
    c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
}

struct Proj { float x, y, z, cx, cy, cz; };  // pixel position, depth, camera-space point

__device__ __forceinline__ Proj project(const PfP &p, const int v, const float px, const float py, const float pz) {
    const float *m = p.w2cs + v * 16, *k = p.ixts + v * 9;
    Proj r;
    r.cx = m[0] * px + m[1] * py + m[2] * pz + m[3];
    r.cy = m[4] * px + m[5] * py + m[6] * pz + m[7];
    r.cz = m[8] * px + m[9] * py + m[10] * pz + m[11];
    const float qx = k[0] * r.cx + k[1] * r.cy + k[2] * r.cz, qy = k[3] * r.cx + k[4] * r.cy + k[5] * r.cz;
    r.z = k[6] * r.cx + k[7] * r.cy + k[8] * r.cz;
    r.x = qx / r.z; r.y = qy / r.z;
    return r;
}

// One thread per (point, view): VP (= V rounded up to a power of two) neighbouring lanes share a point, so the
// backward's sum over the views is a lane shuffle.
template <int VP>
__global__ void __launch_bounds__(256)
point_feats_fwd_kernel(const PfP p, const float4 *__restrict__ packed, float *__restrict__ out) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(tid / VP), v = (int)(tid % VP);
    if (i >= p.n || v >= p.V) return;
    const size_t hw = (size_t)p.h * p.w;
    const float px = p.points[3 * (size_t)i], py = p.points[3 * (size_t)i + 1], pz = p.points[3 * (size_t)i + 2];
    const Proj pr = project(p, v, px, py, pz);
    const Tap t = taps(pr.x, pr.y, p.h, p.w);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (t.in[k]) {
            float c[8];
            chan8(packed, hw, v, t.idx[k], c);
#pragma unroll
            for (int ch = 0; ch < 8; ch++) s[ch] += t.wgt[k] * c[ch];
        }
#pragma unroll
    for (int ch = 0; ch < 8; ch++) out[((size_t)v * 8 + ch) * p.n + i] = ch == 7 ? fabsf(s[ch] - pr.z) : s[ch];
}

template <int VP>
__global__ void __launch_bounds__(256)
point_feats_bwd_kernel(const PfP p, const float4 *__restrict__ packed, const float *__restrict__ g_out,
                       float *__restrict__ d_points, float *dpacked) {
    // scatter staging: per thread 4 tap offsets (into the gradient stack, -1 = outside), 4 weights, 8 channel gradients
    __shared__ int s_off[4][64][4];
    __shared__ float s_wgt[4][64][4], s_g[4][64][8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(tid / VP), v = (int)(tid % VP);
    const bool active = i < p.n && v < p.V;
    const size_t hw = (size_t)p.h * p.w;
    float dpx = 0.f, dpy = 0.f, dpz = 0.f;
    if (dpacked && !active) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_off[wave][lane][k] = -1;
    }
    if (active) {
        const float px = p.points[3 * (size_t)i], py = p.points[3 * (size_t)i + 1], pz = p.points[3 * (size_t)i + 2];
        const Proj pr = project(p, v, px, py, pz);
        const Tap t = taps(pr.x, pr.y, p.h, p.w);
        float val[4][8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (t.in[k]) chan8(packed, hw, v, t.idx[k], val[k]);
            else
#pragma unroll
                for (int c = 0; c < 8; c++) val[k][c] = 0.f;
        }
        float gx = 0.f, gy = 0.f, gz = 0.f, gc[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float g = g_out[((size_t)v * 8 + c) * p.n + i];
            if (c == 7) {  // | s - z |
                const float s = t.wgt[0] * val[0][c] + t.wgt[1] * val[1][c] + t.wgt[2] * val[2][c] + t.wgt[3] * val[3][c];
                const float d = s - pr.z, sg = d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f;
                gz -= sg * g;
                g *= sg;
            }
            gc[c] = g;
            // d s / d x, d s / d y (a tap outside the image contributes its zero)
            gx += g * ((val[1][c] - val[0][c]) * (1.f - t.wy) + (val[3][c] - val[2][c]) * t.wy);
            gy += g * ((val[2][c] - val[0][c]) * (1.f - t.wx) + (val[3][c] - val[1][c]) * t.wx);
        }
        // park this thread's scatter work for the wave-cooperative pass below
        if (dpacked) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                s_off[wave][lane][k] = t.in[k] ? (int)(((size_t)v * hw + t.idx[k]) * 8) : -1;
                s_wgt[wave][lane][k] = t.wgt[k];
            }
#pragma unroll
            for (int c = 0; c < 8; c++) s_g[wave][lane][c] = gc[c];
        }
        // (x, y, z) = (qx / qz, qy / qz, qz), q = K p_c, p_c = R p + t
        const float inv = 1.0f / pr.z;
        const float dqx = gx * inv, dqy = gy * inv, dqz = gz - (gx * pr.x + gy * pr.y) * inv;
        const float *m = p.w2cs + v * 16, *k = p.ixts + v * 9;
        const float dcx = k[0] * dqx + k[3] * dqy + k[6] * dqz, dcy = k[1] * dqx + k[4] * dqy + k[7] * dqz,
                    dcz = k[2] * dqx + k[5] * dqy + k[8] * dqz;
        dpx = m[0] * dcx + m[4] * dcy + m[8] * dcz;
        dpy = m[1] * dcx + m[5] * dcy + m[9] * dcz;
        dpz = m[2] * dcx + m[6] * dcy + m[10] * dcz;
    }
    // Scatter, wave-cooperative: 8 lanes = the 8 channels of one (thread, tap), so the atomics of a lane group
    // fall into one 32-byte pixel of the gradient stack and one instruction touches 8 cache lines instead of 64
    // (the L2 works per line: 21 M single-float atomics per launch became 4 M line operations).
    if (dpacked) {
        __builtin_amdgcn_wave_barrier();
        const int ch = lane & 7;
#pragma unroll 4
        for (int r = 0; r < 32; r++) {
            const int item = r * 8 + (lane >> 3), th = item >> 2, k = item & 3;
            const int off = s_off[wave][th][k];
            if (off >= 0 && ch >= 3) atomicAdd(dpacked + off + ch, s_wgt[wave][th][k] * s_g[wave][th][ch]);  // (0-2: input image)
        }
    }
#pragma unroll
    for (int o = 1; o < VP; o <<= 1) {  // sum over the views of the point (neighbouring lanes; fixed order)
        dpx += __shfl_xor(dpx, o, 64); dpy += __shfl_xor(dpy, o, 64); dpz += __shfl_xor(dpz, o, 64);
    }
    if (active && v == 0) {
        d_points[3 * (size_t)i] = dpx; d_points[3 * (size_t)i + 1] = dpy; d_points[3 * (size_t)i + 2] = dpz;
    }
}

template <int VP>
void launch_fwd(const PfP &p, const float4 *packed, float *out, hipStream_t s) {
    const size_t threads = (size_t)p.n * VP;
    hipLaunchKernelGGL((point_feats_fwd_kernel<VP>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p, packed, out);
}
template <int VP>
void launch_bwd(const PfP &p, const float4 *packed, const float *g_out, float *d_points, float *dpacked, hipStream_t s) {
    const size_t threads = (size_t)p.n * VP;
    hipLaunchKernelGGL((point_feats_bwd_kernel<VP>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p, packed, g_out,
                       d_points, dpacked);
}
void launch_pack(const PfP &p, float4 *packed, hipStream_t s) {
    const size_t n = (size_t)p.V * p.h * p.w;
    hipLaunchKernelGGL(pack_stack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, packed);
}

// ---- rows of several row-major fp32 tensors through one index list, one thread per (row, float) --------------------------------
// forward: dst_k[r][:] = src_k[idx[r]][:];  scatter (the backward for UNIQUE indices): dst_k[idx[r]][:] = src_k[r][:] into
// zero-filled dst_k.  The five per-surfel tensors of the fine pass are 3 + 12 + 1 + 2 + 4 = 22 floats per row: torch issues five
// index_select / index_copy_ launches per scene and direction (57 us each at 262 144 rows); this is one.
struct RowsP {
    const float *src[LARA_ROWS_MAX];
    float *dst[LARA_ROWS_MAX];
    int width[LARA_ROWS_MAX], first[LARA_ROWS_MAX + 1];
    const int64_t *idx;
    int n, count;
};
template <bool SCATTER>
__global__ void __launch_bounds__(256) take_rows_kernel(const RowsP p) {
    const int total = p.first[p.count];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= p.n * total) return;
    const int r = t / total, c = t - r * total;
    int k = 0;
    while (k + 1 < p.count && c >= p.first[k + 1]) k++;
    const int w = p.width[k], cc = c - p.first[k];
    const size_t far = (size_t)p.idx[r] * w + cc, near = (size_t)r * w + cc;
    if (SCATTER) p.dst[k][far] = p.src[k][near];
    else p.dst[k][near] = p.src[k][far];
}

// The fine stage's volume-feature rows (network.py:509: `x.unsqueeze(1).expand(-1, K, -1)[mask.view(-1, K)]`): row r of the subset
// reads voxel vox[r]; vox is ascending (the indices of a mask, divided by K), so the rows of one voxel are a RUN of at most K.
// Forward: a gather, one float4 per thread.  Backward: the first row of every run adds the run up (in row order: bit-reproducible,
// no atomics) and writes the voxel's gradient; voxels without a row keep the caller's zeros.
__global__ void __launch_bounds__(256) voxel_rows_fwd_kernel(const int n, const int c4, const long long *__restrict__ vox,
                                                            const float4 *__restrict__ x, float4 *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * c4) return;
    const int r = (int)(t / c4), c = (int)(t - (long long)r * c4);
    out[t] = x[(size_t)vox[r] * c4 + c];
}
__global__ void __launch_bounds__(256) voxel_rows_bwd_kernel(const int n, const int c4, const long long *__restrict__ vox,
                                                            const float4 *__restrict__ g, float4 *__restrict__ dx) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * c4) return;
    const int r = (int)(t / c4), c = (int)(t - (long long)r * c4);
    const long long v = vox[r];
    if (r > 0 && vox[r - 1] == v) return;      // not the first row of its voxel's run
    float4 a = g[t];
    for (int q = r + 1; q < n && vox[q] == v; q++) {
        const float4 b = g[(size_t)q * c4 + c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    dx[(size_t)v * c4 + c] = a;
}

}  // namespace

extern "C" {

int lara_voxel_rows(int32_t n, int32_t width, const int64_t *vox, const float *src, float *dst, int32_t backward, void *stream) {
    if (n < 0 || width <= 0 || (width & 3)) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!vox || !src || !dst) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int c4 = width / 4;
    const unsigned blocks = (unsigned)(((long long)n * c4 + 255) / 256);
    {
        L2D_PROF(backward ? "voxel_rows_bwd" : "voxel_rows_fwd", s);
        if (backward) hipLaunchKernelGGL(voxel_rows_bwd_kernel, dim3(blocks), dim3(256), 0, s, n, c4, (const long long *)vox, (const float4 *)src, (float4 *)dst);
        else hipLaunchKernelGGL(voxel_rows_fwd_kernel, dim3(blocks), dim3(256), 0, s, n, c4, (const long long *)vox, (const float4 *)src, (float4 *)dst);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int64_t lara_point_feats_workspace_bytes(int32_t V, int32_t h, int32_t w) {
    if (V <= 0 || V > PF_MAX_VIEWS || h <= 0 || w <= 0 || (int64_t)V * h * w * 8 >= (1ll << 31)) return LARA2DGS_E_INVALID;  // 32-bit offsets into the stack
    return (int64_t)V * h * w * 32 * 2;  // the packed stack and (backward) its gradient
}

int lara_point_feats_forward(int32_t n, int32_t V, int32_t h, int32_t w, const float *points, const float *w2cs,
                             const float *ixts, const float *img_ref, const float *image, const float *acc_map,
                             const float *depth, float *out, void *workspace, void *stream) {
    return lara_point_feats_forward_concat(n, V, 0, h, w, points, w2cs, ixts, img_ref, image, acc_map, depth, out, workspace, stream);
}

int lara_point_feats_forward_concat(int32_t n, int32_t V, int32_t row_views, int32_t h, int32_t w, const float *points,
                                    const float *w2cs, const float *ixts, const float *img_ref, const float *image,
                                    const float *acc_map, const float *depth, float *out, void *workspace, void *stream) {
    if (row_views < 0 || (row_views > 0 && row_views < V)) return LARA2DGS_E_INVALID;
    if (n < 0 || V <= 0 || V > PF_MAX_VIEWS || h <= 0 || w <= 0 || (int64_t)V * h * w * 8 >= (1ll << 31)) return LARA2DGS_E_INVALID;  // 32-bit offsets into the stack
    if (n == 0) return LARA2DGS_OK;
    if (!points || !w2cs || !ixts || !img_ref || !image || !acc_map || !depth || !out || !workspace) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const PfP p{n, V, h, w, points, w2cs, ixts, img_ref, image, acc_map, depth, row_views};
    float4 *packed = (float4 *)workspace;
    {
        L2D_PROF("point_feats_pack", s);
        launch_pack(p, packed, s);
    }
    {
        L2D_PROF("point_feats_fwd", s);
        if (V == 1) launch_fwd<1>(p, packed, out, s);
        else if (V == 2) launch_fwd<2>(p, packed, out, s);
        else if (V <= 4) launch_fwd<4>(p, packed, out, s);
        else launch_fwd<8>(p, packed, out, s);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_point_feats_backward(int32_t n, int32_t V, int32_t h, int32_t w, const float *points, const float *w2cs,
                              const float *ixts, const float *img_ref, const float *image, const float *acc_map,
                              const float *depth, const float *g_out, float *d_points, float *d_image,
                              float *d_acc_map, float *d_depth, void *workspace, void *stream) {
    return lara_point_feats_backward_concat(n, V, 0, h, w, points, w2cs, ixts, img_ref, image, acc_map, depth, g_out, d_points,
                                            d_image, d_acc_map, d_depth, workspace, stream);
}

int lara_point_feats_backward_concat(int32_t n, int32_t V, int32_t row_views, int32_t h, int32_t w, const float *points,
                                     const float *w2cs, const float *ixts, const float *img_ref, const float *image,
                                     const float *acc_map, const float *depth, const float *g_out, float *d_points,
                                     float *d_image, float *d_acc_map, float *d_depth, void *workspace, void *stream) {
    if (row_views < 0 || (row_views > 0 && row_views < V)) return LARA2DGS_E_INVALID;
    if (n < 0 || V <= 0 || V > PF_MAX_VIEWS || h <= 0 || w <= 0 || (int64_t)V * h * w * 8 >= (1ll << 31)) return LARA2DGS_E_INVALID;  // 32-bit offsets into the stack
    if (n == 0) return LARA2DGS_OK;
    if (!points || !w2cs || !ixts || !img_ref || !image || !acc_map || !depth || !g_out || !d_points || !workspace)
        return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const PfP p{n, V, h, w, points, w2cs, ixts, img_ref, image, acc_map, depth, row_views};
    const size_t npix = (size_t)V * h * w;
    float4 *packed = (float4 *)workspace;
    const bool maps = d_image || d_acc_map || d_depth;
    float *dpacked = maps ? (float *)workspace + npix * 8 : nullptr;
    {
        L2D_PROF("point_feats_pack", s);
        launch_pack(p, packed, s);   // (stateless: the forward's stack is rebuilt rather than kept alive)
        if (maps && hipMemsetAsync(dpacked, 0, npix * 32, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
    {
        L2D_PROF("point_feats_bwd", s);
        if (V == 1) launch_bwd<1>(p, packed, g_out, d_points, dpacked, s);
        else if (V == 2) launch_bwd<2>(p, packed, g_out, d_points, dpacked, s);
        else if (V <= 4) launch_bwd<4>(p, packed, g_out, d_points, dpacked, s);
        else launch_bwd<8>(p, packed, g_out, d_points, dpacked, s);
        if (maps)
            hipLaunchKernelGGL(unpack_grad_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, p, (const float4 *)dpacked, npix,
                               d_image, d_acc_map, d_depth);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}


/* ---- the fine stage's subset rows (network.py:514-524: `x[mask]` of the five per-surfel tensors), one launch for all of them ---- */
int lara_take_rows(int32_t n, const int64_t *idx, int32_t count, const lara_rows_item *items, int32_t scatter, void *stream) {
    if (n < 0 || count < 0 || count > LARA_ROWS_MAX) return LARA2DGS_E_INVALID;
    if (n == 0 || count == 0) return LARA2DGS_OK;
    if (!idx || !items) return LARA2DGS_E_INVALID;
    RowsP p{};
    p.n = n; p.count = count; p.idx = idx;
    int total = 0;
    for (int k = 0; k < count; k++) {
        if (!items[k].src || !items[k].dst || items[k].width <= 0) return LARA2DGS_E_INVALID;
        p.src[k] = items[k].src; p.dst[k] = items[k].dst; p.width[k] = items[k].width; p.first[k] = total;
        total += items[k].width;
    }
    p.first[count] = total;
    if ((int64_t)n * total >= (1ll << 31)) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(((int64_t)n * total + 255) / 256);
    {
        L2D_PROF(scatter ? "take_rows_bwd" : "take_rows_fwd", s);
        if (scatter) hipLaunchKernelGGL(take_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(take_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
