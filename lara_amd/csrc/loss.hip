// loss.hip -- the pixel terms of LaRa's loss (lightning/loss.py:28-58 minus MS-SSIM), one streaming pass per direction;
// interface and formulas in include/lara_loss.h.  HBM-bound: 68 bytes per pixel forward, 56 in + 56 out backward.
#include "common.h"
#include "../../include/lara_loss.h"

namespace {

constexpr int LS_PIX = 1024;   // pixels per workgroup (4 per thread)

struct LsP {
    int B, V, H, W;
    long long n;   // pixels
    const float *tar, *image, *image_fine, *rend_dist, *rend_normal, *depth_normal, *acc;
};

// pixel i of the [B, H, V*W] maps -> its pixel index in tar_rgb [B, V, H, W]
__device__ __forceinline__ long long tar_pixel(const LsP &p, const long long i) {
    const long long vw = (long long)p.V * p.W, row = i / vw, col = i - row * vw;   // row = b * H + y
    const long long b = row / p.H, y = row - b * p.H, v = col / p.W, x = col - v * p.W;
    return ((b * p.V + v) * p.H + y) * p.W + x;
}

__global__ void __launch_bounds__(256)
loss_terms_kernel(const LsP p, float *__restrict__ partials) {
    __shared__ float red[4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = (long long)blockIdx.x * LS_PIX + k * 256 + threadIdx.x;
        if (i >= p.n) continue;
        const float *t = p.tar + 3 * tar_pixel(p, i);
        const float t0 = t[0], t1 = t[1], t2 = t[2];
        {
            const float *a = p.image + 3 * i;
            const float d0 = a[0] - t0, d1 = a[1] - t1, d2 = a[2] - t2;
            s[0] += d0 * d0 + d1 * d1 + d2 * d2;
        }
        if (p.image_fine) {
            const float *a = p.image_fine + 3 * i;
            const float d0 = a[0] - t0, d1 = a[1] - t1, d2 = a[2] - t2;
            s[1] += d0 * d0 + d1 * d1 + d2 * d2;
        }
        if (p.rend_dist) s[2] += p.rend_dist[i];
        if (p.rend_normal) {
            const float *a = p.rend_normal + 3 * i, *b = p.depth_normal + 3 * i;
            s[3] += (1.0f - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2])) * p.acc[i];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float v = s[q];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) partials[(size_t)blockIdx.x * 4 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// terms[q] = (sum of the workgroups' partials, fixed order) * scale_q
__global__ void __launch_bounds__(1024)
loss_reduce_kernel(const float *__restrict__ partials, const int blocks, const float inv3n, const float invn, float *__restrict__ terms) {
    __shared__ double red[4][1024];
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = threadIdx.x; k < blocks; k += 1024)
#pragma unroll
        for (int q = 0; q < 4; q++) s[q] += (double)partials[(size_t)k * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; q++) red[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d)
#pragma unroll
            for (int q = 0; q < 4; q++) red[q][threadIdx.x] += red[q][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x < 4) terms[threadIdx.x] = (float)(red[threadIdx.x][0] * (double)(threadIdx.x < 2 ? inv3n : invn));
}

struct LsB {
    const float *g;
    float *d_image, *d_image_fine, *d_rend_dist, *d_rend_normal, *d_depth_normal;
    float inv3n, invn;
};

__global__ void __launch_bounds__(256)
loss_terms_bwd_kernel(const LsP p, const LsB o) {
    const float g0 = o.g[0] * 2.0f * o.inv3n, g1 = o.g[1] * 2.0f * o.inv3n, g2 = o.g[2] * o.invn, g3 = o.g[3] * o.invn;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = (long long)blockIdx.x * LS_PIX + k * 256 + threadIdx.x;
        if (i >= p.n) continue;
        if (o.d_image || o.d_image_fine) {
            const float *t = p.tar + 3 * tar_pixel(p, i);
            const float t0 = t[0], t1 = t[1], t2 = t[2];
            if (o.d_image) {
                const float *a = p.image + 3 * i;
                float *d = o.d_image + 3 * i;
                d[0] = g0 * (a[0] - t0); d[1] = g0 * (a[1] - t1); d[2] = g0 * (a[2] - t2);
            }
            if (o.d_image_fine) {
                const float *a = p.image_fine + 3 * i;
                float *d = o.d_image_fine + 3 * i;
                d[0] = g1 * (a[0] - t0); d[1] = g1 * (a[1] - t1); d[2] = g1 * (a[2] - t2);
            }
        }
        if (o.d_rend_dist) o.d_rend_dist[i] = g2;
        if (o.d_rend_normal || o.d_depth_normal) {
            const float w = -g3 * p.acc[i];
            const float *a = p.rend_normal + 3 * i, *b = p.depth_normal + 3 * i;
            if (o.d_rend_normal) { float *d = o.d_rend_normal + 3 * i; d[0] = w * b[0]; d[1] = w * b[1]; d[2] = w * b[2]; }
            if (o.d_depth_normal) { float *d = o.d_depth_normal + 3 * i; d[0] = w * a[0]; d[1] = w * a[1]; d[2] = w * a[2]; }
        }
    }
}

bool bad_dims(int B, int V, int H, int W) { return B < 0 || V <= 0 || H <= 0 || W <= 0 || (long long)B * V * H * W >= (1ll << 40); }

}  // namespace

extern "C" {

int64_t lara_loss_partial_floats(int64_t pixels) { return pixels < 0 ? LARA2DGS_E_INVALID : ((pixels + LS_PIX - 1) / LS_PIX) * 4 + 4; }

int lara_loss_terms_forward(int32_t B, int32_t V, int32_t H, int32_t W, const float *tar_rgb, const float *image,
                            const float *image_fine, const float *rend_dist, const float *rend_normal,
                            const float *depth_normal, const float *acc_map, float *terms, float *partials, void *stream) {
    if (bad_dims(B, V, H, W) || !terms || !partials) return LARA2DGS_E_INVALID;
    if ((rend_normal != nullptr) != (depth_normal != nullptr) || (rend_normal && !acc_map)) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)B * V * H * W;
    if (n == 0) return hipMemsetAsync(terms, 0, 4 * sizeof(float), s) == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
    if (!tar_rgb || !image) return LARA2DGS_E_INVALID;
    const LsP p{B, V, H, W, n, tar_rgb, image, image_fine, rend_dist, rend_normal, depth_normal, acc_map};
    const int blocks = (int)((n + LS_PIX - 1) / LS_PIX);
    {
        L2D_PROF("loss_terms_fwd", s);
        hipLaunchKernelGGL(loss_terms_kernel, dim3(blocks), dim3(256), 0, s, p, partials);
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, blocks, (float)(1.0 / (3.0 * (double)n)),
                           (float)(1.0 / (double)n), terms);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_loss_terms_backward(int32_t B, int32_t V, int32_t H, int32_t W, const float *tar_rgb, const float *image,
                             const float *image_fine, const float *rend_normal, const float *depth_normal,
                             const float *acc_map, const float *g_terms, float *d_image, float *d_image_fine,
                             float *d_rend_dist, float *d_rend_normal, float *d_depth_normal, void *stream) {
    if (bad_dims(B, V, H, W) || !g_terms) return LARA2DGS_E_INVALID;
    const long long n = (long long)B * V * H * W;
    if (n == 0) return LARA2DGS_OK;
    if (((d_image || d_image_fine) && !tar_rgb) || (d_image && !image) || (d_image_fine && !image_fine)) return LARA2DGS_E_INVALID;
    if ((d_rend_normal || d_depth_normal) && (!rend_normal || !depth_normal || !acc_map)) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const LsP p{B, V, H, W, n, tar_rgb, image, image_fine, nullptr, rend_normal, depth_normal, acc_map};
    const LsB o{g_terms, d_image, d_image_fine, d_rend_dist, d_rend_normal, d_depth_normal, (float)(1.0 / (3.0 * (double)n)),
                (float)(1.0 / (double)n)};
    {
        L2D_PROF("loss_terms_bwd", s);
        hipLaunchKernelGGL(loss_terms_bwd_kernel, dim3((unsigned)((n + LS_PIX - 1) / LS_PIX)), dim3(256), 0, s, p, o);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
