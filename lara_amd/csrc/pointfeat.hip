// pointfeat.hip -- LaRa's fine-stage point sampler (lightning/network.py:390-411 with projection :182-187):
// projection of the Gaussian centres into the input views + bilinear gather of the 8-channel image stack + depth
// residual, one thread per (point, view); the backward scatters into the render-derived channels with float
// atomics and sums the coordinate gradient over the views with lane shuffles.
// include/lara_pointfeat.h has the formulas and the contract.
#include "common.h"
#include "../../include/lara_pointfeat.h"

namespace {

constexpr int PF_MAX_VIEWS = 8;

struct PfP {
    int n, V, h, w;
    const float *points, *w2cs, *ixts, *img_ref, *image, *acc, *depth;
};

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

// channel c of view v at pixel index `pix` (c: 0-2 input image, planar; 3-5 render, channel-last; 6 acc; 7 depth)
__device__ __forceinline__ float chan(const PfP &p, const int v, const int c, const int pix) {
    const size_t hw = (size_t)p.h * p.w;
    if (c < 3) return p.img_ref[((size_t)v * 3 + c) * hw + pix];
    if (c < 6) return p.image[((size_t)v * hw + pix) * 3 + (c - 3)];
    return c == 6 ? p.acc[(size_t)v * hw + pix] : p.depth[(size_t)v * hw + pix];
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
point_feats_fwd_kernel(const PfP p, float *__restrict__ out) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(tid / VP), v = (int)(tid % VP);
    if (i >= p.n || v >= p.V) return;
    const float px = p.points[3 * (size_t)i], py = p.points[3 * (size_t)i + 1], pz = p.points[3 * (size_t)i + 2];
    const Proj pr = project(p, v, px, py, pz);
    const Tap t = taps(pr.x, pr.y, p.h, p.w);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (t.in[k]) s += t.wgt[k] * chan(p, v, c, t.idx[k]);
        out[((size_t)v * 8 + c) * p.n + i] = c == 7 ? fabsf(s - pr.z) : s;
    }
}

template <int VP>
__global__ void __launch_bounds__(256)
point_feats_bwd_kernel(const PfP p, const float *__restrict__ g_out, float *__restrict__ d_points,
                       float *d_image, float *d_acc, float *d_depth) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(tid / VP), v = (int)(tid % VP);
    const bool active = i < p.n && v < p.V;
    const size_t hw = (size_t)p.h * p.w;
    float dpx = 0.f, dpy = 0.f, dpz = 0.f;
    if (active) {
        const float px = p.points[3 * (size_t)i], py = p.points[3 * (size_t)i + 1], pz = p.points[3 * (size_t)i + 2];
        const Proj pr = project(p, v, px, py, pz);
        const Tap t = taps(pr.x, pr.y, p.h, p.w);
        float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float g = g_out[((size_t)v * 8 + c) * p.n + i];
            float val[4];
#pragma unroll
            for (int k = 0; k < 4; k++) val[k] = t.in[k] ? chan(p, v, c, t.idx[k]) : 0.f;
            if (c == 7) {  // | s - z |
                const float s = t.wgt[0] * val[0] + t.wgt[1] * val[1] + t.wgt[2] * val[2] + t.wgt[3] * val[3];
                const float d = s - pr.z, sg = d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f;
                gz -= sg * g;
                g *= sg;
            }
            // d s / d x, d s / d y (a tap outside the image contributes its zero)
            gx += g * ((val[1] - val[0]) * (1.f - t.wy) + (val[3] - val[2]) * t.wy);
            gy += g * ((val[2] - val[0]) * (1.f - t.wx) + (val[3] - val[1]) * t.wx);
            if (c >= 3) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (!t.in[k]) continue;
                    if (c < 6) { if (d_image) atomicAdd(d_image + ((size_t)v * hw + t.idx[k]) * 3 + (c - 3), t.wgt[k] * g); }
                    else if (c == 6) { if (d_acc) atomicAdd(d_acc + (size_t)v * hw + t.idx[k], t.wgt[k] * g); }
                    else { if (d_depth) atomicAdd(d_depth + (size_t)v * hw + t.idx[k], t.wgt[k] * g); }
                }
            }
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
#pragma unroll
    for (int o = 1; o < VP; o <<= 1) {  // sum over the views of the point (neighbouring lanes; fixed order)
        dpx += __shfl_xor(dpx, o, 64); dpy += __shfl_xor(dpy, o, 64); dpz += __shfl_xor(dpz, o, 64);
    }
    if (active && v == 0) {
        d_points[3 * (size_t)i] = dpx; d_points[3 * (size_t)i + 1] = dpy; d_points[3 * (size_t)i + 2] = dpz;
    }
}

template <int VP>
void launch_fwd(const PfP &p, float *out, hipStream_t s) {
    const size_t threads = (size_t)p.n * VP;
    hipLaunchKernelGGL((point_feats_fwd_kernel<VP>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p, out);
}
template <int VP>
void launch_bwd(const PfP &p, const float *g_out, float *d_points, float *d_image, float *d_acc, float *d_depth, hipStream_t s) {
    const size_t threads = (size_t)p.n * VP;
    hipLaunchKernelGGL((point_feats_bwd_kernel<VP>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p, g_out, d_points,
                       d_image, d_acc, d_depth);
}

}  // namespace

extern "C" {

int lara_point_feats_forward(int32_t n, int32_t V, int32_t h, int32_t w, const float *points, const float *w2cs,
                             const float *ixts, const float *img_ref, const float *image, const float *acc_map,
                             const float *depth, float *out, void *stream) {
    if (n < 0 || V <= 0 || V > PF_MAX_VIEWS || h <= 0 || w <= 0 || (int64_t)h * w >= (1ll << 31)) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!points || !w2cs || !ixts || !img_ref || !image || !acc_map || !depth || !out) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const PfP p{n, V, h, w, points, w2cs, ixts, img_ref, image, acc_map, depth};
    {
        L2D_PROF("point_feats_fwd", s);
        if (V == 1) launch_fwd<1>(p, out, s);
        else if (V == 2) launch_fwd<2>(p, out, s);
        else if (V <= 4) launch_fwd<4>(p, out, s);
        else launch_fwd<8>(p, out, s);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_point_feats_backward(int32_t n, int32_t V, int32_t h, int32_t w, const float *points, const float *w2cs,
                              const float *ixts, const float *img_ref, const float *image, const float *acc_map,
                              const float *depth, const float *g_out, float *d_points, float *d_image,
                              float *d_acc_map, float *d_depth, void *stream) {
    if (n < 0 || V <= 0 || V > PF_MAX_VIEWS || h <= 0 || w <= 0 || (int64_t)h * w >= (1ll << 31)) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!points || !w2cs || !ixts || !img_ref || !image || !acc_map || !depth || !g_out || !d_points) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const PfP p{n, V, h, w, points, w2cs, ixts, img_ref, image, acc_map, depth};
    {
        L2D_PROF("point_feats_bwd", s);
        if (V == 1) launch_bwd<1>(p, g_out, d_points, d_image, d_acc_map, d_depth, s);
        else if (V == 2) launch_bwd<2>(p, g_out, d_points, d_image, d_acc_map, d_depth, s);
        else if (V <= 4) launch_bwd<4>(p, g_out, d_points, d_image, d_acc_map, d_depth, s);
        else launch_bwd<8>(p, g_out, d_points, d_image, d_acc_map, d_depth, s);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
