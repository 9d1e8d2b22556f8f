// surface.hip -- LaRa's per-view render post-processing (lightning/renderer_2dgs.py:220-268 and :73-89) as
// one streaming kernel per direction: clamp, expected / median depth, normals to world space, finite-difference
// surface normals from the depth map and the rays.  One thread per pixel; the stencil reads go through L1/L2
// (a frame's maps are 10 MB).  include/lara_surface.h has the formulas and the contract.
#include <cfloat>

#include "common.h"
#include "../../include/lara_surface.h"

namespace {

struct SurfP {
    int H, W;
    const float *color, *allmap, *rays;   // of view 0; view v (= blockIdx.y) sits v * (3 | 7 | 6) * H * W floats further
    const float *rot;  // device pointer: 3 x 3 per view, row-major (read by every thread through the scalar cache)
    float ratio;
    int row_views;     // the output / output-gradient maps are [H, row_views * W, C]: view v owns columns [v * W, (v + 1) * W)
};

// this workgroup's view: input planes of view v, and the pixel's index in the [H, row_views * W] output maps
__device__ __forceinline__ SurfP view_of(SurfP p, const int v) {
    const size_t HW = (size_t)p.H * p.W;
    p.color += (size_t)v * 3 * HW;
    p.allmap += (size_t)v * 7 * HW;
    p.rays += (size_t)v * 6 * HW;
    p.rot += v * 9;
    return p;
}
__device__ __forceinline__ size_t out_pix(const SurfP &p, const int v, const int y, const int x) {
    return ((size_t)y * p.row_views + v) * p.W + x;
}

__device__ __forceinline__ float nan_to_num00(const float v) {  // torch.nan_to_num(v, 0, 0): nan, +inf -> 0
    if (v != v || v == INFINITY) return 0.f;
    return v == -INFINITY ? -FLT_MAX : v;
}

// surf_depth of a pixel (renderer_2dgs.py:234-247)
__device__ __forceinline__ float surf_depth(const SurfP &p, const size_t pix, const size_t HW) {
    const float e = nan_to_num00(p.allmap[pix] / p.allmap[HW + pix]);
    const float md = nan_to_num00(p.allmap[5 * HW + pix]);
    return e * (1.0f - p.ratio) + p.ratio * md;
}

__device__ __forceinline__ float3 surf_point(const SurfP &p, const int y, const int x, const size_t HW) {  // :73-75
    const size_t pix = (size_t)y * p.W + x;
    const float d = surf_depth(p, pix, HW);
    const float *r = p.rays + pix * 6;
    return make_float3(r[0] + d * r[3], r[1] + d * r[4], r[2] + d * r[5]);
}

__device__ __forceinline__ float3 cross3(const float3 a, const float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float3 sub3(const float3 a, const float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }

__global__ void __launch_bounds__(256)
surface_fwd_kernel(const SurfP p0, float *__restrict__ image, float *__restrict__ depth, float *__restrict__ acc,
                   float *__restrict__ rnorm, float *__restrict__ dnorm, float *__restrict__ rdist) {
    const SurfP p = view_of(p0, (int)blockIdx.y);
    const size_t HW = (size_t)p.H * p.W;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int y = (int)(pix / p.W), x = (int)(pix - (size_t)y * p.W);
    const size_t o = out_pix(p, (int)blockIdx.y, y, x);
#pragma unroll
    for (int c = 0; c < 3; c++) image[o * 3 + c] = fminf(fmaxf(p.color[c * HW + pix], 0.f), 1.f);
    const float alpha = p.allmap[HW + pix];
    acc[o] = alpha;
    rdist[o] = p.allmap[6 * HW + pix];
    const float n0 = p.allmap[2 * HW + pix], n1 = p.allmap[3 * HW + pix], n2 = p.allmap[4 * HW + pix];
#pragma unroll
    for (int j = 0; j < 3; j++) rnorm[o * 3 + j] = n0 * p.rot[j] + n1 * p.rot[3 + j] + n2 * p.rot[6 + j];
    depth[o] = surf_depth(p, pix, HW);
    float3 n = make_float3(0.f, 0.f, 0.f);
    if (y >= 1 && y < p.H - 1 && x >= 1 && x < p.W - 1) {
        const float3 a = sub3(surf_point(p, y + 1, x, HW), surf_point(p, y - 1, x, HW));
        const float3 b = sub3(surf_point(p, y, x + 1, HW), surf_point(p, y, x - 1, HW));
        const float3 c = cross3(a, b);
        const float inv = 1.0f / fmaxf(sqrtf(c.x * c.x + c.y * c.y + c.z * c.z), 1e-12f);  // F.normalize's eps
        n = make_float3(c.x * inv * alpha, c.y * inv * alpha, c.z * inv * alpha);
    }
    dnorm[o * 3 + 0] = n.x; dnorm[o * 3 + 1] = n.y; dnorm[o * 3 + 2] = n.z;
}

// gradients w.r.t. the two difference vectors of the normal at interior pixel q (zero elsewhere):
//   c = a x b, n = c / max(|c|, eps), out = n * alpha(q);  ga = b x gc, gb = gc x a
__device__ __forceinline__ void normal_vjp(const SurfP &p, const float *__restrict__ g_dn, const int v, const int y, const int x,
                                           const size_t HW, float3 &ga, float3 &gb) {
    ga = gb = make_float3(0.f, 0.f, 0.f);
    if (y < 1 || y >= p.H - 1 || x < 1 || x >= p.W - 1) return;
    const size_t pix = (size_t)y * p.W + x, o = out_pix(p, v, y, x);
    const float alpha = p.allmap[HW + pix];
    const float3 g = make_float3(g_dn[o * 3] * alpha, g_dn[o * 3 + 1] * alpha, g_dn[o * 3 + 2] * alpha);
    const float3 a = sub3(surf_point(p, y + 1, x, HW), surf_point(p, y - 1, x, HW));
    const float3 b = sub3(surf_point(p, y, x + 1, HW), surf_point(p, y, x - 1, HW));
    const float3 c = cross3(a, b);
    const float len = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
    float3 gc;
    if (len > 1e-12f) {
        const float inv = 1.0f / len;
        const float3 n = make_float3(c.x * inv, c.y * inv, c.z * inv);
        const float dot = n.x * g.x + n.y * g.y + n.z * g.z;
        gc = make_float3((g.x - n.x * dot) * inv, (g.y - n.y * dot) * inv, (g.z - n.z * dot) * inv);
    } else {
        gc = make_float3(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f);
    }
    ga = cross3(b, gc);
    gb = cross3(gc, a);
}

__global__ void __launch_bounds__(256)
surface_bwd_kernel(const SurfP p0, const float *__restrict__ g_image, const float *__restrict__ g_depth,
                   const float *__restrict__ g_acc, const float *__restrict__ g_rn, const float *__restrict__ g_dn,
                   const float *__restrict__ g_rd, float *__restrict__ d_color, float *__restrict__ d_allmap) {
    const int vw = (int)blockIdx.y;
    const SurfP p = view_of(p0, vw);
    const size_t HW = (size_t)p.H * p.W;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int y = (int)(pix / p.W), x = (int)(pix - (size_t)y * p.W);
    const size_t o = out_pix(p, vw, y, x);
    d_color += (size_t)vw * 3 * HW;
#pragma unroll
    for (int c = 0; c < 3; c++) {  // clamp passes the gradient inside [0, 1] (bounds included, as torch does)
        const float v = p.color[c * HW + pix];
        d_color[c * HW + pix] = (g_image && v >= 0.f && v <= 1.f) ? g_image[o * 3 + c] : 0.f;
    }
    if (!d_allmap) return;      // no gradient on any of the five maps: the seven planes would be zeros (the caller passes NULL on)
    d_allmap += (size_t)vw * 7 * HW;
    // gradient of the surface depth: direct + through the points of the four neighbouring normals
    float gs = g_depth ? g_depth[o] : 0.f;
    if (g_dn) {
        float3 ga, gb, gp = make_float3(0.f, 0.f, 0.f);
        normal_vjp(p, g_dn, vw, y - 1, x, HW, ga, gb); gp.x += ga.x; gp.y += ga.y; gp.z += ga.z;  // P(p) is its P(y+1, x)
        normal_vjp(p, g_dn, vw, y + 1, x, HW, ga, gb); gp.x -= ga.x; gp.y -= ga.y; gp.z -= ga.z;
        normal_vjp(p, g_dn, vw, y, x - 1, HW, ga, gb); gp.x += gb.x; gp.y += gb.y; gp.z += gb.z;  // its P(y, x+1)
        normal_vjp(p, g_dn, vw, y, x + 1, HW, ga, gb); gp.x -= gb.x; gp.y -= gb.y; gp.z -= gb.z;
        const float *r = p.rays + pix * 6;
        gs += gp.x * r[3] + gp.y * r[4] + gp.z * r[5];
    }
    const float c0 = p.allmap[pix], alpha = p.allmap[HW + pix], md = p.allmap[5 * HW + pix];
    const float e = c0 / alpha;
    const bool e_ok = e == e && e != INFINITY && e != -INFINITY && alpha != 0.f;
    const bool md_ok = md == md && md != INFINITY && md != -INFINITY;
    const float ge = e_ok ? gs * (1.0f - p.ratio) : 0.f;
    d_allmap[pix] = e_ok ? ge / alpha : 0.f;
    d_allmap[HW + pix] = (g_acc ? g_acc[o] : 0.f) + (e_ok ? -ge * c0 / (alpha * alpha) : 0.f);
#pragma unroll
    for (int i = 0; i < 3; i++)
        d_allmap[(2 + i) * HW + pix] = g_rn ? p.rot[3 * i] * g_rn[o * 3] + p.rot[3 * i + 1] * g_rn[o * 3 + 1] + p.rot[3 * i + 2] * g_rn[o * 3 + 2] : 0.f;
    d_allmap[5 * HW + pix] = md_ok ? gs * p.ratio : 0.f;
    d_allmap[6 * HW + pix] = g_rd ? g_rd[o] : 0.f;
}

// ---- the three activations in front of the rasteriser call (renderer_2dgs.py:181-189), one thread per Gaussian ----------
__global__ void __launch_bounds__(256)
activate_fwd_kernel(const int64_t P, const float *__restrict__ opacity, const float2 *__restrict__ scales,
                    const float4 *__restrict__ rotations, float *__restrict__ o_out, float2 *__restrict__ s_out,
                    float4 *__restrict__ r_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    o_out[i] = 1.0f / (1.0f + expf(-opacity[i]));                      // torch.sigmoid
    if (scales) { const float2 v = scales[i]; s_out[i] = make_float2(expf(v.x), expf(v.y)); }   // torch.exp
    if (rotations) {                                                    // F.normalize: x / max(|x|, 1e-12)
        const float4 q = rotations[i];
        const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        r_out[i] = make_float4(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm);      // (divisions, as torch: x / clamp_min(norm, eps))
    }
}

__global__ void __launch_bounds__(256)
activate_bwd_kernel(const int64_t P, const float *__restrict__ o_act, const float2 *__restrict__ s_act,
                    const float4 *__restrict__ rotations, const float *__restrict__ g_o, const float2 *__restrict__ g_s,
                    const float4 *__restrict__ g_r, float *__restrict__ d_o, float2 *__restrict__ d_s, float4 *__restrict__ d_r) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (d_o) { const float y = o_act[i]; d_o[i] = g_o ? g_o[i] * (1.0f - y) * y : 0.f; }
    if (d_s) {
        const float2 y = s_act[i], g = g_s ? g_s[i] : make_float2(0.f, 0.f);
        d_s[i] = make_float2(g.x * y.x, g.y * y.y);
    }
    if (d_r) {
        const float4 q = rotations[i], g = g_r ? g_r[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        if (n > 1e-12f) {      // y = q / n: dq = (g - y (y . g)) / n
            const float inv = 1.0f / n;
            const float4 y = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            const float dot = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
            d_r[i] = make_float4((g.x - y.x * dot) * inv, (g.y - y.y * dot) * inv, (g.z - y.z * dot) * inv, (g.w - y.w * dot) * inv);
        } else {               // clamped denominator: y = q * 1e12
            d_r[i] = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
        }
    }
}

}  // namespace

extern "C" {

int lara_activate_gaussians_forward(int64_t P, const float *opacity, const float *scales, const float *rotations,
                                    float *opacity_out, float *scales_out, float *rotations_out, void *stream) {
    if (P < 0) return LARA2DGS_E_INVALID;
    if (P == 0) return LARA2DGS_OK;
    if (!opacity || !opacity_out || (scales && !scales_out) || (rotations && !rotations_out)) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("activate_fwd", s);
        hipLaunchKernelGGL(activate_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, opacity, (const float2 *)scales,
                           (const float4 *)rotations, opacity_out, (float2 *)scales_out, (float4 *)rotations_out);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_activate_gaussians_backward(int64_t P, const float *opacity_act, const float *scales_act, const float *rotations,
                                     const float *g_opacity, const float *g_scales, const float *g_rotations,
                                     float *d_opacity, float *d_scales, float *d_rotations, void *stream) {
    if (P < 0) return LARA2DGS_E_INVALID;
    if (P == 0) return LARA2DGS_OK;
    if ((d_opacity && !opacity_act) || (d_scales && !scales_act) || (d_rotations && !rotations)) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("activate_bwd", s);
        hipLaunchKernelGGL(activate_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, opacity_act,
                           (const float2 *)scales_act, (const float4 *)rotations, g_opacity, (const float2 *)g_scales,
                           (const float4 *)g_rotations, d_opacity, (float2 *)d_scales, (float4 *)d_rotations);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_surface_maps_forward_views(int32_t n_views, int32_t H, int32_t W, const float *color, const float *allmap, const float *rays,
                                    const float *rots, float depth_ratio, float *image, float *depth, float *acc_map,
                                    float *rend_normal, float *depth_normal, float *rend_dist, void *stream) {
    if (n_views < 0 || H < 0 || W < 0 || n_views > 65535) return LARA2DGS_E_INVALID;
    if (n_views == 0 || H == 0 || W == 0) return LARA2DGS_OK;
    if (!color || !allmap || !rays || !rots || !image || !depth || !acc_map || !rend_normal || !depth_normal || !rend_dist)
        return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    SurfP p{H, W, color, allmap, rays, rots, depth_ratio, n_views};
    const size_t HW = (size_t)H * W;
    {
        L2D_PROF("surface_fwd", s);
        hipLaunchKernelGGL(surface_fwd_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)n_views), dim3(256), 0, s, p, image, depth,
                           acc_map, rend_normal, depth_normal, rend_dist);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_surface_maps_backward_views(int32_t n_views, int32_t H, int32_t W, const float *color, const float *allmap, const float *rays,
                                     const float *rots, float depth_ratio, const float *g_image, const float *g_depth,
                                     const float *g_acc_map, const float *g_rend_normal, const float *g_depth_normal,
                                     const float *g_rend_dist, float *d_color, float *d_allmap, void *stream) {
    if (n_views < 0 || H < 0 || W < 0 || n_views > 65535) return LARA2DGS_E_INVALID;
    if (n_views == 0 || H == 0 || W == 0) return LARA2DGS_OK;
    if (!color || !allmap || !rays || !rots || !d_color) return LARA2DGS_E_INVALID;
    if (!d_allmap && (g_depth || g_acc_map || g_rend_normal || g_depth_normal || g_rend_dist)) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    SurfP p{H, W, color, allmap, rays, rots, depth_ratio, n_views};
    const size_t HW = (size_t)H * W;
    {
        L2D_PROF("surface_bwd", s);
        hipLaunchKernelGGL(surface_bwd_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)n_views), dim3(256), 0, s, p, g_image, g_depth,
                           g_acc_map, g_rend_normal, g_depth_normal, g_rend_dist, d_color, d_allmap);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_surface_maps_forward(int32_t H, int32_t W, const float *color, const float *allmap, const float *rays,
                              const float *rot, float depth_ratio, float *image, float *depth, float *acc_map,
                              float *rend_normal, float *depth_normal, float *rend_dist, void *stream) {
    return lara_surface_maps_forward_views(1, H, W, color, allmap, rays, rot, depth_ratio, image, depth, acc_map, rend_normal,
                                           depth_normal, rend_dist, stream);
}

int lara_surface_maps_backward(int32_t H, int32_t W, const float *color, const float *allmap, const float *rays,
                               const float *rot, float depth_ratio, const float *g_image, const float *g_depth,
                               const float *g_acc_map, const float *g_rend_normal, const float *g_depth_normal,
                               const float *g_rend_dist, float *d_color, float *d_allmap, void *stream) {
    return lara_surface_maps_backward_views(1, H, W, color, allmap, rays, rot, depth_ratio, g_image, g_depth, g_acc_map,
                                            g_rend_normal, g_depth_normal, g_rend_dist, d_color, d_allmap, stream);
}

}  // extern "C"
