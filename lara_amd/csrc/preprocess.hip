// preprocess.hip -- per-surfel stages of the 2DGS rasteriser for gfx950 (forward + backward).
//
// Compiled with -ffp-contract=off: the INTEGER results of this stage (radius, tile rectangle,
// tiles touched, depth key bits) must be bit-identical to the CPU oracle's, so the fp32 operation
// order below is part of the contract and no FMA contraction is allowed.  The stage is a pure
// stream over P surfels (88 B in, 80 B + 12 B out per surfel) -> HBM-bound.
//
// Replaces the reference's (absent) `preprocessCUDA` forward/backward; behaviour restated from the
// published 2DGS rasteriser, call-site contract at lightning/renderer_2dgs.py:209-218.
#include "common.h"

namespace {

__device__ __constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                                           0.31539156525252005f, -1.0925484305920792f,
                                           0.5462742152960396f};
__device__ __constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                           -0.4570457994644658f, 0.3731763325901154f,
                                           -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

struct Pm43 { float m[4][3]; };

// projmatrix (row-vector convention) times ndc->pixel:  pixel = ((ndc + 1) * W - 1) / 2
__device__ __forceinline__ Pm43 build_Pm(const ViewDev &v) {
    Pm43 r;
    const float hw = (float)v.W / 2.0f, hh = (float)v.H / 2.0f;
    const float cw = (float)(v.W - 1) / 2.0f, ch = (float)(v.H - 1) / 2.0f;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const float r0 = v.projmatrix[4 * a + 0], r1 = v.projmatrix[4 * a + 1], r3 = v.projmatrix[4 * a + 3];
        r.m[a][0] = r0 * hw + r3 * cw;
        r.m[a][1] = r1 * hh + r3 * ch;
        r.m[a][2] = r3;
    }
    return r;
}

__device__ __forceinline__ void point4x3(const float *m, const float p[3], float o[3]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void vec4x3(const float *m, const float p[3], float o[3]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2];
}
__device__ __forceinline__ void vec4x3T(const float *m, const float p[3], float o[3]) {
    o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
    o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
    o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}

// R[r][c] from quaternion (w,x,y,z)
__device__ __forceinline__ void quat_to_rotmat(const float4 q, float R[3][3], float n[4]) {
    const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const float s = 1.0f / sqrtf(n2);
    const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
    n[0] = w; n[1] = x; n[2] = y; n[3] = z;
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[1][0] = 2.f * (x * y + w * z);
    R[2][0] = 2.f * (x * z - w * y);
    R[0][1] = 2.f * (x * y - w * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[2][1] = 2.f * (y * z + w * x);
    R[0][2] = 2.f * (x * z + w * y);
    R[1][2] = 2.f * (y * z - w * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// T(i,j) = sum_a Mrow_i[a] Pm[a][j];  rows: u axis, v axis, centre;  cols: x*w, y*w, w
__device__ __forceinline__ void compute_transmat(const ViewDev &v, const Pm43 &Pm, const float p[3],
                                                 const float2 scale, const float4 rot,
                                                 float Tm[3][3], float normal[3], float R[3][3],
                                                 float qn[4]) {
    quat_to_rotmat(rot, R, qn);
    const float sx = v.scale_modifier * scale.x, sy = v.scale_modifier * scale.y;
    const float L0[3] = {R[0][0] * sx, R[1][0] * sx, R[2][0] * sx};
    const float L1[3] = {R[0][1] * sy, R[1][1] * sy, R[2][1] * sy};
    const float L2[3] = {R[0][2], R[1][2], R[2][2]};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Tm[0][j] = L0[0] * Pm.m[0][j] + L0[1] * Pm.m[1][j] + L0[2] * Pm.m[2][j];
        Tm[1][j] = L1[0] * Pm.m[0][j] + L1[1] * Pm.m[1][j] + L1[2] * Pm.m[2][j];
        Tm[2][j] = p[0] * Pm.m[0][j] + p[1] * Pm.m[1][j] + p[2] * Pm.m[2][j] + Pm.m[3][j];
    }
    vec4x3(v.viewmatrix, L2, normal);
}

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

template <int DEG>
__device__ __forceinline__ float sh_channel(const float *s, float x, float y, float z) {
    // s[k] = coefficient k of this channel
    float r = SH_C0 * s[0];
    if (DEG > 0) {
        r = r - SH_C1 * y * s[1] + SH_C1 * z * s[2] - SH_C1 * x * s[3];
        if (DEG > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + kSH_C2[0] * xy * s[4] + kSH_C2[1] * yz * s[5] +
                kSH_C2[2] * (2.0f * zz - xx - yy) * s[6] + kSH_C2[3] * xz * s[7] +
                kSH_C2[4] * (xx - yy) * s[8];
            if (DEG > 2) {
                r = r + kSH_C3[0] * y * (3.0f * xx - yy) * s[9] + kSH_C3[1] * xy * z * s[10] +
                    kSH_C3[2] * y * (4.0f * zz - xx - yy) * s[11] +
                    kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[12] +
                    kSH_C3[4] * x * (4.0f * zz - xx - yy) * s[13] +
                    kSH_C3[5] * z * (xx - yy) * s[14] + kSH_C3[6] * x * (xx - 3.0f * yy) * s[15];
            }
        }
    }
    return r + 0.5f;
}

// ------------------------------------------------------------------------------------------------
// forward: one thread per surfel
// ------------------------------------------------------------------------------------------------
// Returns the surfel's tile rectangle (empty = culled); writes its record, radius.
template <int DEG>
__device__ __forceinline__ ushort4
surfel_forward(const ViewDev &v, const int idx, const float *__restrict__ means3D,
               const float *__restrict__ shs, const float *__restrict__ colors_precomp,
               const float *__restrict__ opacities, const float2 *__restrict__ scales,
               const float4 *__restrict__ rotations, const float *__restrict__ transmat_precomp,
               float4 *__restrict__ geom, float4 *__restrict__ cullbox,
               int32_t *__restrict__ radii, float &depth_out) {
    const ushort4 culled = make_ushort4(0, 0, 0, 0);
    radii[idx] = 0;
    depth_out = 0.f;

    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float p_view[3];
    point4x3(v.viewmatrix, p, p_view);
    if (p_view[2] <= 0.2f) return culled;
    // opt-in (not in the reference): alpha = min(0.99, opacity * G) <= opacity, and the composite skips alpha < 1/255
    if (v.cull_transparent && opacities[idx] < (1.0f / 255.0f)) return culled;

    const Pm43 Pm = build_Pm(v);
    float Tm[3][3], normal[3];
    if (transmat_precomp == nullptr) {
        float R[3][3], qn[4];
        compute_transmat(v, Pm, p, scales[idx], rotations[idx], Tm, normal, R, qn);
    } else {
        const float *tp = transmat_precomp + 9 * (size_t)idx;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) Tm[i][j] = tp[3 * j + i];
        normal[0] = 0.f; normal[1] = 0.f; normal[2] = 1.f;
    }
    const float T0[3] = {Tm[0][0], Tm[1][0], Tm[2][0]};
    const float T1[3] = {Tm[0][1], Tm[1][1], Tm[2][1]};
    const float T3[3] = {Tm[0][2], Tm[1][2], Tm[2][2]};
    const float cosv = -(p_view[0] * normal[0] + p_view[1] * normal[1] + p_view[2] * normal[2]);
    if (cosv == 0.0f) return culled;
    const float mult = cosv > 0.0f ? 1.0f : -1.0f;
    normal[0] *= mult; normal[1] *= mult; normal[2] *= mult;

    // 3-sigma bounding box of the projected surfel
    const float t[3] = {CUTOFF * CUTOFF, CUTOFF * CUTOFF, -1.0f};
    const float distance = T3[0] * T3[0] * t[0] + T3[1] * T3[1] * t[1] + T3[2] * T3[2] * t[2];
    if (distance == 0.0f) return culled;
    const float inv = 1.0f / distance;
    const float f[3] = {inv * t[0], inv * t[1], inv * t[2]};
    const float ptx = f[0] * T0[0] * T3[0] + f[1] * T0[1] * T3[1] + f[2] * T0[2] * T3[2];
    const float pty = f[0] * T1[0] * T3[0] + f[1] * T1[1] * T3[1] + f[2] * T1[2] * T3[2];
    const float t0 = f[0] * T0[0] * T0[0] + f[1] * T0[1] * T0[1] + f[2] * T0[2] * T0[2];
    const float t1 = f[0] * T1[0] * T1[0] + f[1] * T1[1] * T1[1] + f[2] * T1[2] * T1[2];
    const float h0 = ptx * ptx - t0, h1 = pty * pty - t1;
    const float ext0 = sqrtf(fmaxf(1e-4f, h0)), ext1 = sqrtf(fmaxf(1e-4f, h1));
    const float radius = ceilf(fmaxf(fmaxf(ext0, ext1), CUTOFF * FILTER_SIZE));
    const int max_radius = (int)radius;  // v_cvt_i32_f32: saturating, NaN -> 0 (as the oracle)
    const int rx0 = imin(v.gx, imax(0, (int)((ptx - max_radius) / TILE)));
    const int ry0 = imin(v.gy, imax(0, (int)((pty - max_radius) / TILE)));
    const int rx1 = imin(v.gx, imax(0, (int)((ptx + max_radius + TILE - 1) / TILE)));
    const int ry1 = imin(v.gy, imax(0, (int)((pty + max_radius + TILE - 1) / TILE)));
    if ((uint32_t)(rx1 - rx0) * (uint32_t)(ry1 - ry0) == 0) return culled;

    float rgb[3];
    uint32_t clamp_bits = 0;
    if (colors_precomp == nullptr) {
        float dir[3] = {p[0] - v.campos[0], p[1] - v.campos[1], p[2] - v.campos[2]};
        const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        const float x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
        const float *sh = shs + (size_t)idx * v.M * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float s[(DEG + 1) * (DEG + 1)];
#pragma unroll
            for (int k = 0; k < (DEG + 1) * (DEG + 1); k++) s[k] = sh[3 * k + ch];
            const float r = sh_channel<DEG>(s, x, y, z);
            if (r < 0.0f) clamp_bits |= 1u << ch;
            rgb[ch] = fmaxf(r, 0.0f);
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) rgb[ch] = colors_precomp[3 * (size_t)idx + ch];
    }

    // Conservative pixel box of {alpha >= 1/255} for the composite's quadrant culling.
    //   alpha >= 1/255  <=>  min(rho3d, rho2d) <= tau,  tau = 2 ln(255 opacity)
    //   rho2d <= tau: disc of radius sqrt(tau/2) around the 3-sigma box centre (ptx, pty)
    //   rho3d <= tau: the projected disc of radius sqrt(tau); its exact AABB is the formula above
    //                 with cutoff^2 = tau, valid while the disc stays in front of the w = 0 plane
    //                 (d < 0); otherwise no culling for this surfel.
    // Margins cover fp32 rounding here and in the composite's own evaluation of rho.
    const float opa = opacities[idx];
    const float INF = __uint_as_float(0x7f800000u);
    float4 cb = make_float4(INF, -INF, INF, -INF);  // empty: can never reach 1/255
    if (opa >= 1.0f / 255.0f) {
        const float tau = 2.0f * logf(255.0f * opa) * 1.0001f + 1e-3f;
        const float rA = sqrtf(0.5f * tau) + 0.01f;
        float minx = ptx - rA, maxx = ptx + rA, miny = pty - rA, maxy = pty + rA;
        const float tt[3] = {tau, tau, -1.0f};
        const float dd = T3[0] * T3[0] * tt[0] + T3[1] * T3[1] * tt[1] + T3[2] * T3[2] * tt[2];
        bool boxed = false;
        if (dd < 0.0f) {
            const float iv = 1.0f / dd;
            const float ff[3] = {iv * tt[0], iv * tt[1], iv * tt[2]};
            const float cx = ff[0] * T0[0] * T3[0] + ff[1] * T0[1] * T3[1] + ff[2] * T0[2] * T3[2];
            const float cy = ff[0] * T1[0] * T3[0] + ff[1] * T1[1] * T3[1] + ff[2] * T1[2] * T3[2];
            const float qx = ff[0] * T0[0] * T0[0] + ff[1] * T0[1] * T0[1] + ff[2] * T0[2] * T0[2];
            const float qy = ff[0] * T1[0] * T1[0] + ff[1] * T1[1] * T1[1] + ff[2] * T1[2] * T1[2];
            const float hx = sqrtf(fmaxf(0.0f, cx * cx - qx) + 1e-5f * cx * cx + 0.01f) * 1.001f + 0.05f;
            const float hy = sqrtf(fmaxf(0.0f, cy * cy - qy) + 1e-5f * cy * cy + 0.01f) * 1.001f + 0.05f;
            if (hx == hx && hy == hy && cx == cx && cy == cy) {  // no NaN
                boxed = true;
                minx = fminf(minx, cx - hx); maxx = fmaxf(maxx, cx + hx);
                miny = fminf(miny, cy - hy); maxy = fmaxf(maxy, cy + hy);
            }
        }
        cb = boxed ? make_float4(minx, maxx, miny, maxy) : make_float4(-INF, INF, -INF, INF);
    }
    cullbox[idx] = cb;

    float4 *g = geom + (size_t)idx * 5;
    g[0] = make_float4(T0[0], T0[1], T0[2], T1[0]);
    g[1] = make_float4(T1[1], T1[2], T3[0], T3[1]);
    g[2] = make_float4(T3[2], ptx, pty, opa);
    g[3] = make_float4(normal[0], normal[1], normal[2], p_view[2]);
    g[4] = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamp_bits));
    radii[idx] = max_radius;
    depth_out = p_view[2];
    return make_ushort4((unsigned short)rx0, (unsigned short)ry0, (unsigned short)rx1,
                        (unsigned short)ry1);
}

// forward kernel: one thread per surfel + binning pass 1 (per-tile population count).  Consecutive
// surfel ids are spatially coherent in LaRa (voxel-grid order), so a workgroup's 256 surfels hit
// only a few dozen distinct tiles: counts are aggregated in an LDS histogram and flushed with one
// device-scope atomic per touched tile instead of one per (surfel, tile) pair.
template <int DEG>
__device__ __forceinline__ void
preprocess_fwd_block(const ViewDev &v, const int block, const float *__restrict__ means3D, const float *__restrict__ shs,
                     const float *__restrict__ colors_precomp, const float *__restrict__ opacities,
                     const float2 *__restrict__ scales, const float4 *__restrict__ rotations,
                     const float *__restrict__ transmat_precomp, float4 *__restrict__ geom,
                     float4 *__restrict__ cullbox, uint4 *__restrict__ rect_out,
                     uint32_t *__restrict__ tile_count, uint32_t *__restrict__ block_tot,
                     int32_t *__restrict__ radii, const int use_lds, uint32_t *hist, uint32_t *wsum) {
    const int idx = block * blockDim.x + threadIdx.x;
    const int slice = block % L2D_SLICES;
    if (use_lds) {
        for (int t = threadIdx.x; t < v.tiles; t += blockDim.x) hist[t] = 0;
        __syncthreads();
    }
    ushort4 r = make_ushort4(0, 0, 0, 0);
    float view_depth = 0.f;
    if (idx < v.P)
        r = surfel_forward<DEG>(v, idx, means3D, shs, colors_precomp, opacities, scales, rotations,
                                transmat_precomp, geom, cullbox, radii, view_depth);
    // exclusive scan of the tiles touched inside the workgroup: surfel-major pair numbering, used by
    // the backward to gather a surfel's per-tile gradient rows without atomics
    const uint32_t tt = (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y);
    uint32_t incl = tt;
    {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
        if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    }
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += wsum[w];
    if (threadIdx.x == 255) block_tot[block] = woff + incl;
    if (idx < v.P) {
        // tile rectangle + depth key bits + local pair offset, 16 B per surfel, for the scatter pass
        // (the depth the record holds, from the register: reading it back from `geom` missed the L2 -- the record's stores do not
        // allocate there -- and cost a 64-byte fetch per surfel: FETCH_SIZE said 92 MB for this kernel's 46 MB of inputs in rounds
        // 1-4; tools/ubench/traffic_calib.hip's surfel88 / rec80_write patterns rule the other explanations out)
        const float depth = tt ? view_depth : 0.f;
        rect_out[idx] = make_uint4((uint32_t)r.x | ((uint32_t)r.y << 16), (uint32_t)r.z | ((uint32_t)r.w << 16),
                                   __float_as_uint(depth), woff + incl - tt);
    }
    for (int y = r.y; y < r.w; y++)
        for (int x = r.x; x < r.z; x++) {
            if (use_lds) atomicAdd(&hist[y * v.gx + x], 1u);
            else __hip_atomic_fetch_add(&tile_count[(y * v.gx + x) * L2D_SLICES + slice], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    if (use_lds) {
        __syncthreads();
        for (int t = threadIdx.x; t < v.tiles; t += blockDim.x) {
            const uint32_t c = hist[t];
            if (c) __hip_atomic_fetch_add(&tile_count[t * L2D_SLICES + slice], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int DEG>
__global__ void __launch_bounds__(256)
preprocess_fwd_kernel(ViewDev v, const float *__restrict__ means3D, const float *__restrict__ shs,
                      const float *__restrict__ colors_precomp, const float *__restrict__ opacities,
                      const float2 *__restrict__ scales, const float4 *__restrict__ rotations,
                      const float *__restrict__ transmat_precomp, float4 *__restrict__ geom,
                      float4 *__restrict__ cullbox, uint4 *__restrict__ rect_out,
                      uint32_t *__restrict__ tile_count, uint32_t *__restrict__ block_tot,
                      int32_t *__restrict__ radii, const int use_lds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    __shared__ uint32_t wsum[4];
    preprocess_fwd_block<DEG>(v, (int)blockIdx.x, means3D, shs, colors_precomp, opacities, scales, rotations, transmat_precomp,
                              geom, cullbox, rect_out, tile_count, block_tot, radii, use_lds, hist, wsum);
}

// ONE launch for the n cameras of a multi-view call ("shared preprocess inputs, per-camera T").
// The camera is the FAST index of the workgroup id: the n workgroups that process the same 256 surfels run next to
// each other, so the surfels' 88 input bytes come from HBM once and from L2 n - 1 times.
struct PreViews {
    int n;
    const float *bg[L2D_MAX_VIEWS], *viewmatrix[L2D_MAX_VIEWS], *projmatrix[L2D_MAX_VIEWS], *campos[L2D_MAX_VIEWS];
    float4 *geom[L2D_MAX_VIEWS], *cullbox[L2D_MAX_VIEWS];
    uint4 *rect[L2D_MAX_VIEWS];
    uint32_t *tile_count[L2D_MAX_VIEWS], *block_tot[L2D_MAX_VIEWS];
    int32_t *radii[L2D_MAX_VIEWS];
};

template <int DEG>
__global__ void __launch_bounds__(256)
preprocess_fwd_views_kernel(ViewDev v, PreViews pv, const float *__restrict__ means3D, const float *__restrict__ shs,
                            const float *__restrict__ colors_precomp, const float *__restrict__ opacities,
                            const float2 *__restrict__ scales, const float4 *__restrict__ rotations,
                            const float *__restrict__ transmat_precomp, const int use_lds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    __shared__ uint32_t wsum[4];
    const int view = (int)(blockIdx.x % (unsigned)pv.n), block = (int)(blockIdx.x / (unsigned)pv.n);
    v.bg = pv.bg[view]; v.viewmatrix = pv.viewmatrix[view]; v.projmatrix = pv.projmatrix[view]; v.campos = pv.campos[view];
    preprocess_fwd_block<DEG>(v, block, means3D, shs, colors_precomp, opacities, scales, rotations, transmat_precomp,
                              pv.geom[view], pv.cullbox[view], pv.rect[view], pv.tile_count[view], pv.block_tot[view],
                              pv.radii[view], use_lds, hist, wsum);
}

__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float *__restrict__ means3D, const float *__restrict__ viewmatrix,
                    uint8_t *__restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float pv[3];
    point4x3(viewmatrix, p, pv);
    present[idx] = pv[2] > 0.2f;
}

// ------------------------------------------------------------------------------------------------
// backward: one thread per surfel; reads the tile-reduced accumulators of the composite backward
//   grad[idx][0..8]  dL/dT (Tu,Tv,Tw)   [9..10] dL/dmean2D   [11..13] dL/dnormal
//   grad[idx][14]    dL/dopacity        [15..17] dL/drgb
// ------------------------------------------------------------------------------------------------
// everything one view contributes to a surfel's gradients
template <int DEG>
struct SurfelGrad {
    static constexpr int NSH = (DEG + 1) * (DEG + 1) * 3;
    float dmean[3], m2out[2], dopac, dscale[2], drot[4], dT[9], dcol[3], dsh[NSH];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 3; k++) { dmean[k] = 0.f; dcol[k] = 0.f; }
        m2out[0] = m2out[1] = dopac = dscale[0] = dscale[1] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) drot[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) dT[k] = 0.f;
#pragma unroll
        for (int k = 0; k < NSH; k++) dsh[k] = 0.f;
    }
    __device__ __forceinline__ void add(const SurfelGrad &o) {   // (no FMA contraction in this file: plain adds)
#pragma unroll
        for (int k = 0; k < 3; k++) { dmean[k] += o.dmean[k]; dcol[k] += o.dcol[k]; }
        m2out[0] += o.m2out[0]; m2out[1] += o.m2out[1]; dopac += o.dopac; dscale[0] += o.dscale[0]; dscale[1] += o.dscale[1];
#pragma unroll
        for (int k = 0; k < 4; k++) drot[k] += o.drot[k];
#pragma unroll
        for (int k = 0; k < 9; k++) dT[k] += o.dT[k];
#pragma unroll
        for (int k = 0; k < NSH; k++) dsh[k] += o.dsh[k];
    }
};

template <int DEG>
__device__ __forceinline__ void
surfel_backward(const ViewDev &v, const int idx, const float *__restrict__ means3D, const float *__restrict__ shs,
                const float *__restrict__ colors_precomp, const float2 *__restrict__ scales,
                const float4 *__restrict__ rotations, const float *__restrict__ transmat_precomp,
                const int32_t *__restrict__ radii, const float4 *__restrict__ geom,
                const uint32_t *__restrict__ pair_base, const float4 *__restrict__ pair_grad,
                const uint8_t *__restrict__ pair_valid, SurfelGrad<DEG> &G) {
    G.zero();
    const bool visible = radii[idx] > 0;

    // sum the surfel's per-tile gradient rows in tile order (deterministic; no atomics anywhere)
    float gacc[GRAD_F];
#pragma unroll
    for (int k = 0; k < GRAD_F; k++) gacc[k] = 0.f;
    if (visible) {
        const uint32_t q0 = pair_base[idx], q1 = pair_base[idx + 1];
        for (uint32_t q = q0; q < q1; q++) {     // the surfel's rows are contiguous (written surfel-major by composite_bwd)
            if (!pair_valid[q]) continue;        // no pixel used this pair
            const float4 *row = pair_grad + (size_t)q * (GRAD_F / 4);
#pragma unroll
            for (int k = 0; k < GRAD_F / 4; k++) {
                const float4 w4 = row[k];
                gacc[4 * k] += w4.x; gacc[4 * k + 1] += w4.y; gacc[4 * k + 2] += w4.z; gacc[4 * k + 3] += w4.w;
            }
        }
    }
    G.dopac = gacc[14];

    float dmean[3] = {0.f, 0.f, 0.f};
    float2 dscale = make_float2(0.f, 0.f);
    float4 drot = make_float4(0.f, 0.f, 0.f, 0.f);
    float dT[9];
#pragma unroll
    for (int k = 0; k < 9; k++) dT[k] = gacc[k];
    float m2out[3] = {0.f, 0.f, 0.f};

    if (visible) {
        const float4 *g = geom + (size_t)idx * 5;
        const float4 g0 = g[0], g1 = g[1], g2 = g[2], g4 = g[4];
        const float T0[3] = {g0.x, g0.y, g0.z};
        const float T1[3] = {g0.w, g1.x, g1.y};
        const float T3[3] = {g1.z, g1.w, g2.x};
        const float m2x = gacc[9], m2y = gacc[10];
        if (m2x != 0.0f || m2y != 0.0f) {
            // through the box-centre formula: centre = sum(f * T0 * T3), f = t / dot(t, T3*T3)
            const float t[3] = {9.0f, 9.0f, -1.0f};
            const float d = t[0] * T3[0] * T3[0] + t[1] * T3[1] * T3[1] + t[2] * T3[2] * T3[2];
            const float f[3] = {t[0] * (1.0f / d), t[1] * (1.0f / d), t[2] * (1.0f / d)};
            float dL_dT3[3], dL_df[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dT[0 + k] += m2x * f[k] * T3[k];
                dT[3 + k] += m2y * f[k] * T3[k];
                dL_dT3[k] = m2x * f[k] * T0[k] + m2y * f[k] * T1[k];
                dL_df[k] = m2x * T0[k] * T3[k] + m2y * T1[k] * T3[k];
            }
            const float dL_dd = (dL_df[0] * f[0] + dL_df[1] * f[1] + dL_df[2] * f[2]) * (-1.0f / d);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                dL_dT3[k] += dL_dd * (t[k] * T3[k] * 2.0f);
                dT[6 + k] += dL_dT3[k];
            }
        }
        const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        if (transmat_precomp == nullptr) {
            const Pm43 Pm = build_Pm(v);
            float Tm[3][3], normal[3], R[3][3], qn[4];
            const float2 sc = scales[idx];
            compute_transmat(v, Pm, p, sc, rotations[idx], Tm, normal, R, qn);
            float dM[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int a = 0; a < 3; a++)
                    dM[i][a] = dT[0 + i] * Pm.m[a][0] + dT[3 + i] * Pm.m[a][1] + dT[6 + i] * Pm.m[a][2];
            const float dn[3] = {gacc[11], gacc[12], gacc[13]};
            float dtn[3];
            vec4x3T(v.viewmatrix, dn, dtn);
            float p_view[3];
            point4x3(v.viewmatrix, p, p_view);
            const float cosv = -(p_view[0] * normal[0] + p_view[1] * normal[1] + p_view[2] * normal[2]);
            const float mult = cosv > 0.0f ? 1.0f : -1.0f;
            dtn[0] *= mult; dtn[1] *= mult; dtn[2] *= mult;
            const float sx = v.scale_modifier * sc.x, sy = v.scale_modifier * sc.y;
            float V[3][3];  // V[r][c] = dL/dR(r,c)
#pragma unroll
            for (int r = 0; r < 3; r++) { V[r][0] = dM[0][r] * sx; V[r][1] = dM[1][r] * sy; V[r][2] = dtn[r]; }
            dscale.x = v.scale_modifier * (dM[0][0] * R[0][0] + dM[0][1] * R[1][0] + dM[0][2] * R[2][0]);
            dscale.y = v.scale_modifier * (dM[1][0] * R[0][1] + dM[1][1] * R[1][1] + dM[1][2] * R[2][1]);
            const float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
            drot.x = 2.f * (x * (V[2][1] - V[1][2]) + y * (V[0][2] - V[2][0]) + z * (V[1][0] - V[0][1]));
            drot.y = 2.f * (-2.f * x * (V[1][1] + V[2][2]) + y * (V[1][0] + V[0][1]) + z * (V[2][0] + V[0][2]) + w * (V[2][1] - V[1][2]));
            drot.z = 2.f * (x * (V[1][0] + V[0][1]) - 2.f * y * (V[0][0] + V[2][2]) + z * (V[2][1] + V[1][2]) + w * (V[0][2] - V[2][0]));
            drot.w = 2.f * (x * (V[2][0] + V[0][2]) + y * (V[2][1] + V[1][2]) - 2.f * z * (V[0][0] + V[1][1]) + w * (V[1][0] - V[0][1]));
            dmean[0] = dM[2][0]; dmean[1] = dM[2][1]; dmean[2] = dM[2][2];
        }

        if (colors_precomp == nullptr) {
            const float *sh = shs + (size_t)idx * v.M * 3;
            float *dsh = G.dsh;
            const float dir_o[3] = {p[0] - v.campos[0], p[1] - v.campos[1], p[2] - v.campos[2]};
            const float len = sqrtf(dir_o[0] * dir_o[0] + dir_o[1] * dir_o[1] + dir_o[2] * dir_o[2]);
            const float x = dir_o[0] / len, y = dir_o[1] / len, z = dir_o[2] / len;
            const uint32_t clamp_bits = __float_as_uint(g4.w);
            float ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float gg = gacc[15 + ch] * ((clamp_bits >> ch) & 1u ? 0.0f : 1.0f);
                float dx = 0.f, dy = 0.f, dz = 0.f;
                dsh[0 * 3 + ch] = SH_C0 * gg;
                if (DEG > 0) {
                    const float s1 = sh[1 * 3 + ch], s2 = sh[2 * 3 + ch], s3 = sh[3 * 3 + ch];
                    dsh[1 * 3 + ch] = -SH_C1 * y * gg;
                    dsh[2 * 3 + ch] = SH_C1 * z * gg;
                    dsh[3 * 3 + ch] = -SH_C1 * x * gg;
                    dx = -SH_C1 * s3; dy = -SH_C1 * s1; dz = SH_C1 * s2;
                    if (DEG > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        const float s4 = sh[4 * 3 + ch], s5 = sh[5 * 3 + ch], s6 = sh[6 * 3 + ch],
                                    s7 = sh[7 * 3 + ch], s8 = sh[8 * 3 + ch];
                        dsh[4 * 3 + ch] = kSH_C2[0] * xy * gg;
                        dsh[5 * 3 + ch] = kSH_C2[1] * yz * gg;
                        dsh[6 * 3 + ch] = kSH_C2[2] * (2.f * zz - xx - yy) * gg;
                        dsh[7 * 3 + ch] = kSH_C2[3] * xz * gg;
                        dsh[8 * 3 + ch] = kSH_C2[4] * (xx - yy) * gg;
                        dx += kSH_C2[0] * y * s4 + kSH_C2[2] * 2.f * -x * s6 + kSH_C2[3] * z * s7 + kSH_C2[4] * 2.f * x * s8;
                        dy += kSH_C2[0] * x * s4 + kSH_C2[1] * z * s5 + kSH_C2[2] * 2.f * -y * s6 + kSH_C2[4] * 2.f * -y * s8;
                        dz += kSH_C2[1] * y * s5 + kSH_C2[2] * 2.f * 2.f * z * s6 + kSH_C2[3] * x * s7;
                        if (DEG > 2) {
                            const float s9 = sh[9 * 3 + ch], s10 = sh[10 * 3 + ch], s11 = sh[11 * 3 + ch],
                                        s12 = sh[12 * 3 + ch], s13 = sh[13 * 3 + ch], s14 = sh[14 * 3 + ch],
                                        s15 = sh[15 * 3 + ch];
                            dsh[9 * 3 + ch] = kSH_C3[0] * y * (3.f * xx - yy) * gg;
                            dsh[10 * 3 + ch] = kSH_C3[1] * xy * z * gg;
                            dsh[11 * 3 + ch] = kSH_C3[2] * y * (4.f * zz - xx - yy) * gg;
                            dsh[12 * 3 + ch] = kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gg;
                            dsh[13 * 3 + ch] = kSH_C3[4] * x * (4.f * zz - xx - yy) * gg;
                            dsh[14 * 3 + ch] = kSH_C3[5] * z * (xx - yy) * gg;
                            dsh[15 * 3 + ch] = kSH_C3[6] * x * (xx - 3.f * yy) * gg;
                            dx += kSH_C3[0] * s9 * 3.f * 2.f * xy + kSH_C3[1] * s10 * yz +
                                  kSH_C3[2] * s11 * -2.f * xy + kSH_C3[3] * s12 * -3.f * 2.f * xz +
                                  kSH_C3[4] * s13 * (-3.f * xx + 4.f * zz - yy) +
                                  kSH_C3[5] * s14 * 2.f * xz + kSH_C3[6] * s15 * 3.f * (xx - yy);
                            dy += kSH_C3[0] * s9 * 3.f * (xx - yy) + kSH_C3[1] * s10 * xz +
                                  kSH_C3[2] * s11 * (-3.f * yy + 4.f * zz - xx) +
                                  kSH_C3[3] * s12 * -3.f * 2.f * yz + kSH_C3[4] * s13 * -2.f * xy +
                                  kSH_C3[5] * s14 * -2.f * yz + kSH_C3[6] * s15 * -3.f * 2.f * xy;
                            dz += kSH_C3[1] * s10 * xy + kSH_C3[2] * s11 * 4.f * 2.f * yz +
                                  kSH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) +
                                  kSH_C3[4] * s13 * 4.f * 2.f * xz + kSH_C3[5] * s14 * (xx - yy);
                        }
                    }
                }
                ddir[0] += dx * gg; ddir[1] += dy * gg; ddir[2] += dz * gg;
            }
            const float sum2 = dir_o[0] * dir_o[0] + dir_o[1] * dir_o[1] + dir_o[2] * dir_o[2];
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - dir_o[0] * dir_o[0]) * ddir[0] - dir_o[1] * dir_o[0] * ddir[1] - dir_o[2] * dir_o[0] * ddir[2]) * invsum32;
            dmean[1] += (-dir_o[0] * dir_o[1] * ddir[0] + (sum2 - dir_o[1] * dir_o[1]) * ddir[1] - dir_o[2] * dir_o[1] * ddir[2]) * invsum32;
            dmean[2] += (-dir_o[0] * dir_o[2] * ddir[0] - dir_o[1] * dir_o[2] * ddir[1] + (sum2 - dir_o[2] * dir_o[2]) * ddir[2]) * invsum32;
        }
        // screen-space gradient handed back for densification heuristics (published behaviour)
        const float depth = g2.x;
        m2out[0] = gacc[2] * depth * 0.5f * (float)v.W;
        m2out[1] = gacc[5] * depth * 0.5f * (float)v.H;
    }

#pragma unroll
    for (int k = 0; k < 3; k++) { G.dmean[k] = dmean[k]; G.dcol[k] = gacc[15 + k]; }
    G.m2out[0] = m2out[0]; G.m2out[1] = m2out[1];
    G.dscale[0] = dscale.x; G.dscale[1] = dscale.y;
    G.drot[0] = drot.x; G.drot[1] = drot.y; G.drot[2] = drot.z; G.drot[3] = drot.w;
#pragma unroll
    for (int k = 0; k < 9; k++) G.dT[k] = visible ? dT[k] : 0.f;
}

template <int DEG>
__device__ __forceinline__ void
write_surfel_grad(const ViewDev &v, const int idx, const SurfelGrad<DEG> &G, const bool has_sh, const bool has_col,
                  const bool has_tm, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                  float *__restrict__ dL_dshs, float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacities,
                  float2 *__restrict__ dL_dscales, float4 *__restrict__ dL_drots, float *__restrict__ dL_dtransmat) {
    dL_dopacities[idx] = G.dopac;
    dL_dmeans3D[3 * idx + 0] = G.dmean[0];
    dL_dmeans3D[3 * idx + 1] = G.dmean[1];
    dL_dmeans3D[3 * idx + 2] = G.dmean[2];
    dL_dmeans2D[3 * idx + 0] = G.m2out[0];
    dL_dmeans2D[3 * idx + 1] = G.m2out[1];
    dL_dmeans2D[3 * idx + 2] = 0.f;
    if (!has_tm) {
        dL_dscales[idx] = make_float2(G.dscale[0], G.dscale[1]);
        dL_drots[idx] = make_float4(G.drot[0], G.drot[1], G.drot[2], G.drot[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) dL_dtransmat[9 * (size_t)idx + k] = G.dT[k];
    }
    if (has_sh) {
        float *dsh = dL_dshs + (size_t)idx * v.M * 3;
#pragma unroll
        for (int k = 0; k < SurfelGrad<DEG>::NSH; k++) dsh[k] = G.dsh[k];
        for (int k = SurfelGrad<DEG>::NSH; k < v.M * 3; k++) dsh[k] = 0.f;   // coefficients above the active degree
    }
    if (has_col) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * (size_t)idx + ch] = G.dcol[ch];
    }
}

template <int DEG>
__global__ void __launch_bounds__(256)
preprocess_bwd_kernel(ViewDev v, const float *__restrict__ means3D, const float *__restrict__ shs,
                      const float *__restrict__ colors_precomp, const float2 *__restrict__ scales,
                      const float4 *__restrict__ rotations, const float *__restrict__ transmat_precomp,
                      const int32_t *__restrict__ radii, const float4 *__restrict__ geom,
                      const uint32_t *__restrict__ pair_base,
                      const float4 *__restrict__ pair_grad, const uint8_t *__restrict__ pair_valid,
                      float *__restrict__ dL_dmeans3D,
                      float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dshs,
                      float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacities,
                      float2 *__restrict__ dL_dscales, float4 *__restrict__ dL_drots,
                      float *__restrict__ dL_dtransmat) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= v.P) return;
    SurfelGrad<DEG> G;
    surfel_backward<DEG>(v, idx, means3D, shs, colors_precomp, scales, rotations, transmat_precomp, radii, geom, pair_base,
                         pair_grad, pair_valid, G);
    write_surfel_grad<DEG>(v, idx, G, colors_precomp == nullptr, colors_precomp != nullptr, transmat_precomp != nullptr,
                           dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales, dL_drots, dL_dtransmat);
}

// ONE launch for the n views of a multi-view call: a thread walks its surfel through the views in order, adds each
// view's contribution in registers (view 0, then + view 1, ...: the order and the roundings of adding the per-view
// gradient tensors one after the other) and writes the summed gradient once -- the surfel's inputs are read once, no
// per-view gradient tensors, no summation pass.
struct BwdViews {
    int n;
    const float *viewmatrix[L2D_MAX_VIEWS], *projmatrix[L2D_MAX_VIEWS], *campos[L2D_MAX_VIEWS];
    const int32_t *radii[L2D_MAX_VIEWS];
    const float4 *geom[L2D_MAX_VIEWS], *pair_grad[L2D_MAX_VIEWS];
    const uint32_t *pair_base[L2D_MAX_VIEWS];
    const uint8_t *pair_valid[L2D_MAX_VIEWS];
};

template <int DEG>
__global__ void __launch_bounds__(256)
preprocess_bwd_views_kernel(ViewDev v, BwdViews bv, const int accumulate, const float *__restrict__ means3D,
                            const float *__restrict__ shs, const float *__restrict__ colors_precomp,
                            const float2 *__restrict__ scales, const float4 *__restrict__ rotations,
                            const float *__restrict__ transmat_precomp, float *dL_dmeans3D, float *dL_dmeans2D,
                            float *dL_dshs, float *dL_dcolors, float *dL_dopacities, float2 *dL_dscales, float4 *dL_drots,
                            float *dL_dtransmat) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= v.P) return;
    const bool has_sh = colors_precomp == nullptr, has_col = !has_sh, has_tm = transmat_precomp != nullptr;
    SurfelGrad<DEG> A;
    if (accumulate) {   // (a call with more than L2D_MAX_VIEWS views: continue the running sums of the launch before)
        A.zero();
        A.dopac = dL_dopacities[idx];
        for (int k = 0; k < 3; k++) A.dmean[k] = dL_dmeans3D[3 * idx + k];
        A.m2out[0] = dL_dmeans2D[3 * idx]; A.m2out[1] = dL_dmeans2D[3 * idx + 1];
        if (!has_tm) {
            const float2 a = dL_dscales[idx]; const float4 b = dL_drots[idx];
            A.dscale[0] = a.x; A.dscale[1] = a.y; A.drot[0] = b.x; A.drot[1] = b.y; A.drot[2] = b.z; A.drot[3] = b.w;
        } else {
            for (int k = 0; k < 9; k++) A.dT[k] = dL_dtransmat[9 * (size_t)idx + k];
        }
        if (has_sh) for (int k = 0; k < SurfelGrad<DEG>::NSH; k++) A.dsh[k] = dL_dshs[(size_t)idx * v.M * 3 + k];
        if (has_col) for (int k = 0; k < 3; k++) A.dcol[k] = dL_dcolors[3 * (size_t)idx + k];
    }
#pragma unroll 1
    for (int view = 0; view < bv.n; view++) {
        v.viewmatrix = bv.viewmatrix[view]; v.projmatrix = bv.projmatrix[view]; v.campos = bv.campos[view];
        SurfelGrad<DEG> G;
        surfel_backward<DEG>(v, idx, means3D, shs, colors_precomp, scales, rotations, transmat_precomp, bv.radii[view],
                             bv.geom[view], bv.pair_base[view], bv.pair_grad[view], bv.pair_valid[view], G);
        if (view == 0 && !accumulate) A = G;
        else A.add(G);
    }
    write_surfel_grad<DEG>(v, idx, A, has_sh, has_col, has_tm, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities,
                           dL_dscales, dL_drots, dL_dtransmat);
}

}  // namespace

int launch_preprocess_fwd(const ViewDev &v, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          const float *rotations, const float *transmat_precomp, StateView st,
                          ScratchView sc, int32_t *radii, hipStream_t s) {
    if (v.P == 0) return LARA2DGS_OK;
    const dim3 grid((v.P + 255) / 256), block(256);
    const int use_lds = v.tiles <= L2D_LDS_HIST_TILES;
    const size_t lds_bytes = use_lds ? (size_t)v.tiles * 4 : 0;
#define L2D_PRE(DEG)                                                                             \
    hipLaunchKernelGGL(preprocess_fwd_kernel<DEG>, grid, block, lds_bytes, s, v, means3D, shs,   \
                       colors_precomp, opacities, (const float2 *)scales,                        \
                       (const float4 *)rotations, transmat_precomp, st.geom, st.cullbox,         \
                       sc.rect, sc.tile_count, sc.block_tot, radii, use_lds)
    {
        L2D_PROF("preprocess_fwd", s);
        switch (colors_precomp ? 0 : v.deg) {
        case 0: L2D_PRE(0); break;
        case 1: L2D_PRE(1); break;
        case 2: L2D_PRE(2); break;
        default: L2D_PRE(3); break;
        }
    }
#undef L2D_PRE
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_preprocess_fwd_views(const ViewDev &v, int n, const ViewDev *views, const float *means3D, const float *shs,
                                const float *colors_precomp, const float *opacities, const float *scales,
                                const float *rotations, const float *transmat_precomp, const StateView *st,
                                const ScratchView *sc, int32_t *const *radii, hipStream_t s) {
    if (v.P == 0 || n <= 0) return LARA2DGS_OK;
    if (n > L2D_MAX_VIEWS) return LARA2DGS_E_INVALID;
    PreViews pv{};
    pv.n = n;
    for (int i = 0; i < n; i++) {
        pv.bg[i] = views[i].bg; pv.viewmatrix[i] = views[i].viewmatrix; pv.projmatrix[i] = views[i].projmatrix;
        pv.campos[i] = views[i].campos;
        pv.geom[i] = st[i].geom; pv.cullbox[i] = st[i].cullbox; pv.rect[i] = sc[i].rect;
        pv.tile_count[i] = sc[i].tile_count; pv.block_tot[i] = sc[i].block_tot; pv.radii[i] = radii[i];
    }
    const dim3 grid((unsigned)((v.P + 255) / 256) * (unsigned)n), block(256);
    const int use_lds = v.tiles <= L2D_LDS_HIST_TILES;
    const size_t lds_bytes = use_lds ? (size_t)v.tiles * 4 : 0;
#define L2D_PREV(DEG)                                                                                    \
    hipLaunchKernelGGL(preprocess_fwd_views_kernel<DEG>, grid, block, lds_bytes, s, v, pv, means3D, shs, \
                       colors_precomp, opacities, (const float2 *)scales, (const float4 *)rotations,     \
                       transmat_precomp, use_lds)
    {
        L2D_PROF("preprocess_fwd_views", s);
        switch (colors_precomp ? 0 : v.deg) {
        case 0: L2D_PREV(0); break;
        case 1: L2D_PREV(1); break;
        case 2: L2D_PREV(2); break;
        default: L2D_PREV(3); break;
        }
    }
#undef L2D_PREV
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_preprocess_bwd(const ViewDev &v, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *scales, const float *rotations,
                          const float *transmat_precomp, const int32_t *radii, StateView st,
                          ScratchView sc, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs,
                          float *dL_dcolors, float *dL_dopacities, float *dL_dscales,
                          float *dL_drotations, float *dL_dtransmat, hipStream_t s) {
    if (v.P == 0) return LARA2DGS_OK;
    const dim3 grid((v.P + 255) / 256), block(256);
#define L2D_PREB(DEG)                                                                            \
    hipLaunchKernelGGL(preprocess_bwd_kernel<DEG>, grid, block, 0, s, v, means3D, shs,           \
                       colors_precomp, (const float2 *)scales, (const float4 *)rotations,        \
                       transmat_precomp, radii, (const float4 *)st.geom, st.pair_base,           \
                       (const float4 *)sc.pair_grad, (const uint8_t *)sc.pair_valid,                 \
                       dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities,             \
                       (float2 *)dL_dscales, (float4 *)dL_drotations, dL_dtransmat)
    {
        L2D_PROF("preprocess_bwd", s);
        switch (colors_precomp ? 0 : v.deg) {
        case 0: L2D_PREB(0); break;
        case 1: L2D_PREB(1); break;
        case 2: L2D_PREB(2); break;
        default: L2D_PREB(3); break;
        }
    }
#undef L2D_PREB
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present,
                        hipStream_t s) {
    if (P == 0) return LARA2DGS_OK;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D,
                       viewmatrix, present);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_preprocess_bwd_views(const ViewDev &v, int n, const ViewDev *views, int accumulate, const float *means3D,
                                const float *shs, const float *colors_precomp, const float *scales, const float *rotations,
                                const float *transmat_precomp, const int32_t *const *radii, const StateView *st,
                                const ScratchView *sc, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs,
                                float *dL_dcolors, float *dL_dopacities, float *dL_dscales, float *dL_drotations,
                                float *dL_dtransmat, hipStream_t s) {
    if (v.P == 0 || n <= 0) return LARA2DGS_OK;
    if (n > L2D_MAX_VIEWS) return LARA2DGS_E_INVALID;
    BwdViews bv{};
    bv.n = n;
    for (int i = 0; i < n; i++) {
        bv.viewmatrix[i] = views[i].viewmatrix; bv.projmatrix[i] = views[i].projmatrix; bv.campos[i] = views[i].campos;
        bv.radii[i] = radii[i]; bv.geom[i] = st[i].geom; bv.pair_base[i] = st[i].pair_base;
        bv.pair_grad[i] = (const float4 *)sc[i].pair_grad; bv.pair_valid[i] = (const uint8_t *)sc[i].pair_valid;
    }
    const dim3 grid((v.P + 255) / 256), block(256);
#define L2D_PREBV(DEG)                                                                                         \
    hipLaunchKernelGGL(preprocess_bwd_views_kernel<DEG>, grid, block, 0, s, v, bv, accumulate, means3D, shs,   \
                       colors_precomp, (const float2 *)scales, (const float4 *)rotations, transmat_precomp,   \
                       dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, (float2 *)dL_dscales,      \
                       (float4 *)dL_drotations, dL_dtransmat)
    {
        L2D_PROF("preprocess_bwd_views", s);
        switch (colors_precomp ? 0 : v.deg) {
        case 0: L2D_PREBV(0); break;
        case 1: L2D_PREBV(1); break;
        case 2: L2D_PREBV(2); break;
        default: L2D_PREBV(3); break;
        }
    }
#undef L2D_PREBV
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
