// composite.hip -- front-to-back alpha compositing of a tile's sorted surfel list (forward) and the
// reverse traversal that produces per-surfel gradients (backward), for gfx950.
//
// Replaces the reference's (absent) `renderCUDA` forward / backward; semantics restated from the
// published 2DGS rasteriser; output contract pinned by lightning/renderer_2dgs.py:220-242
// (colour = C + T*bg; allmap = [sum w*depth, 1-T, sum w*normal (view space), median depth,
// distortion]).
//
// Mapping: one 256-thread workgroup per 16x16 tile; each of its 4 wave64s owns an 8x8 pixel
// quadrant (lane = pixel), which keeps a wave's footprint compact for wave-level early-out.
// A tile's splat records (20 floats, gathered through the sorted id list) are staged through LDS
// 256 at a time as five float4 planes; every lane then reads the same record (LDS broadcast).
#include "common.h"

namespace {

constexpr int CHUNK = 256;

struct Splat {
    float Tu[3], Tv[3], Tw[3], xy[2], opa, nrm[3], rgb[3];
};

__device__ __forceinline__ Splat load_splat(const float4 *s0, const float4 *s1, const float4 *s2,
                                            const float4 *s3, const float4 *s4, int j) {
    const float4 a = s0[j], b = s1[j], c = s2[j], d = s3[j], e = s4[j];
    Splat s;
    s.Tu[0] = a.x; s.Tu[1] = a.y; s.Tu[2] = a.z;
    s.Tv[0] = a.w; s.Tv[1] = b.x; s.Tv[2] = b.y;
    s.Tw[0] = b.z; s.Tw[1] = b.w; s.Tw[2] = c.x;
    s.xy[0] = c.y; s.xy[1] = c.z; s.opa = c.w;
    s.nrm[0] = d.x; s.nrm[1] = d.y; s.nrm[2] = d.z;
    s.rgb[0] = e.x; s.rgb[1] = e.y; s.rgb[2] = e.z;
    return s;
}

struct Hit {
    float sx, sy, rho3d, rho2d, dx, dy, depth, G, alpha, pz;
    float k[3], l[3];
};

// Ray / surfel evaluation for one pixel.  Returns false when this list entry is skipped.
__device__ __forceinline__ bool eval_splat(const Splat &s, float pxf, float pyf, Hit &h) {
    h.k[0] = pxf * s.Tw[0] - s.Tu[0]; h.k[1] = pxf * s.Tw[1] - s.Tu[1]; h.k[2] = pxf * s.Tw[2] - s.Tu[2];
    h.l[0] = pyf * s.Tw[0] - s.Tv[0]; h.l[1] = pyf * s.Tw[1] - s.Tv[1]; h.l[2] = pyf * s.Tw[2] - s.Tv[2];
    const float px = h.k[1] * h.l[2] - h.k[2] * h.l[1];
    const float py = h.k[2] * h.l[0] - h.k[0] * h.l[2];
    const float pz = h.k[0] * h.l[1] - h.k[1] * h.l[0];
    if (pz == 0.0f) return false;
    h.pz = pz;
    const float rz = __builtin_amdgcn_rcpf(pz);  // v_rcp_f32 (1 ulp); parity is tolerance-based here
    h.sx = px * rz; h.sy = py * rz;
    h.rho3d = h.sx * h.sx + h.sy * h.sy;
    h.dx = s.xy[0] - pxf; h.dy = s.xy[1] - pyf;
    h.rho2d = FILTER_INV_SQUARE * (h.dx * h.dx + h.dy * h.dy);
    const float rho = fminf(h.rho3d, h.rho2d);
    h.depth = (h.rho3d <= h.rho2d) ? (h.sx * s.Tw[0] + h.sy * s.Tw[1]) + s.Tw[2] : s.Tw[2];
    if (h.depth < NEAR_N) return false;
    const float power = -0.5f * rho;
    if (power > 0.0f) return false;
    h.G = __expf(power);
    h.alpha = fminf(0.99f, s.opa * h.G);
    if (h.alpha < 1.0f / 255.0f) return false;
    return true;
}

__device__ __forceinline__ void stage_chunk(const uint32_t *__restrict__ point_list,
                                            const float4 *__restrict__ geom, uint32_t pos,
                                            bool valid, float4 *s0, float4 *s1, float4 *s2,
                                            float4 *s3, float4 *s4) {
    if (valid) {
        const uint32_t id = point_list[pos];
        const float4 *g = geom + (size_t)id * 5;
        const int t = threadIdx.x;
        s0[t] = g[0]; s1[t] = g[1]; s2[t] = g[2]; s3[t] = g[3]; s4[t] = g[4];
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
composite_fwd_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ geom,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                     float *__restrict__ out_color, float *__restrict__ out_allmap) {
    __shared__ float4 s0[CHUNK], s1[CHUNK], s2[CHUNK], s3[CHUNK], s4[CHUNK];
    const int tile = blockIdx.x;
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pxi = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < v.W && pyi < v.H;
    const size_t HW = (size_t)v.H * v.W;
    const size_t pix = (size_t)pyi * v.W + pxi;
    const float pxf = (float)pxi, pyf = (float)pyi;

    if (header[1]) {  // binning capacity exceeded: make the failure loud in the data
        if (inside) {
            const float qnan = __uint_as_float(0x7fc00000u);
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = qnan;
            for (int ch = 0; ch < 7; ch++) out_allmap[ch * HW + pix] = qnan;
            final_T[pix] = qnan; final_T[pix + HW] = qnan; final_T[pix + 2 * HW] = qnan;
            n_contrib[pix] = 0; n_contrib[pix + HW] = 0;
        }
        return;
    }

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dd = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0;

    for (int base = 0; base < total; base += CHUNK) {
        if (__syncthreads_count(done) == 256) break;
        stage_chunk(point_list, geom, range.x + base + threadIdx.x, base + (int)threadIdx.x < total,
                    s0, s1, s2, s3, s4);
        __syncthreads();
        const int cnt = min(CHUNK, total - base);
        for (int j = 0; j < cnt && !done; j++) {
            const Splat s = load_splat(s0, s1, s2, s3, s4, j);
            Hit h;
            if (!eval_splat(s, pxf, pyf, h)) continue;
            const float test_T = T * (1.0f - h.alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            const float w = h.alpha * T;
            const float A = 1.0f - T;
            const float m = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N * __builtin_amdgcn_rcpf(h.depth));
            distortion += (m * m * A + M2 - 2.0f * m * M1) * w;
            Dd += h.depth * w;
            M1 += m * w;
            M2 += m * m * w;
            if (T > 0.5f) { median_depth = h.depth; median_contributor = (uint32_t)(base + j + 1); }
            N0 += s.nrm[0] * w; N1 += s.nrm[1] * w; N2 += s.nrm[2] * w;
            C0 += s.rgb[0] * w; C1 += s.rgb[1] * w; C2 += s.rgb[2] * w;
            T = test_T;
            last_contributor = (uint32_t)(base + j + 1);
        }
    }

    if (inside) {
        final_T[pix] = T;
        final_T[pix + HW] = M1;
        final_T[pix + 2 * HW] = M2;
        n_contrib[pix] = last_contributor;
        n_contrib[pix + HW] = median_contributor;
        out_color[0 * HW + pix] = C0 + T * v.bg[0];
        out_color[1 * HW + pix] = C1 + T * v.bg[1];
        out_color[2 * HW + pix] = C2 + T * v.bg[2];
        out_allmap[0 * HW + pix] = Dd;
        out_allmap[1 * HW + pix] = 1.0f - T;
        out_allmap[2 * HW + pix] = N0;
        out_allmap[3 * HW + pix] = N1;
        out_allmap[4 * HW + pix] = N2;
        out_allmap[5 * HW + pix] = median_depth;
        out_allmap[6 * HW + pix] = distortion;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// wave64 sum via DPP: row_shr 1,2,4,8 inside each row of 16, then row_bcast15 / row_bcast31; the
// total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_step<0x118, 0xf>(v);  // row_shr:8
    v = dpp_step<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_step<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ void atomic_add_f32(float *p, float x) {
    __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256)
composite_bwd_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ geom,
                     const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                     const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dallmap,
                     float *__restrict__ grad) {
    __shared__ float4 s0[CHUNK], s1[CHUNK], s2[CHUNK], s3[CHUNK], s4[CHUNK];
    __shared__ uint32_t s_id[CHUNK];
    __shared__ uint32_t s_maxc;
    if (header[1]) return;
    const int tile = blockIdx.x;
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pxi = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < v.W && pyi < v.H;
    const size_t HW = (size_t)v.H * v.W;
    const size_t pix = inside ? (size_t)pyi * v.W + pxi : 0;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint2 range = ranges[tile];

    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    const uint32_t median_contributor = inside ? n_contrib[pix + HW] : 0u;
    float dpix[3] = {0.f, 0.f, 0.f}, dnrm[3] = {0.f, 0.f, 0.f};
    float dL_ddepth = 0.f, dL_daccum = 0.f, dL_dmedian = 0.f, dL_dreg = 0.f;
    float final_D = 0.f, final_D2 = 0.f;
    if (inside) {
        for (int ch = 0; ch < 3; ch++) dpix[ch] = dL_dcolor[ch * HW + pix];
        dL_ddepth = dL_dallmap[0 * HW + pix];
        dL_daccum = dL_dallmap[1 * HW + pix];
        for (int ch = 0; ch < 3; ch++) dnrm[ch] = dL_dallmap[(2 + ch) * HW + pix];
        dL_dmedian = dL_dallmap[5 * HW + pix];
        dL_dreg = dL_dallmap[6 * HW + pix];
        final_D = final_T[pix + HW];
        final_D2 = final_T[pix + 2 * HW];
    }
    const float final_A = 1.0f - T_final;
    const float bg_dot_dpixel = v.bg[0] * dpix[0] + v.bg[1] * dpix[1] + v.bg[2] * dpix[2];

    float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f};
    float accum_normal_rec[3] = {0.f, 0.f, 0.f}, last_normal[3] = {0.f, 0.f, 0.f};
    float accum_depth_rec = 0.f, accum_alpha_rec = 0.f, last_depth = 0.f, last_alpha = 0.f;
    float last_dL_dT = 0.f;

    // the tile only needs entries [0, max over pixels of last_contributor)
    if (threadIdx.x == 0) s_maxc = 0;
    __syncthreads();
    atomicMax(&s_maxc, last_contributor);
    __syncthreads();
    const int total = (int)s_maxc;

    // back to front, CHUNK entries at a time: chunk c covers list positions [hi - CHUNK, hi)
    for (int hi = total; hi > 0; hi -= CHUNK) {
        const int lo = max(0, hi - CHUNK);
        const int cnt = hi - lo;
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = point_list[range.x + lo + threadIdx.x];
            s_id[threadIdx.x] = id;
        }
        stage_chunk(point_list, geom, range.x + lo + threadIdx.x, (int)threadIdx.x < cnt, s0, s1, s2, s3, s4);
        __syncthreads();
        for (int j = cnt - 1; j >= 0; j--) {
            const uint32_t contributor = (uint32_t)(lo + j);  // 0-based position in the list
            const bool in_range = contributor < last_contributor;
            const Splat s = load_splat(s0, s1, s2, s3, s4, j);
            Hit h;
            const bool active = in_range && eval_splat(s, pxf, pyf, h);
            if (__ballot(active) == 0ull) continue;  // wave-uniform skip

            float g[18];
#pragma unroll
            for (int k = 0; k < 18; k++) g[k] = 0.f;
            if (active) {
                const float alpha = h.alpha, G = h.G, c_d = h.depth;
                const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                T = T * inv_1ma;
                const float w = alpha * T;
                float dL_dalpha = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float c = s.rgb[ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum_rec[ch]) * dpix[ch];
                    g[15 + ch] = w * dpix[ch];
                }
                float dL_dz = 0.0f, dL_dweight = 0.0f;
                const float inv_cd = __builtin_amdgcn_rcpf(c_d);
                const float m_d = FAR_N / (FAR_N - NEAR_N) * (1.f - NEAR_N * inv_cd);
                const float dmd_dd = (FAR_N * NEAR_N) / (FAR_N - NEAR_N) * inv_cd * inv_cd;
                if (contributor + 1 == median_contributor) dL_dz += dL_dmedian;
                dL_dweight += (final_D2 + m_d * m_d * final_A - 2.f * m_d * final_D) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + (1.f - alpha) * last_dL_dT;
                const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;

                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                accum_alpha_rec = last_alpha * 1.0f + (1.f - last_alpha) * accum_alpha_rec;
                dL_dalpha += (1.f - accum_alpha_rec) * dL_daccum;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                    last_normal[ch] = s.nrm[ch];
                    dL_dalpha += (s.nrm[ch] - accum_normal_rec[ch]) * dnrm[ch];
                    g[11 + ch] = alpha * T * dnrm[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;

                const float dL_dG = s.opa * dL_dalpha;
                dL_dz += alpha * T * dL_ddepth;

                if (h.rho3d <= h.rho2d) {
                    const float dL_dsx = dL_dG * -G * h.sx + dL_dz * s.Tw[0];
                    const float dL_dsy = dL_dG * -G * h.sy + dL_dz * s.Tw[1];
                    const float inv_pz = __builtin_amdgcn_rcpf(h.pz);
                    const float dsx_pz = dL_dsx * inv_pz, dsy_pz = dL_dsy * inv_pz;
                    const float dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * h.sx + dsy_pz * h.sy);
                    const float dkx = h.l[1] * dpz - h.l[2] * dpy;
                    const float dky = h.l[2] * dpx - h.l[0] * dpz;
                    const float dkz = h.l[0] * dpy - h.l[1] * dpx;
                    const float dlx = dpy * h.k[2] - dpz * h.k[1];
                    const float dly = dpz * h.k[0] - dpx * h.k[2];
                    const float dlz = dpx * h.k[1] - dpy * h.k[0];
                    g[0] = -dkx; g[1] = -dky; g[2] = -dkz;
                    g[3] = -dlx; g[4] = -dly; g[5] = -dlz;
                    g[6] = pxf * dkx + pyf * dlx + dL_dz * h.sx;
                    g[7] = pxf * dky + pyf * dly + dL_dz * h.sy;
                    g[8] = pxf * dkz + pyf * dlz + dL_dz;
                } else {
                    g[9] = dL_dG * (-G * FILTER_INV_SQUARE * h.dx);
                    g[10] = dL_dG * (-G * FILTER_INV_SQUARE * h.dy);
                    g[6] = h.sx * dL_dz;
                    g[7] = h.sy * dL_dz;
                    g[8] = dL_dz;
                }
                g[14] = G * dL_dalpha;
            }
            // wave reduction over the 64 pixels of this quadrant, then one atomic per component
#pragma unroll
            for (int k = 0; k < 18; k++) g[k] = wave_sum_to_lane63(g[k]);
            if (lane == 63) {
                float *dst = grad + (size_t)s_id[j] * GRAD_F;
#pragma unroll
                for (int k = 0; k < 18; k++) atomic_add_f32(dst + k, g[k]);
            }
        }
    }
}

}  // namespace

int launch_composite_fwd(const ViewDev &v, StateView st, float *out_color, float *out_allmap,
                         hipStream_t s) {
    {
        L2D_PROF("composite_fwd", s);
        hipLaunchKernelGGL(composite_fwd_kernel, dim3(v.tiles), dim3(256), 0, s, v, st.header, st.ranges,
                           st.point_list, (const float4 *)st.geom, st.final_T, st.n_contrib, out_color,
                           out_allmap);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_composite_bwd(const ViewDev &v, StateView st, ScratchView sc, const float *dL_dcolor,
                         const float *dL_dallmap, hipStream_t s) {
    {
        L2D_PROF("composite_bwd", s);
        hipLaunchKernelGGL(composite_bwd_kernel, dim3(v.tiles), dim3(256), 0, s, v, st.header, st.ranges,
                           st.point_list, (const float4 *)st.geom, st.final_T, st.n_contrib, dL_dcolor,
                           dL_dallmap, sc.grad);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
