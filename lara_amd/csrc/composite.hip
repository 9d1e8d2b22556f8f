// composite.hip -- front-to-back alpha compositing of a tile's sorted surfel list (forward) and the
// reverse traversal that produces per-surfel gradients (backward), for gfx950.
//
// Replaces the reference's (absent) `renderCUDA` forward / backward; semantics restated from the
// published 2DGS rasteriser; output contract pinned by lightning/renderer_2dgs.py:220-242
// (colour = C + T*bg; allmap = [sum w*depth, 1-T, sum w*normal (view space), median depth,
// distortion]).
//
// Structure (both directions).  One 256-thread workgroup per 16x16 tile (the binning contract),
// four wave64s = four 8x8 quadrants.  Inside a wave every group of 4 lanes (a DPP quad) owns a
// 2x2 pixel block and walks ITS OWN candidate list ("quad-SIMT"): at LaRa's statistics a surfel
// reaches alpha >= 1/255 on ~14 pixels of a tile, so with one candidate stream per wave only 11 of
// 64 lanes did useful work; per-quad streams need 2.5x fewer wave iterations.
// A tile's list is consumed 256 entries at a time in two phases:
//   phase S  (thread = list entry)  gather the surfel record through the sorted id list, turn it
//            into TILE-RELATIVE coefficients -- the ray/surfel intersection p = k x l is affine in
//            the pixel offset: p(lx,ly) = A + lx*B + ly*C with A = k0 x l0, B = Tw x l0,
//            C = k0 x Tw, k0/l0 = the reference's k/l at the tile origin -- and rasterise the
//            surfel's conservative alpha>=1/255 box onto the tile's 8x8 grid of 2x2 blocks (64-bit
//            mask).  Costs 1/64 of a wave-instruction per entry.
//   phase P  (lane = pixel)  per 64 entries each wave transposes the (entry x block) bit matrix
//            with 16 ballots, every quad keeps the mask of its block and walks only its set bits;
//            records are read from LDS with per-quad addresses.  Skipping is exact: an entry is
//            skipped for a block only if the reference would have skipped it for all 4 pixels
//            (alpha < 1/255), and list positions (`contributor` numbering) are kept.
// Backward adds: the 22 per-pixel partial derivatives of an entry (coefficient space) are reduced
// over the quad with a 2-step DPP reduce-scatter, accumulated per tile in LDS (ds_add_f32),
// transformed to dL/dT etc. by the entry's own thread (phase S2) and only then added to HBM: one
// atomic per (tile, surfel, component) instead of one per (pixel, surfel, component).
#include "common.h"

namespace {

constexpr int FWD_CHUNK = 512;  // list entries staged per round, forward
constexpr int BWD_CHUNK = 128;  // backward (its LDS also holds the per-wave result slices)
constexpr int REC4 = 6;        // float4 planes per staged entry (see stage_entry)
constexpr int ACC_STRIDE = 22; // floats per (wave, entry) result slot in the backward (21 used)

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// phase S: thread `e` stages list entry `pos` (if valid) as tile-relative coefficients
//   plane 0: A.xyz B.x   plane 1: B.yz C.xy   plane 2: C.z dx0 dy0 Tw.x
//   plane 3: Tw.yz opacity mask_lo   plane 4: normal.xyz r   plane 5: g b mask_hi -
//   mask bit (gy*8 + gx) <=> the 2x2 block (gx, gy) of the tile may see the surfel
// `id` and its cull box `cb` (minx, maxx, miny, maxy of {alpha >= 1/255}, conservative) were fetched
// one round ahead by the caller, so only the record gather sits on the round's critical path.
template <int CHUNK>
__device__ __forceinline__ void stage_entry(const float4 *__restrict__ geom, const uint32_t id,
                                            const float4 cb, const bool valid, const float X0,
                                            const float Y0, float4 *rec, uint32_t *ids,
                                            const int e = threadIdx.x) {
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0;
    uint32_t mask_lo = 0, mask_hi = 0;
    if (valid) {
        // block gx covers pixels X0+2gx, X0+2gx+1: overlap <=> minx <= X0+2gx+1 and maxx >= X0+2gx
        const int gx0 = (int)ceilf(fminf(fmaxf((cb.x - X0 - 1.f) * 0.5f, 0.f), 8.f));
        const int gx1 = (int)floorf(fminf(fmaxf((cb.y - X0) * 0.5f, -1.f), 7.f));
        const int gy0 = (int)ceilf(fminf(fmaxf((cb.z - Y0 - 1.f) * 0.5f, 0.f), 8.f));
        const int gy1 = (int)floorf(fminf(fmaxf((cb.w - Y0) * 0.5f, -1.f), 7.f));
        if (gx0 <= gx1 && gy0 <= gy1) {
            const uint32_t cols = ((1u << (gx1 - gx0 + 1)) - 1u) << gx0;  // 8 bits
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r >= gy0 && r <= gy1) mask_lo |= cols << (8 * r);
                if (r + 4 >= gy0 && r + 4 <= gy1) mask_hi |= cols << (8 * r);
            }
        }
    }
    if (mask_lo | mask_hi) {  // the 80-byte record is only fetched for surfels the tile can see
        const float4 *g = geom + (size_t)id * 5;
        const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
        const float Tu[3] = {g0.x, g0.y, g0.z}, Tv[3] = {g0.w, g1.x, g1.y}, Tw[3] = {g1.z, g1.w, g2.x};
        const float k0[3] = {X0 * Tw[0] - Tu[0], X0 * Tw[1] - Tu[1], X0 * Tw[2] - Tu[2]};
        const float l0[3] = {Y0 * Tw[0] - Tv[0], Y0 * Tw[1] - Tv[1], Y0 * Tw[2] - Tv[2]};
        float A[3], B[3], C[3];
        cross3(k0, l0, A);
        cross3(Tw, l0, B);
        cross3(k0, Tw, C);
        r0 = make_float4(A[0], A[1], A[2], B[0]);
        r1 = make_float4(B[1], B[2], C[0], C[1]);
        r2 = make_float4(C[2], g2.y - X0, g2.z - Y0, Tw[0]);
        r3 = make_float4(Tw[1], Tw[2], g2.w, __uint_as_float(mask_lo));
        r4 = make_float4(g3.x, g3.y, g3.z, g4.x);
        r5 = make_float4(g4.y, g4.z, __uint_as_float(mask_hi), 0.f);
    }
    rec[0 * CHUNK + e] = r0; rec[1 * CHUNK + e] = r1; rec[2 * CHUNK + e] = r2;
    rec[3 * CHUNK + e] = r3; rec[4 * CHUNK + e] = r4; rec[5 * CHUNK + e] = r5;
    if (ids) ids[e] = valid ? id : 0u;
}

struct Hit {
    float sx, sy, rz, depth, G, alpha, ddx, ddy;
    bool use3d;
};

// phase P: the four LDS planes an evaluation needs (kept in registers one entry ahead of use, so
// that the LDS latency -- and, in the backward, the ds_add_f64 queued in front -- is hidden)
struct EntryRec { float4 r0, r1, r2, r3; };
template <int CHUNK>
__device__ __forceinline__ EntryRec load_entry(const float4 *rec, const int j) {
    EntryRec e;
    e.r0 = rec[0 * CHUNK + j]; e.r1 = rec[1 * CHUNK + j]; e.r2 = rec[2 * CHUNK + j]; e.r3 = rec[3 * CHUNK + j];
    return e;
}

// evaluate a staged entry for this lane's pixel (lx, ly = offsets inside the tile)
__device__ __forceinline__ bool eval_rec(const EntryRec &e, const float lx, const float ly, Hit &h,
                                         float Tw[3], float &opa) {
    const float4 r0 = e.r0, r1 = e.r1, r2 = e.r2, r3 = e.r3;
    const float px = r0.x + lx * r0.w + ly * r1.z;
    const float py = r0.y + lx * r1.x + ly * r1.w;
    const float pz = r0.z + lx * r1.y + ly * r2.x;
    Tw[0] = r2.w; Tw[1] = r3.x; Tw[2] = r3.y; opa = r3.z;
    h.rz = __builtin_amdgcn_rcpf(pz);  // v_rcp_f32 (1 ulp); parity here is tolerance based
    h.sx = px * h.rz; h.sy = py * h.rz;
    const float rho3d = h.sx * h.sx + h.sy * h.sy;
    h.ddx = r2.y - lx; h.ddy = r2.z - ly;
    const float rho2d = FILTER_INV_SQUARE * (h.ddx * h.ddx + h.ddy * h.ddy);
    h.use3d = rho3d <= rho2d;
    const float rho = fminf(rho3d, rho2d);
    h.depth = h.use3d ? (h.sx * Tw[0] + h.sy * Tw[1]) + Tw[2] : Tw[2];
    const float power = -0.5f * rho;
    h.G = __expf(power);
    h.alpha = fminf(0.99f, opa * h.G);
    return (pz != 0.0f) & (h.depth >= NEAR_N) & !(power > 0.0f) & !(h.alpha < 1.0f / 255.0f);
}

// Transpose the (64 entries x 16 blocks of this wave) bit matrix: 16 ballots, then every lane keeps
// the 64-entry mask of its own quad.  `bm` = this lane's entry's 32-bit half mask (the half that
// holds the wave's quadrant rows), `bit0` = bit index of the wave's first block inside that half.
__device__ __forceinline__ unsigned long long quad_masks(const uint32_t bm, const int qx4, const int grp) {
    unsigned long long m = 0ull;
#pragma unroll
    for (int g = 0; g < 16; g++) {
        const int bit = (g >> 2) * 8 + qx4 + (g & 3);
        const unsigned long long b = __ballot((bm >> bit) & 1u);
        m = (grp == g) ? b : m;
    }
    return m;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// per-pixel compositing state; blend() is branch-free per lane (invalid lanes blend with alpha 0)
struct FwdPixel {
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dd = 0.f, M1 = 0.f, M2 = 0.f, distortion = 0.f, median_depth = 0.f;
    uint32_t last_contributor = 0, median_contributor = 0;
    bool done = false;

    template <int CHUNK>
    __device__ __forceinline__ void blend(const float4 *rec, const int j, const int base, bool valid,
                                          const Hit &h) {
        valid = valid && !done;
        const float test_T = T * (1.0f - h.alpha);
        const bool kill = valid && test_T < 0.0001f;  // would push T below 1e-4: not composited, pixel ends
        done = done || kill;
        valid = valid && !kill;
        if (__ballot(valid) == 0ull) return;
        const float4 r4 = rec[4 * CHUNK + j], r5 = rec[5 * CHUNK + j];
        const float a = valid ? h.alpha : 0.f;
        const float depth = valid ? h.depth : 1.0f;
        const float w = a * T;
        const float A = 1.0f - T;
        const float mm = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N * __builtin_amdgcn_rcpf(depth));
        distortion += (mm * mm * A + M2 - 2.0f * mm * M1) * w;
        Dd += depth * w;
        M1 += mm * w;
        M2 += mm * mm * w;
        const bool med = valid && T > 0.5f;
        median_depth = med ? depth : median_depth;
        median_contributor = med ? (uint32_t)(base + j + 1) : median_contributor;
        N0 += r4.x * w; N1 += r4.y * w; N2 += r4.z * w;
        C0 += r4.w * w; C1 += r5.x * w; C2 += r5.y * w;
        T = valid ? test_T : T;
        last_contributor = valid ? (uint32_t)(base + j + 1) : last_contributor;
    }
};

__global__ void __launch_bounds__(256)
composite_fwd_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ geom,
                     const uint32_t *__restrict__ tile_order, const float4 *__restrict__ cullbox,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                     float *__restrict__ out_color, float *__restrict__ out_allmap) {
    constexpr int CHUNK = FWD_CHUNK;
    __shared__ float4 rec[REC4 * CHUNK];
    const int tile = (v.dbg & 8u) ? (int)blockIdx.x : (int)tile_order[blockIdx.x];
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 2;  // quad = 2x2 pixel block; 4x4 quads per 8x8 quadrant
    const int lxi = (wave & 1) * 8 + (grp & 3) * 2 + (lane & 1);
    const int lyi = (wave >> 1) * 8 + (grp >> 2) * 2 + ((lane >> 1) & 1);
    const int pxi = tx * TILE + lxi, pyi = ty * TILE + lyi;
    const bool inside = pxi < v.W && pyi < v.H;
    const size_t HW = (size_t)v.H * v.W;
    const size_t pix = (size_t)pyi * v.W + pxi;
    const float lx = (float)lxi, ly = (float)lyi;
    const float X0 = (float)(tx * TILE), Y0 = (float)(ty * TILE);

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
    FwdPixel px;
    px.done = !inside;
    uint32_t *dbg_hdr = const_cast<uint32_t *>(header);
    const long long dbg_t0 = (v.dbg & 32u) ? (long long)__builtin_readcyclecounter() : 0ll;
    int dbg_rounds = 0;

    // software pipeline of the list walk: ids two rounds ahead, cull boxes one round ahead
    const int tid = threadIdx.x;
    constexpr int SPT = CHUNK / 256;  // list entries staged per thread and round
    uint32_t id1[SPT], id2[SPT];
    float4 cb1[SPT];
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        const int o = q * 256 + tid;
        id1[q] = o < total ? point_list[range.x + o] : 0u;
        id2[q] = CHUNK + o < total ? point_list[range.x + CHUNK + o] : 0u;
    }
#pragma unroll
    for (int q = 0; q < SPT; q++) cb1[q] = q * 256 + tid < total ? cullbox[id1[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < total; base += CHUNK) {
        if (__syncthreads_count(px.done) == 256) break;
        dbg_rounds++;
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const int o = q * 256 + tid;
            const uint32_t id0 = id1[q];
            const float4 cb0 = cb1[q];
            id1[q] = id2[q];
            id2[q] = base + 2 * CHUNK + o < total ? point_list[range.x + base + 2 * CHUNK + o] : 0u;
            cb1[q] = base + CHUNK + o < total ? cullbox[id1[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
            stage_entry<CHUNK>(geom, id0, cb0, base + o < total, X0, Y0, rec, nullptr, o);
        }
        __syncthreads();
        if (__ballot(!px.done) == 0ull) continue;  // this quadrant is finished; keep serving barriers
#pragma unroll 1
        for (int sub = 0; sub < CHUNK; sub += 64) {
            if (base + sub >= total) break;
            const float4 *mrec = rec + ((wave >> 1) ? 5 : 3) * CHUNK + min(sub + lane, CHUNK - 1);
            const uint32_t bm = sub + lane < CHUNK ? __float_as_uint((wave >> 1) ? mrec->z : mrec->w) : 0u;
            if (__ballot((bm & (0x0f0f0f0fu << ((wave & 1) * 4))) != 0u) == 0ull) continue;
            const unsigned long long m = quad_masks(bm, (wave & 1) * 4, grp);
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                uint32_t mm = half ? (uint32_t)(m >> 32) : (uint32_t)m;
                const int jb = sub + 32 * half;
                // a quad whose four pixels are finished stops consuming its list
                if (((__ballot(px.done) >> (lane & ~3)) & 0xfull) == 0xfull) mm = 0u;
                // each quad pops its own next entries; two per trip (independent evaluations),
                // blended strictly in list order; records are fetched one trip ahead
                bool has0 = mm != 0u;
                int j0 = jb + (has0 ? __builtin_ctz(mm) : 0);
                mm &= mm - 1u;
                bool has1 = mm != 0u;
                int j1 = jb + (has1 ? __builtin_ctz(mm) : 0);
                mm &= mm - 1u;  // stays 0 when already empty
                EntryRec c0 = load_entry<CHUNK>(rec, j0), c1 = load_entry<CHUNK>(rec, j1);
                while (__ballot(has0) != 0ull) {
                    const bool n0 = mm != 0u;
                    const int k0 = jb + (n0 ? __builtin_ctz(mm) : 0);
                    mm &= mm - 1u;
                    const bool n1 = mm != 0u;
                    const int k1 = jb + (n1 ? __builtin_ctz(mm) : 0);
                    mm &= mm - 1u;
                    const EntryRec x0 = load_entry<CHUNK>(rec, k0), x1 = load_entry<CHUNK>(rec, k1);
                    Hit h0, h1;
                    float Tw0[3], Tw1[3], opa0, opa1;
                    const bool e0 = eval_rec(c0, lx, ly, h0, Tw0, opa0) && has0;
                    const bool e1 = eval_rec(c1, lx, ly, h1, Tw1, opa1) && has1;
                    if (v.dbg & 4u) {  // statistics: quad candidates, valid (pixel, entry) pairs
                        atomicAdd(&dbg_hdr[4], ((lane & 3) == 0) ? (unsigned)has0 + (unsigned)has1 : 0u);
                        atomicAdd(&dbg_hdr[5], (unsigned)e0 + (unsigned)e1);
                        if (lane == 0) atomicAdd(&dbg_hdr[6], 1u);
                    }
                    px.blend<CHUNK>(rec, j0, base, e0, h0);
                    px.blend<CHUNK>(rec, j1, base, e1, h1);
                    c0 = x0; c1 = x1; j0 = k0; j1 = k1; has0 = n0; has1 = n1;
                    if (((__ballot(px.done) >> (lane & ~3)) & 0xfull) == 0xfull) { mm = 0u; has0 = false; has1 = false; }
                }
            }
        }
    }
    const float T = px.T;

    if (inside) {
        final_T[pix] = T;
        final_T[pix + HW] = px.M1;
        final_T[pix + 2 * HW] = px.M2;
        n_contrib[pix] = px.last_contributor;
        n_contrib[pix + HW] = px.median_contributor;
        out_color[0 * HW + pix] = px.C0 + T * v.bg[0];
        out_color[1 * HW + pix] = px.C1 + T * v.bg[1];
        out_color[2 * HW + pix] = px.C2 + T * v.bg[2];
        out_allmap[0 * HW + pix] = px.Dd;
        out_allmap[1 * HW + pix] = 1.0f - T;
        out_allmap[2 * HW + pix] = px.N0;
        out_allmap[3 * HW + pix] = px.N1;
        out_allmap[4 * HW + pix] = px.N2;
        out_allmap[5 * HW + pix] = px.median_depth;
        out_allmap[6 * HW + pix] = px.distortion;
        if ((v.dbg & 32u) && lane == 0) {  // per-wave cycle count / rounds, debug only (clobbers two outputs)
            out_allmap[6 * HW + pix] = (float)((long long)__builtin_readcyclecounter() - dbg_t0);
            out_allmap[5 * HW + pix] = (float)dbg_rounds;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// quad (4-lane) reduce-scatter of 22 per-lane values with DPP quad_perm
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_full(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// step 1 (xor 1): lane keeps the 11 values of its own parity; step 2 (xor 2): keeps every second of
// those.  Result: r[i] = quad sum of g[4i + (lane&3)], i = 0..4; r[5] = quad sum of g[20 + (lane&1)]
// (valid in lanes with bit 1 clear).
__device__ __forceinline__ void quad_reduce_scatter(const float g[22], float r[6], const int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float v1[11];
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const float mine = b0 ? g[2 * i + 1] : g[2 * i];
        const float send = b0 ? g[2 * i] : g[2 * i + 1];
        v1[i] = mine + dpp_full<0xB1>(send);  // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const float mine = b1 ? v1[2 * i + 1] : v1[2 * i];
        const float send = b1 ? v1[2 * i] : v1[2 * i + 1];
        r[i] = mine + dpp_full<0x4E>(send);  // quad_perm [2,3,0,1]
    }
    r[5] = v1[10] + dpp_full<0x4E>(v1[10]);
}

// ------------------------------------------------------------------------------------------------
// packed butterfly reduction of 21 per-lane values over the 64 lanes of a wave
// (v_permlane32_swap / v_permlane16_swap / DPP: ~55 VALU ops instead of 126 for 21 plain reductions)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float f_of(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ unsigned u_of(float f) { return __float_as_uint(f); }

// lanes 0..31 <- a[l] + a[l+32] ; lanes 32..63 <- b[l-32] + b[l]
__device__ __forceinline__ float pair32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(u_of(a), u_of(b), false, false);
    return f_of(r[0]) + f_of(r[1]);
}
// rows 0,2 <- a summed over (row, row+1) ; rows 1,3 <- b summed over (row-1, row)
__device__ __forceinline__ float pair16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(u_of(a), u_of(b), false, false);
    return f_of(r[0]) + f_of(r[1]);
}
// lanes with bit3 clear <- a[l] + a[l^8] ; bit3 set <- b[l] + b[l^8]
__device__ __forceinline__ float pair8(float a, float b, bool bit3) {
    const float t = a + dpp_full<0x128>(a), u = b + dpp_full<0x128>(b);  // row_ror:8
    return bit3 ? u : t;
}
// lanes with bit2 clear <- a[l] + a[l+4] ; bit2 set <- b[l] + b[l-4]
__device__ __forceinline__ float pair4(float a, float b, bool bit2) {
    int y = __builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x104, 0xf, 0x5, false);  // row_shl:4 -> banks 0,2
    y = __builtin_amdgcn_update_dpp(y, __float_as_int(b), 0x114, 0xf, 0xa, false);      // row_shr:4 -> banks 1,3
    return (bit2 ? b : a) + __int_as_float(y);
}
// lanes with bit1 clear <- a[l] + a[l^2] ; bit1 set <- b[l] + b[l^2]
__device__ __forceinline__ float pair2(float a, float b, bool bit1) {
    const float t = a + dpp_full<0x4E>(a), u = b + dpp_full<0x4E>(b);  // quad_perm [2,3,0,1]
    return bit1 ? u : t;
}
__device__ __forceinline__ float fold1(float a) { return a + dpp_full<0xB1>(a); }  // quad_perm [1,0,3,2]

// After the call, lane l holds the full 64-lane sum of value slot_of_lane(l) (see below).
__device__ __forceinline__ float butterfly21(const float v[21], const int lane) {
    float r[11];
#pragma unroll
    for (int i = 0; i < 10; i++) r[i] = pair32(v[2 * i], v[2 * i + 1]);
    r[10] = pair32(v[20], v[20]);
    float q[6];
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = pair16(r[2 * i], r[2 * i + 1]);
    q[5] = pair16(r[10], r[10]);
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    const float o0 = pair8(q[0], q[1], b3), o1 = pair8(q[2], q[3], b3), o2 = pair8(q[4], q[5], b3);
    const float n0 = pair4(o0, o1, b2), n1 = pair4(o2, o2, b2);
    return fold1(pair2(n0, n1, b1));
}
// which of the 21 values lane l ends up with (-1: duplicate holder, must not write)
__device__ __forceinline__ int slot_of_lane(const int l) {
    if (l & 1) return -1;
    const int b1 = (l >> 1) & 1, b2 = (l >> 2) & 1, b3 = (l >> 3) & 1, b4 = (l >> 4) & 1, b5 = (l >> 5) & 1;
    int o;  // index among o0..o2
    if (b1) { if (b2) return -1; o = 2; } else o = b2;
    const int q = 2 * o + b3;  // index among q0..q5
    if (q == 5) return (b4 | b5) ? -1 : 20;
    const int r = 2 * q + b4;  // index among r0..r9
    return 2 * r + b5;
}

__device__ __forceinline__ void atomic_add_f32(float *p, float x) {
    __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
composite_bwd_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ geom,
                     const uint32_t *__restrict__ tile_order, const float4 *__restrict__ cullbox,
                     const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                     const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dallmap,
                     float4 *__restrict__ pair_grad, uint32_t *__restrict__ pair_valid) {
    constexpr int CHUNK = BWD_CHUNK;
    // Per-wave result slices instead of LDS atomics: a wave visits an entry at most once per round,
    // so it can park the entry's 21 sums with a plain ds_write; phase S2 adds the (<= 4) slices in
    // a fixed order.  (LDS float atomics are slow on gfx950: ds_add_f32 169 / ds_add_f64 18 cycles
    // per wave instruction with a long latency -- tools/ubench/lds_atomic.hip.)
    __shared__ float4 rec[REC4 * CHUNK];
    __shared__ float acc[4 * CHUNK * ACC_STRIDE];  // [wave][entry][slot]
    __shared__ unsigned long long touched[4][(CHUNK + 63) / 64];
    __shared__ uint32_t s_id[CHUNK];
    __shared__ uint32_t s_maxc;
    if (header[1]) return;
    const int tile = (v.dbg & 8u) ? (int)blockIdx.x : (int)tile_order[blockIdx.x];
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 2;
    const int slot = slot_of_lane(lane);
    const int lxi = (wave & 1) * 8 + (grp & 3) * 2 + (lane & 1);
    const int lyi = (wave >> 1) * 8 + (grp >> 2) * 2 + ((lane >> 1) & 1);
    const int pxi = tx * TILE + lxi, pyi = ty * TILE + lyi;
    const bool inside = pxi < v.W && pyi < v.H;
    const size_t HW = (size_t)v.H * v.W;
    const size_t pix = inside ? (size_t)pyi * v.W + pxi : 0;
    const float lx = (float)lxi, ly = (float)lyi;
    const float X0 = (float)(tx * TILE), Y0 = (float)(ty * TILE);
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

    // back to front, CHUNK entries at a time: chunk c covers list positions [lo, lo + cnt)
    const int nchunks = (total + CHUNK - 1) / CHUNK;
    // software pipeline of the list walk (threads 0..CHUNK-1 stage): ids two rounds ahead, cull
    // boxes one round ahead
    const int tid = threadIdx.x;
    const bool stager = tid < CHUNK;
    auto in_chunk = [&](int c) { return stager && c >= 0 && c * CHUNK + tid < total; };
    uint32_t id1 = in_chunk(nchunks - 1) ? point_list[range.x + (nchunks - 1) * CHUNK + tid] : 0u;
    uint32_t id2 = in_chunk(nchunks - 2) ? point_list[range.x + (nchunks - 2) * CHUNK + tid] : 0u;
    float4 cb1 = in_chunk(nchunks - 1) ? cullbox[id1] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = nchunks - 1; c >= 0; c--) {
        const int lo = c * CHUNK;
        const int cnt = min(CHUNK, total - lo);
        const uint32_t id0 = id1;
        const float4 cb0 = cb1;
        id1 = id2;
        id2 = in_chunk(c - 2) ? point_list[range.x + (c - 2) * CHUNK + tid] : 0u;
        cb1 = in_chunk(c - 1) ? cullbox[id1] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();  // previous chunk's phase S2 is done with rec / acc / s_id
        if (stager) stage_entry<CHUNK>(geom, id0, cb0, tid < cnt, X0, Y0, rec, s_id);
        if (threadIdx.x < 4 * ((CHUNK + 63) / 64)) (&touched[0][0])[threadIdx.x] = 0ull;
        __syncthreads();

#pragma unroll 1
        for (int sub = ((cnt - 1) >> 6) << 6; sub >= 0; sub -= 64) {
            // The backward keeps ONE candidate stream per wave (8x8 quadrant): its per-entry sums
            // must be reduced over all pixels anyway, and a wave-wide butterfly + one 22-lane
            // ds_add_f64 is far cheaper than per-quad LDS atomics (which collide on the same entry).
            const float4 *mrec = rec + ((wave >> 1) ? 5 : 3) * CHUNK + min(sub + lane, CHUNK - 1);
            const uint32_t bm = sub + lane < CHUNK ? __float_as_uint((wave >> 1) ? mrec->z : mrec->w) : 0u;
            unsigned long long m = __ballot((bm & (0x0f0f0f0fu << ((wave & 1) * 4))) != 0u);
            if (m == 0ull) continue;
            unsigned long long tmask = 0ull;  // entries of this sub-chunk this wave produced sums for
            int jn = sub + 63 - __builtin_clzll(m);
            m &= ~(1ull << (jn - sub));
            EntryRec cur = load_entry<CHUNK>(rec, jn);
            float4 cur4 = rec[4 * CHUNK + jn], cur5 = rec[5 * CHUNK + jn];
            bool more = true;
            while (more) {
                const int j = jn;
                const EntryRec ent = cur;
                const float4 r4 = cur4, r5 = cur5;
                more = m != 0ull;
                if (more) {  // fetch the next entry before this one's ds_add_f64 enters the LDS queue
                    jn = sub + 63 - __builtin_clzll(m);
                    m &= ~(1ull << (jn - sub));
                    cur = load_entry<CHUNK>(rec, jn);
                    cur4 = rec[4 * CHUNK + jn]; cur5 = rec[5 * CHUNK + jn];
                }
                const uint32_t contributor = (uint32_t)(lo + j);  // 0-based list position
                Hit h;
                float Tw[3], opa;
                const bool active = eval_rec(ent, lx, ly, h, Tw, opa) && contributor < last_contributor;
                if (__ballot(active) == 0ull) continue;

                float g[21];
#pragma unroll
                for (int k = 0; k < 21; k++) g[k] = 0.f;
                if (active) {
                    const float nrm[3] = {r4.x, r4.y, r4.z}, rgb[3] = {r4.w, r5.x, r5.y};
                    const float alpha = h.alpha, G = h.G, c_d = h.depth;
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * inv_1ma;
                    const float w = alpha * T;
                    float dL_dalpha = 0.0f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = rgb[ch];
                        dL_dalpha += (rgb[ch] - accum_rec[ch]) * dpix[ch];
                        g[18 + ch] = w * dpix[ch];
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
                        last_normal[ch] = nrm[ch];
                        dL_dalpha += (nrm[ch] - accum_normal_rec[ch]) * dnrm[ch];
                        g[14 + ch] = w * dnrm[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;

                    const float dL_dG = opa * dL_dalpha;
                    dL_dz += w * dL_ddepth;
                    // depth = s . Tw.xy + Tw.z (the published backward uses this form in both branches)
                    g[9] = dL_dz * h.sx; g[10] = dL_dz * h.sy; g[11] = dL_dz;
                    if (h.use3d) {
                        const float dL_dsx = dL_dG * -G * h.sx + dL_dz * Tw[0];
                        const float dL_dsy = dL_dG * -G * h.sy + dL_dz * Tw[1];
                        const float dpx = dL_dsx * h.rz, dpy = dL_dsy * h.rz;
                        const float dpz = -(dpx * h.sx + dpy * h.sy);
                        g[0] = dpx; g[1] = dpy; g[2] = dpz;
                        g[3] = lx * dpx; g[4] = lx * dpy; g[5] = lx * dpz;
                        g[6] = ly * dpx; g[7] = ly * dpy; g[8] = ly * dpz;
                    } else {
                        g[12] = dL_dG * (-G * FILTER_INV_SQUARE * h.ddx);
                        g[13] = dL_dG * (-G * FILTER_INV_SQUARE * h.ddy);
                    }
                    g[17] = G * dL_dalpha;
                }
                if (v.dbg & 2u) { float z = 0.f; for (int k = 0; k < 21; k++) z += g[k]; if (z == 123.456f) acc[0] = z; continue; }
                const float sred = butterfly21(g, lane);
                if (v.dbg & 16u) { if (sred == 123.456f) acc[0] = sred; continue; }
                if (slot >= 0) acc[(wave * CHUNK + j) * ACC_STRIDE + slot] = sred;  // plain ds_write, 21 lanes
                tmask |= 1ull << (j - sub);
            }
            if (lane == 0) touched[wave][sub >> 6] = tmask;
        }
        __syncthreads();

        // phase S2: thread e turns its entry's 21 coefficient-space sums into dL/d(Tu,Tv,Tw,...)
        const int e = threadIdx.x;
        bool any = false;
        float sacc[21];
#pragma unroll
        for (int k = 0; k < 21; k++) sacc[k] = 0.f;
        if (e < cnt) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if ((touched[w][e >> 6] >> (e & 63)) & 1ull) {
                    any = true;
                    const float *src = acc + (w * CHUNK + e) * ACC_STRIDE;
#pragma unroll
                    for (int k = 0; k < 21; k++) sacc[k] += src[k];
                }
            }
        }
        // one 80-byte gradient row per touched (tile, surfel) pair, written exactly once, coalesced;
        // a bitmap marks the rows that exist.  preprocess_bwd gathers each surfel's rows (no atomics
        // on gradients, fixed summation order).
        const uint32_t p = range.x + (uint32_t)(lo + e);
        if (any) {
            const uint32_t id = s_id[e];
            const float4 *gm = geom + (size_t)id * 5;
            const float4 g0 = gm[0], g1 = gm[1], g2 = gm[2];
            const float Tu[3] = {g0.x, g0.y, g0.z}, Tv[3] = {g0.w, g1.x, g1.y}, Tw[3] = {g1.z, g1.w, g2.x};
            const float k0[3] = {X0 * Tw[0] - Tu[0], X0 * Tw[1] - Tu[1], X0 * Tw[2] - Tu[2]};
            const float l0[3] = {Y0 * Tw[0] - Tv[0], Y0 * Tw[1] - Tv[1], Y0 * Tw[2] - Tv[2]};
            const float a[3] = {sacc[0], sacc[1], sacc[2]}, b[3] = {sacc[3], sacc[4], sacc[5]},
                        cc[3] = {sacc[6], sacc[7], sacc[8]};
            // A = k0 x l0, B = Tw x l0, C = k0 x Tw ; for y = u x v: dL/du = v x dL/dy, dL/dv = dL/dy x u
            float t1[3], t2[3], dk0[3], dl0[3], dTw[3];
            cross3(l0, a, t1); cross3(Tw, cc, t2);
            for (int i = 0; i < 3; i++) dk0[i] = t1[i] + t2[i];
            cross3(a, k0, t1); cross3(b, Tw, t2);
            for (int i = 0; i < 3; i++) dl0[i] = t1[i] + t2[i];
            cross3(l0, b, t1); cross3(cc, k0, t2);
            for (int i = 0; i < 3; i++) dTw[i] = t1[i] + t2[i] + X0 * dk0[i] + Y0 * dl0[i] + sacc[9 + i];
            float4 *row = pair_grad + (size_t)p * (GRAD_F / 4);
            row[0] = make_float4(-dk0[0], -dk0[1], -dk0[2], -dl0[0]);
            row[1] = make_float4(-dl0[1], -dl0[2], dTw[0], dTw[1]);
            row[2] = make_float4(dTw[2], sacc[12], sacc[13], sacc[14]);
            row[3] = make_float4(sacc[15], sacc[16], sacc[17], sacc[18]);
            row[4] = make_float4(sacc[19], sacc[20], 0.f, 0.f);
        }
        // publish the wave's 64 validity bits with at most three atomic ORs
        const unsigned long long bits = __ballot(any);
        if (bits) {
            const uint32_t p0 = range.x + (uint32_t)(lo + (e & ~63));  // position of lane 0's entry
            const uint32_t sh = p0 & 31u, w0 = p0 >> 5;
            const int l = e & 63;
            uint32_t word = 0;
            if (l == 0) word = (uint32_t)(bits << sh);
            else if (l == 1) word = (uint32_t)(bits >> (32u - sh));
            else if (l == 2) word = sh ? (uint32_t)(bits >> (64u - sh)) : 0u;
            if (l < 3 && word) atomicOr(&pair_valid[w0 + l], word);
        }
    }
}

// reduce-scatter self-test: in [64][22] (lane major) -> out [16 quads][22] quad sums
__global__ void __launch_bounds__(64)
selftest_butterfly_kernel(const float *__restrict__ in, float *__restrict__ out) {
    const int lane = threadIdx.x, q4 = lane & 3;
    float g[22], r[6];
    for (int k = 0; k < 22; k++) g[k] = in[lane * 22 + k];
    quad_reduce_scatter(g, r, lane);
    float *dst = out + (lane >> 2) * 22 + q4;
    for (int i = 0; i < 5; i++) dst[4 * i] = r[i];
    if (q4 < 2) dst[20] = r[5];
}

}  // namespace

int launch_composite_fwd(const ViewDev &v, StateView st, float *out_color, float *out_allmap,
                         hipStream_t s) {
    {
        L2D_PROF("composite_fwd", s);
        hipLaunchKernelGGL(composite_fwd_kernel, dim3(v.tiles), dim3(256), 0, s, v, st.header, st.ranges,
                           st.point_list, (const float4 *)st.geom, st.tile_order,
                           (const float4 *)st.cullbox, st.final_T, st.n_contrib, out_color, out_allmap);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_composite_bwd(const ViewDev &v, StateView st, ScratchView sc, const float *dL_dcolor,
                         const float *dL_dallmap, hipStream_t s) {
    {
        L2D_PROF("composite_bwd", s);
        hipLaunchKernelGGL(composite_bwd_kernel, dim3(v.tiles), dim3(256), 0, s, v, st.header, st.ranges,
                           st.point_list, (const float4 *)st.geom, st.tile_order,
                           (const float4 *)st.cullbox, st.final_T, st.n_contrib, dL_dcolor, dL_dallmap,
                           sc.pair_grad, sc.pair_valid);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_selftest_butterfly(const float *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL(selftest_butterfly_kernel, dim3(1), dim3(64), 0, s, in, out);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
