// composite.hip -- front-to-back alpha compositing of a tile's sorted surfel list (forward) and the
// reverse traversal that produces per-surfel gradients (backward), for gfx950.
//
// Replaces the reference's (absent) `renderCUDA` forward / backward; semantics restated from the
// published 2DGS rasteriser; output contract pinned by lightning/renderer_2dgs.py:220-242
// (colour = C + T*bg; allmap = [sum w*depth, 1-T, sum w*normal (view space), median depth,
// distortion]).
//
// Structure (both directions).  256-thread workgroups over 16x16 tiles (the binning contract): one per
// tile in the forward, one per (tile, 512-entry segment of its list) in the backward.  Four wave64s
// = four 8x8 quadrants.  Inside a wave every group of 4 lanes (a DPP quad) owns a 2x2 pixel block
// and walks ITS OWN candidate list ("quad-SIMT"): at LaRa's statistics a surfel reaches
// alpha >= 1/255 on ~14 pixels of a tile, so with one candidate stream per wave only 11 of 64 lanes
// did useful work.
// A list is consumed in rounds (512 entries forward, windows of 128 backward) in two phases:
//   phase S  (thread = list entry)  gather the surfel record through the sorted id list, turn it
//            into TILE-RELATIVE coefficients -- the ray/surfel intersection p = k x l is affine in
//            the pixel offset: p(lx,ly) = A + lx*B + ly*C with A = k0 x l0, B = Tw x l0,
//            C = k0 x Tw, k0/l0 = the reference's k/l at the tile origin -- and scan-convert
//            {alpha >= 1/255} (a conic per pixel row, plus the low-pass disc) onto the tile's 8x8
//            grid of 2x2 blocks (64-bit mask).
//   phase P  (lane = pixel)  per 64 entries each wave transposes the (entry x block) bit matrix
//            with 16 ballots, every quad keeps the words of its block and walks only its set bits
//            over the whole round; records are read from LDS with per-quad addresses.  Skipping is
//            conservative-exact: an entry is skipped for a block only if the reference would have
//            skipped it for all 4 pixels (alpha < 1/255), and list positions (`contributor`
//            numbering) are kept.
// Backward adds: the 22 per-pixel partial derivatives of an entry (coefficient space) are summed
// over the quad with a 2-step DPP reduce-scatter and parked in the LDS slot of the (entry, block)
// pair -- one writer per slot, no atomics --, a second phase (two lanes per entry) adds an entry's
// slots, maps the sums to dL/dT etc. and writes ONE gradient row per (tile, surfel) pair;
// preprocess_bwd gathers a surfel's rows.  The reference issues one atomic per (pixel, surfel,
// component).
#include "common.h"

namespace {

constexpr int FWD_CHUNK = 512;  // list entries staged per round, forward
constexpr int REC4 = 6;        // float4 planes per staged entry (see stage_entry)

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
// Returns the candidate block rectangle packed as gx0 | gx1 << 4 | gy0 << 8 | gy1 << 12 | 1 << 16
// (0 when the tile cannot see the surfel).
template <int CHUNK>
__device__ __forceinline__ uint32_t stage_entry(const float4 *__restrict__ geom, const uint32_t id,
                                                const float4 cb, const bool valid, const float X0,
                                                const float Y0, float4 *rec, uint32_t *ids,
                                                const int e = threadIdx.x, uint2 *mask_out = nullptr) {
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0;
    uint32_t mask_lo = 0, mask_hi = 0, rectpack = 0;
    int gx0 = 0, gx1 = -1, gy0 = 0, gy1 = -1;
    if (valid) {
        // block gx covers pixels X0+2gx, X0+2gx+1: overlap <=> minx <= X0+2gx+1 and maxx >= X0+2gx
        gx0 = (int)ceilf(fminf(fmaxf((cb.x - X0 - 1.f) * 0.5f, 0.f), 8.f));
        gx1 = (int)floorf(fminf(fmaxf((cb.y - X0) * 0.5f, -1.f), 7.f));
        gy0 = (int)ceilf(fminf(fmaxf((cb.z - Y0 - 1.f) * 0.5f, 0.f), 8.f));
        gy1 = (int)floorf(fminf(fmaxf((cb.w - Y0) * 0.5f, -1.f), 7.f));
    }
    if (gx0 <= gx1 && gy0 <= gy1) {  // the 80-byte record is only fetched for surfels the tile can see
        const float4 *g = geom + (size_t)id * 5;
        const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
        const float Tu[3] = {g0.x, g0.y, g0.z}, Tv[3] = {g0.w, g1.x, g1.y}, Tw[3] = {g1.z, g1.w, g2.x};
        const float k0[3] = {X0 * Tw[0] - Tu[0], X0 * Tw[1] - Tu[1], X0 * Tw[2] - Tu[2]};
        const float l0[3] = {Y0 * Tw[0] - Tv[0], Y0 * Tw[1] - Tv[1], Y0 * Tw[2] - Tv[2]};
        float A[3], B[3], C[3];
        cross3(k0, l0, A);
        cross3(Tw, l0, B);
        cross3(k0, Tw, C);
        const float cx = g2.y - X0, cy = g2.z - Y0, opa = g2.w;
        // Scan-convert {alpha >= 1/255} onto the tile, one pixel row at a time.  alpha >= 1/255 <=>
        // rho <= tau = 2 ln(255 o) with rho = min(rho3d, rho2d):
        //   rho3d <= tau  <=>  px^2 + py^2 - tau pz^2 <= 0, a conic; on a row p = u + lx B, so the
        //                      row's pixels lie between the roots of qa lx^2 + 2 qb lx + qc;
        //   rho2d <= tau  <=>  (lx - cx)^2 <= tau/2 - (ly - cy)^2, the low-pass disc.
        // The span kept is the hull of the two intervals, widened by 0.02 px, with tau inflated
        // (the evaluation uses v_rcp / v_exp approximations): conservative, never exact-or-under.
        // A conic that is not an ellipse on this row pencil (qa <= 0) falls back to the cull box.
        const float tau = 2.0f * __logf(255.0f * opa) * 1.001f + 0.01f;
        const float nb = B[0] * B[0] + B[1] * B[1], qa = nb - tau * B[2] * B[2];
        const bool ellipse = qa > 1e-5f * (nb + tau * B[2] * B[2]);
        const float inv_qa = ellipse ? __builtin_amdgcn_rcpf(qa) : 0.f;
        const float bx0 = (float)(2 * gx0), bx1 = (float)(2 * gx1 + 1);
        uint32_t colsum = 0;
        for (int by = gy0; by <= gy1; by++) {
            float lo = 1e9f, hi = -1e9f;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float y = (float)(2 * by + q);
                const float u0 = A[0] + y * C[0], u1 = A[1] + y * C[1], u2 = A[2] + y * C[2];
                const float qb = u0 * B[0] + u1 * B[1] - tau * (u2 * B[2]);
                const float qc = u0 * u0 + u1 * u1 - tau * (u2 * u2);
                const float disc = qb * qb - qa * qc + 4e-6f * (qb * qb);
                if (!ellipse) {
                    lo = bx0; hi = bx1;
                } else if (disc >= 0.f) {
                    const float sq = __builtin_amdgcn_sqrtf(disc);
                    lo = fminf(lo, (-qb - sq) * inv_qa);
                    hi = fmaxf(hi, (-qb + sq) * inv_qa);
                }
                const float h = 0.5f * tau - (y - cy) * (y - cy);
                if (h >= 0.f) {
                    const float sh = __builtin_amdgcn_sqrtf(h);
                    lo = fminf(lo, cx - sh);
                    hi = fmaxf(hi, cx + sh);
                }
            }
            const int ilo = (int)ceilf(fmaxf(lo - 0.02f, bx0)), ihi = (int)floorf(fminf(hi + 0.02f, bx1));
            if (ilo <= ihi) {
                const uint32_t cols = ((2u << (ihi >> 1)) - 1u) & ~((1u << (ilo >> 1)) - 1u);  // 8 bits
                colsum |= cols;
                if (by < 4) mask_lo |= cols << (8 * by);
                else mask_hi |= cols << (8 * (by - 4));
            }
        }
        if (colsum) {
            const int rx0 = __builtin_ctz(colsum), rx1 = 31 - __builtin_clz(colsum);
            const int ry0 = mask_lo ? (__builtin_ctz(mask_lo) >> 3) : 4 + (__builtin_ctz(mask_hi) >> 3);
            const int ry1 = mask_hi ? 4 + ((31 - __builtin_clz(mask_hi)) >> 3) : ((31 - __builtin_clz(mask_lo)) >> 3);
            rectpack = (uint32_t)rx0 | ((uint32_t)rx1 << 4) | ((uint32_t)ry0 << 8) | ((uint32_t)ry1 << 12) | (1u << 16);
            r0 = make_float4(A[0], A[1], A[2], B[0]);
            r1 = make_float4(B[1], B[2], C[0], C[1]);
            r2 = make_float4(C[2], cx, cy, Tw[0]);
            r3 = make_float4(Tw[1], Tw[2], opa, __uint_as_float(mask_lo));
            r4 = make_float4(g3.x, g3.y, g3.z, g4.x);
            r5 = make_float4(g4.y, g4.z, __uint_as_float(mask_hi), 0.f);
        }
    }
    rec[0 * CHUNK + e] = r0; rec[1 * CHUNK + e] = r1; rec[2 * CHUNK + e] = r2;
    rec[3 * CHUNK + e] = r3; rec[4 * CHUNK + e] = r4; rec[5 * CHUNK + e] = r5;
    if (ids) ids[e] = valid ? id : 0u;
    if (mask_out && valid) *mask_out = make_uint2(mask_lo, mask_hi);   // kept for the backward (stage_entry_masked)
    return rectpack;
}

// phase S of the backward: the same planes from the mask the forward stored for this list position -- no cull box, no
// logarithm, no per-row conics; the record is only fetched for entries some block of the tile can see.
template <int CHUNK>
__device__ __forceinline__ void stage_entry_masked(const float4 *__restrict__ geom, const uint32_t id, const uint2 mask,
                                                   const bool valid, const float X0, const float Y0, float4 *rec,
                                                   uint32_t *ids, const int e) {
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0;
    if (valid && (mask.x | mask.y)) {
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
        r3 = make_float4(Tw[1], Tw[2], g2.w, __uint_as_float(mask.x));
        r4 = make_float4(g3.x, g3.y, g3.z, g4.x);
        r5 = make_float4(g4.y, g4.z, __uint_as_float(mask.y), 0.f);
    }
    rec[0 * CHUNK + e] = r0; rec[1 * CHUNK + e] = r1; rec[2 * CHUNK + e] = r2;
    rec[3 * CHUNK + e] = r3; rec[4 * CHUNK + e] = r4; rec[5 * CHUNK + e] = r5;
    ids[e] = valid ? id : 0u;
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

// One workgroup per tile walks the tile's whole list.  (Round 3 built a depth-segment split of the long lists -- transmittance
// prepass, per-segment walks from the true prefix, a combine pass: exact, and 406 us against 254 at LaRa's statistics; DESIGN.md
// section 3.2.  It shipped as an opt-in until round 5 and is gone: the tails are filled by the other views of a multi-view launch.)
// KEEP = the call keeps what a backward needs (candidate masks, segment checkpoints, per-pixel finals, contributor counts,
// per-segment costs, the tile's deepest contributor); a forward-only call (lara2dgs_view.forward_only) writes the images and
// nothing else -- at LaRa's init statistics 62.5 MB written per view become 10.5 (profiles/traffic_r05.json).
template <bool KEEP>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ geom,
                     const uint32_t *__restrict__ tile_order, const float4 *__restrict__ cullbox,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                     const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_cnt,
                     float *__restrict__ ckpt, uint2 *__restrict__ pair_mask, uint32_t *__restrict__ tile_maxc,
                     uint32_t *__restrict__ seg_cost,
                     float *__restrict__ out_color, float *__restrict__ out_allmap, const ViewBatch vb) {
    {   // this workgroup's view (blockIdx.z; a single-view launch has strides 0)
        const long long sst = vb.state_stride, HWb = (long long)v.H * v.W * 4;
        header = l2d_view_ptr(header, sst); ranges = l2d_view_ptr(ranges, sst); point_list = l2d_view_ptr(point_list, sst);
        geom = l2d_view_ptr(geom, sst); tile_order = l2d_view_ptr(tile_order, sst); cullbox = l2d_view_ptr(cullbox, sst);
        final_T = l2d_view_ptr(final_T, sst); n_contrib = l2d_view_ptr(n_contrib, sst); seg_base = l2d_view_ptr(seg_base, sst);
        seg_cnt = l2d_view_ptr(seg_cnt, sst); ckpt = l2d_view_ptr(ckpt, sst); pair_mask = l2d_view_ptr(pair_mask, sst);
        tile_maxc = l2d_view_ptr(tile_maxc, sst); seg_cost = l2d_view_ptr(seg_cost, sst);
        out_color = l2d_view_ptr(out_color, vb.n ? 3 * HWb : 0); out_allmap = l2d_view_ptr(out_allmap, vb.n ? 7 * HWb : 0);
        if (vb.n) v.bg = vb.bg[blockIdx.z];
    }
    constexpr int CHUNK = FWD_CHUNK;
    static_assert(L2D_SEG % FWD_CHUNK == 0, "segment boundaries must fall on round boundaries");
    __shared__ float4 rec[REC4 * CHUNK];
    __shared__ unsigned long long qmask[4][16][CHUNK / 64];  // per wave, per quad: candidate words of the round
    // Two words that live in the staged records' one unused field (plane 5's .w, staged as 0.f = 0u every round): with words of
    // their own the kernel's LDS is 53 256 bytes + padding, and three workgroups only fit a CU up to 53 248 (measured on
    // composite_bwd in round 6: 54 128 bytes ran at two).
    uint32_t &s_cost = *(uint32_t *)&rec[5 * CHUNK + 0].w;   // wave-trips of the current round, summed over the four waves
    uint32_t &s_tmax = *(uint32_t *)&rec[5 * CHUNK + 1].w;   // largest last contributor over the tile's pixels (used after the last round)
    uint32_t &s_ndone = *(uint32_t *)&rec[5 * CHUNK + 2].w;  // waves whose 64 pixels are all finished (__syncthreads_count costs 256 bytes of LDS)
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

    if (header[1]) {  // binning capacity exceeded: make the failure loud in the data (the host repeats the call at the reported size)
        if (inside) {
            const float qnan = __uint_as_float(0x7fc00000u);
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = qnan;
            for (int ch = 0; ch < 7; ch++) out_allmap[ch * HW + pix] = qnan;
            if (KEEP) {
                for (int ch = 0; ch < L2D_CKPT_F; ch++) final_T[pix + ch * HW] = qnan;
                n_contrib[pix] = 0; n_contrib[pix + HW] = 0;
            }
        }
        return;
    }

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    // (tried in round 4: s_setprio for the waves of the long lists, so that a single-view launch -- which lasts as long as its
    // longest list, 5.3 k entries against a mean of 1.5 k -- gets that list done sooner.  No effect, 260.5 vs 261 us: the long
    // list's workgroup is not held back by its co-residents' issue slots but by its own per-round chain of gathers and barriers.)
    const int lo = 0, hi = total;   // this workgroup's entries [lo, hi)
    FwdPixel px;
    px.done = !inside;
    uint32_t *dbg_hdr = const_cast<uint32_t *>(header);
    const long long dbg_t0 = (v.dbg & 32u) ? (long long)__builtin_readcyclecounter() : 0ll;
    int dbg_rounds = 0;

    // What each 512-entry round costs (wave-trips of its walk) is what the backward's work item over the same entries will
    // cost: recorded per (tile, segment) -- full segments at seg_base[tile] + s, everything from the last (partial) segment on in
    // the tile's own slot -- and used by bwd_order_kernel to launch the backward's items dearest first.
    const int tid = threadIdx.x;
    const uint32_t cost_nb = KEEP ? seg_cnt[tile] : 0u;
    uint32_t *const cost_full = KEEP ? seg_cost + seg_base[tile] : nullptr,
             *const cost_last = KEEP ? seg_cost + (v.cap / L2D_SEG + 1u) + tile : nullptr;
    if (KEEP)
        for (uint32_t q = tid; q <= cost_nb; q += 256) *(q < cost_nb ? cost_full + q : cost_last) = 0u;
    int cost_round = -1;
    constexpr int SPT = CHUNK / 256;  // list entries staged per thread and round
    uint32_t id1[SPT], id2[SPT];
    float4 cb1[SPT];
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        const int o = lo + q * 256 + tid;
        id1[q] = o < hi ? point_list[range.x + o] : 0u;
        id2[q] = CHUNK + o < hi ? point_list[range.x + CHUNK + o] : 0u;
    }
#pragma unroll
    for (int q = 0; q < SPT; q++) cb1[q] = lo + q * 256 + tid < hi ? cullbox[id1[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < 3) *(uint32_t *)&rec[5 * CHUNK + threadIdx.x].w = 0u;
    __syncthreads();
    for (int base = lo; base < hi; base += CHUNK) {
        {   // every pixel of the tile finished?  (the word is cleared by the staging below -- 0.f = 0u -- once everybody has read it or
            // reads the cleared word: both say "go on")
            const bool wave_done = __ballot(px.done) == ~0ull;
            if (wave_done && lane == 0) atomicAdd(&s_ndone, 1u);
            __syncthreads();
            if (s_ndone == 4u) break;
        }
        if (KEEP && tid == 0) {     // (between this barrier and the staging barrier no wave is in its walk)
            if (cost_round >= 0) { if ((uint32_t)cost_round < cost_nb) cost_full[cost_round] = s_cost; else *cost_last += s_cost; }
            s_cost = 0u;
        }
        cost_round = (base - lo) / CHUNK;
        dbg_rounds++;
        if (KEEP && base && base % L2D_SEG == 0 && !px.done) {
            // crossing a segment boundary: park the running sums over entries [0, base) so that the
            // backward can start a walk here (pixels that are done never read theirs)
            float *ck = ckpt + ((size_t)seg_base[tile] + (size_t)(base / L2D_SEG - 1)) * (L2D_CKPT_F * 256) + tid;
            ck[0 * 256] = px.T; ck[1 * 256] = px.M1; ck[2 * 256] = px.M2;
            ck[3 * 256] = px.C0; ck[4 * 256] = px.C1; ck[5 * 256] = px.C2;
            ck[6 * 256] = px.Dd; ck[7 * 256] = px.N0; ck[8 * 256] = px.N1; ck[9 * 256] = px.N2;
        }
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const int o = q * 256 + tid;
            const uint32_t id0 = id1[q];
            const float4 cb0 = cb1[q];
            id1[q] = id2[q];
            id2[q] = base + 2 * CHUNK + o < hi ? point_list[range.x + base + 2 * CHUNK + o] : 0u;
            cb1[q] = base + CHUNK + o < hi ? cullbox[id1[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
            stage_entry<CHUNK>(geom, id0, cb0, base + o < hi, X0, Y0, rec, nullptr, o,
                               KEEP && base + o < hi ? pair_mask + range.x + base + o : nullptr);
        }
        __syncthreads();
        if (__ballot(!px.done) == 0ull) continue;  // this quadrant is finished; keep serving barriers
        // Every quad gets the candidate mask of ITS 2x2 block over the whole round (one 64-bit word
        // per 64 entries, transposed out of the staged block masks with 16 ballots each, parked in the
        // wave's own LDS slice) and then walks all of it in one loop: a wave iteration lasts as long
        // as its busiest quad, and over 512 entries the quads' candidate counts are far more even
        // than over 64 (quad-slot efficiency 50 % -> 70 %).
        const int nw = (min(CHUNK, hi - base) + 63) >> 6;
        unsigned long long *qm = &qmask[wave][grp][0];
#pragma unroll 1
        for (int w = 0; w < nw; w++) {
            const float4 *mrec = rec + ((wave >> 1) ? 5 : 3) * CHUNK + w * 64 + lane;
            const uint32_t bm = __float_as_uint((wave >> 1) ? mrec->z : mrec->w);
            unsigned long long m = 0ull;
            if (__ballot((bm & (0x0f0f0f0fu << ((wave & 1) * 4))) != 0u) != 0ull) m = quad_masks(bm, (wave & 1) * 4, grp);
            if ((lane & 3) == 0) qm[w] = m;
        }
        int w = 0;
        unsigned long long mm = qm[0];
        // a quad whose four pixels are finished stops consuming its list
        if (((__ballot(px.done) >> (lane & ~3)) & 0xfull) == 0xfull) { mm = 0ull; w = nw; }
        // (64-bit words: popping 32-bit ones halves the bit twiddling but doubles the LDS word fetches of sparse
        // quads, and measured slower: 258 -> 264 us init, 110 -> 120 us trained)
        auto next = [&](bool &has, int &j) {
            while (mm == 0ull && w + 1 < nw) { w++; mm = qm[w]; }
            has = mm != 0ull;
            j = has ? 64 * w + __builtin_ctzll(mm) : 0;
            mm &= mm - 1ull;  // stays 0 when already empty
        };
        // each quad pops its own next entries; two per trip (independent evaluations), blended
        // strictly in list order; records are fetched one trip ahead.  The loop is unrolled by two with the
        // roles of the two record sets swapped between the halves: rotating them through copies cost 32
        // v_mov per trip (13 % of the loop's issue slots).
        bool hA0, hA1, hB0 = false, hB1 = false;
        int jA0, jA1, jB0 = 0, jB1 = 0;
        next(hA0, jA0);
        next(hA1, jA1);
        EntryRec rA0 = load_entry<CHUNK>(rec, jA0), rA1 = load_entry<CHUNK>(rec, jA1), rB0, rB1;
        auto trip = [&](const EntryRec &c0, const EntryRec &c1, const int j0, const int j1, const bool has0, const bool has1,
                        EntryRec &x0, EntryRec &x1, int &k0, int &k1, bool &n0, bool &n1) {
            next(n0, k0);
            next(n1, k1);
            x0 = load_entry<CHUNK>(rec, k0);
            x1 = load_entry<CHUNK>(rec, k1);
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
            if (((__ballot(px.done) >> (lane & ~3)) & 0xfull) == 0xfull) { mm = 0ull; w = nw; n0 = false; n1 = false; }
        };
        uint32_t ntrips = 0;
        while (true) {
            if (__ballot(hA0) == 0ull) break;
            trip(rA0, rA1, jA0, jA1, hA0, hA1, rB0, rB1, jB0, jB1, hB0, hB1);
            ntrips++;
            if (__ballot(hB0) == 0ull) break;
            trip(rB0, rB1, jB0, jB1, hB0, hB1, rA0, rA1, jA0, jA1, hA0, hA1);
            ntrips++;
        }
        if (KEEP && lane == 0 && ntrips) atomicAdd(&s_cost, ntrips);
    }
    const float T = px.T;
    if (KEEP) {   // the backward's work items of this tile end at the deepest contributor of any of its pixels: one word per tile
        if (threadIdx.x == 0) s_tmax = 0u;
        __syncthreads();
        uint32_t m = px.last_contributor;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if (lane == 0) atomicMax(&s_tmax, m);
        __syncthreads();
        if (threadIdx.x == 0) {
            tile_maxc[tile] = s_tmax;
            if (cost_round >= 0) { if ((uint32_t)cost_round < cost_nb) cost_full[cost_round] = s_cost; else *cost_last += s_cost; }
        }
    }
    if (inside) {
        if (KEEP) {
            final_T[pix] = T;
            final_T[pix + HW] = px.M1;
            final_T[pix + 2 * HW] = px.M2;
            final_T[pix + 3 * HW] = px.C0; final_T[pix + 4 * HW] = px.C1; final_T[pix + 5 * HW] = px.C2;
            final_T[pix + 6 * HW] = px.Dd;
            final_T[pix + 7 * HW] = px.N0; final_T[pix + 8 * HW] = px.N1; final_T[pix + 9 * HW] = px.N2;
            n_contrib[pix] = px.last_contributor;
            n_contrib[pix + HW] = px.median_contributor;
        }
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
// v_mul_legacy_f32: 0 * anything (inf, NaN) = 0; every other product as v_mul_f32
__device__ __forceinline__ float mul_legacy(const float a, const float b) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL>
__device__ __forceinline__ float dpp_full(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// The quad's 2x2 pixels are lanes 0..3 = (x, y) in {(0,0), (1,0), (0,1), (1,1)} of the block.  22 sums over the four lanes
// land 5-6 per lane (slot float 4 i + lane&3 = r[i]; r[5] in lanes 0, 1 only):
//
//   rows 0-2 (one per component c of dp = dL/dp):  lane 0: S_c = sum dp_c    lane 1: sum lx dp_c    lane 2: sum ly dp_c
//                                                  lane 3: sum q_c  (q = three more values: here dz sx, dz sy, dz)
//   The pixel offsets are lx = bx + x, ly = by + y, so sum lx dp = bx S + (dp1 + dp3), sum ly dp = by S + (dp2 + dp3): partial
//   sums the butterfly forms anyway.  After step 1 (xor 1) every lane holds its row's t = dp + dp', q-row sum tq; step 2
//   exchanges  mine = a1 t + a2 dp + a3 tq  against the partner's  send = b1 t + b2 dp + b3 tq  with per-LANE constants
//     lane 0: mine = t            send = by t            -> t0 + t2                     = S
//     lane 2: mine = (by+1) t     send = t               -> by t0 + (by+1) t2           = sum ly dp
//     lane 1: mine = bx t + dp    send = tq              -> bx (t0 + t2) + dp1 + dp3    = sum lx dp
//     lane 3: mine = tq           send = bx t + dp       -> tq0 + tq2                   = sum q
//   i.e. 3 DPP adds + 6 full-rate multiply-adds for four sums, against 3.1 DPP adds + 5.8 (half-rate) selects + the two
//   lx / ly products in the plain reduce-scatter.  (Every input is an exact zero on a lane that does not contribute and finite
//   elsewhere, so the zero coefficients are harmless.)
//   rows 3-5: the plain reduce-scatter of the other ten values h: r[3] = sum h[lane&3], r[4] = sum h[4 + lane&3],
//   r[5] = sum h[8 + lane&1].
struct QuadCoef { float a1, a2, a3, b1, b2, b3; };
__device__ __forceinline__ QuadCoef quad_coef(const int lane, const float lx, const float ly) {
    QuadCoef k;
    const int l = lane & 3;
    k.a1 = l == 0 ? 1.f : (l == 1 ? lx - 1.f : (l == 2 ? ly : 0.f));
    k.a2 = l == 1 ? 1.f : 0.f;
    k.a3 = l == 3 ? 1.f : 0.f;
    k.b1 = l == 0 ? ly : (l == 2 ? 1.f : (l == 3 ? lx - 1.f : 0.f));
    k.b2 = l == 3 ? 1.f : 0.f;
    k.b3 = l == 1 ? 1.f : 0.f;
    return k;
}
__device__ __forceinline__ void quad_reduce_scatter(const float dp[3], const float q[3], const float h[10], const QuadCoef &k,
                                                    float r[6], const int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float d = dp[c];
        asm volatile("" : "+v"(d));      // (left alone, the compiler re-associates dp + dpp(dp) into v_mov 0 + v_mov_dpp + v_fmac)
        const float t = d + dpp_full<0xB1>(d);           // quad_perm [1,0,3,2]
        const float tq = q[c] + dpp_full<0xB1>(q[c]);
        const float mine = k.a1 * t + (k.a2 * d + k.a3 * tq);
        const float send = k.b1 * t + (k.b2 * d + k.b3 * tq);
        r[c] = mine + dpp_full<0x4E>(send);              // quad_perm [2,3,0,1]
    }
    // (A DPP instruction's bank_mask cannot replace the selects: a "bank" is four CONSECUTIVE lanes of a row -- a whole quad --, not a
    // lane position inside the quad.  Tried in round 6 with the masks read as lane positions: 374 -> 366 us and wrong sums.  With the
    // block's four pixels at lane stride 4 the masks would fit, but the full exchanges of rows 0-2 then need two masked DPP adds
    // each where one quad_perm does today: 16 + 12 DPP adds against 17 + 14 selects, 13 issue clocks of 600 per trip.)
    float v1[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const float mine = b0 ? h[2 * i + 1] : h[2 * i];
        const float send = b0 ? h[2 * i] : h[2 * i + 1];
        v1[i] = mine + dpp_full<0xB1>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float mine = b1 ? v1[2 * i + 1] : v1[2 * i];
        const float send = b1 ? v1[2 * i] : v1[2 * i + 1];
        r[3 + i] = mine + dpp_full<0x4E>(send);
    }
    r[5] = v1[4] + dpp_full<0x4E>(v1[4]);
}
// the 16-sum form of the colour-only backward: rows 0-2 as above, r[3] = sum h[lane & 3]
__device__ __forceinline__ void quad_reduce_scatter16(const float dp[3], const float q[3], const float h[4], const QuadCoef &k,
                                                      float r[4], const int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float d = dp[c], qq = q[c];
        asm volatile("" : "+v"(d));
        asm volatile("" : "+v"(qq));     // (q is a product here: left alone it is fused into the sum -- v_mov_dpp + v_fmac, another rounding)
        const float t = d + dpp_full<0xB1>(d);
        const float tq = qq + dpp_full<0xB1>(qq);
        const float mine = k.a1 * t + (k.a2 * d + k.a3 * tq);
        const float send = k.b1 * t + (k.b2 * d + k.b3 * tq);
        r[c] = mine + dpp_full<0x4E>(send);
    }
    float v1[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float mine = b0 ? h[2 * i + 1] : h[2 * i];
        const float send = b0 ? h[2 * i] : h[2 * i + 1];
        v1[i] = mine + dpp_full<0xB1>(send);
    }
    const float mine = b1 ? v1[1] : v1[0];
    const float send = b1 ? v1[0] : v1[1];
    r[3] = mine + dpp_full<0x4E>(send);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Quad-SIMT reverse traversal + two-level, atomic-free reduction.
//
// One workgroup per (tile, 512-entry segment of its list); pixels whose walk began above the segment
// resume from the forward's checkpoint (see the prologue).  A window of 128 entries is staged at a
// time; inside it, rounds of up to 128 entries:
//   phase P  (lane = pixel, each DPP quad walks its own candidate list back to front) runs the
//            per-pixel recurrences of the reference's backward (T, the collapsed accum_rec
//            recurrence, the distortion terms), forms the pixel's 22 coefficient-space partial
//            derivatives, adds the quad's 2x2 pixels with a DPP reduce-scatter and parks the sums in
//            the LDS slot of its (entry, block): plain ds_write, one writer per slot.  An entry's
//            slab has one slot per candidate block of its mask (slot = base + popcount(mask below
//            the block)); a 64-lane scan over the round hands out the bases, and a round takes as
//            many entries as fit the 374-slot pool.
//   phase S2 (two lanes per entry) adds the entry's block slots, maps the sums back through
//            A = k0 x l0, B = Tw x l0, C = k0 x Tw and writes the 80-byte gradient row.  Beside it, on
//            the waves it leaves idle: the next round's slot table and the next window's staging.
// Versus reducing per entry across the wave inside the traversal (one candidate stream per wave, a
// 55-instruction butterfly and a 64-lane gradient evaluation for 11 useful lanes) this needs half
// the wave instructions, and it keeps the backward free of floating-point atomics.
constexpr int SLAB_WIN = 128;     // list entries staged per window (one per thread of the two staging waves)
constexpr int SLAB_CHUNK = 128;   // entries per slab round (two ballot words)
#ifndef L2D_SLAB_POOL
#define L2D_SLAB_POOL 374
#endif
constexpr int SLAB_POOL = L2D_SLAB_POOL;    // (entry, 2x2 block) slots per round (tools/build_variant.sh -DL2D_SLAB_POOL=n for A/B runs); 374: the
                                            // kernel's LDS must stay below 53 248 bytes for three workgroups per CU (384 slots: two, 460 us)
constexpr int SLAB_F = 24;        // floats per slot: 22 quad-reduced sums, padded to 96 bytes (36 KB in all)
// MAPS = false: the call has a gradient on the COLOUR image only (dL_dallmap is NULL = zero: LaRa's fine pass always, its coarse pass
// for the first 1000 iterations -- lightning/loss.py:35-60 puts the distortion and normal terms on the coarse maps alone).  Then
// dL/ddepth of every pair is zero and with it the depth / distortion / median / normal chains of the walk; an (entry, block) slot
// carries 16 sums instead of 22 (64 bytes, written by ONE ds_write_b128 per lane), the pool holds 561 slots instead of 374.  Every
// term that is left is computed as in the full kernel, contraction for contraction: the two give the same bits for seven planes of
// zeros (tests/test_raster_parity_gpu.py, tools/color_only_check.py).
constexpr int SLAB_F_COLOR = 16;
#ifdef L2D_COLOR_OCC4      // A/B: the colour-only form at four workgroups per CU (384 slots = 24 KB, 128 VGPRs, 20 bytes of scratch): measured
                           // 285 -> 270 us per view at init statistics, 77 -> 87 us trained-like (same box, two pairs) -- not shipped
constexpr int SLAB_POOL_COLOR = SLAB_POOL;
#else
constexpr int SLAB_POOL_COLOR = SLAB_POOL * SLAB_F / SLAB_F_COLOR;
#endif

template <bool MAPS>
#ifdef L2D_BWD_WAVES       // waves per SIMD the register allocation aims at (tools/build_variant.sh -DL2D_BWD_WAVES=n for A/B runs)
__global__ void __launch_bounds__(256, L2D_BWD_WAVES)
#elif defined(L2D_COLOR_OCC4)
__global__ void __launch_bounds__(256, MAPS ? 1 : 4)
#else
__global__ void __launch_bounds__(256)
#endif
composite_bwd_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ geom,
                     const uint32_t *__restrict__ tile_order, const float4 *__restrict__ cullbox,
                     const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                     const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_cnt,
                     const uint32_t *__restrict__ bwd_order,
                     const uint2 *__restrict__ bwd_items, const float *__restrict__ ckpt,
                     const uint2 *__restrict__ pair_mask, const uint32_t *__restrict__ tile_maxc,
                     const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dallmap,
                     const uint32_t *__restrict__ pair_pos,
                     float4 *__restrict__ pair_grad, uint8_t *__restrict__ pair_valid, const ViewBatch vb) {
    {
        const long long sst = vb.state_stride, qst = vb.scratch_stride, HWb = (long long)v.H * v.W * 4;
        header = l2d_view_ptr(header, sst); ranges = l2d_view_ptr(ranges, sst); point_list = l2d_view_ptr(point_list, sst);
        geom = l2d_view_ptr(geom, sst); tile_order = l2d_view_ptr(tile_order, sst); cullbox = l2d_view_ptr(cullbox, sst);
        final_T = l2d_view_ptr(final_T, sst); n_contrib = l2d_view_ptr(n_contrib, sst); seg_base = l2d_view_ptr(seg_base, sst);
        seg_cnt = l2d_view_ptr(seg_cnt, sst); bwd_order = l2d_view_ptr(bwd_order, sst); bwd_items = l2d_view_ptr(bwd_items, sst);
        ckpt = l2d_view_ptr(ckpt, sst); pair_mask = l2d_view_ptr(pair_mask, sst); pair_pos = l2d_view_ptr(pair_pos, sst);
        tile_maxc = l2d_view_ptr(tile_maxc, sst);
        pair_grad = l2d_view_ptr(pair_grad, qst); pair_valid = l2d_view_ptr(pair_valid, qst);
        dL_dcolor = l2d_view_ptr(dL_dcolor, vb.n ? 3 * HWb : 0); dL_dallmap = l2d_view_ptr(dL_dallmap, vb.n ? 7 * HWb : 0);
        if (vb.n) v.bg = vb.bg[blockIdx.z];
    }
    constexpr int WIN = SLAB_WIN;
    constexpr int POOL = MAPS ? SLAB_POOL : SLAB_POOL_COLOR, SF = MAPS ? SLAB_F : SLAB_F_COLOR;
    __shared__ float4 rec[REC4 * WIN];
    __shared__ __attribute__((aligned(16))) float pool[(POOL + 1) * SF];    // (+ one slot that stays zero, for phase S2)
    // The round's slot table (first pool slot and live-block mask per entry, number of entries that fit) and the window's surfel ids
    // exist twice: while two lanes per entry add up round r's slots (phase S2), the waves that phase leaves idle stage the next
    // window's records and size round r + 1 -- see the loop below.
    __shared__ uint16_t s_base_[2][SLAB_CHUNK];  // first pool slot of each entry of the round
    __shared__ unsigned long long s_live_[2][SLAB_CHUNK];  // an entry's candidate blocks whose quad still walks it
    __shared__ uint32_t s_ql[64];            // per 2x2 block: last contributor over its four pixels
    __shared__ uint32_t s_id_[2][WIN];
    __shared__ uint2 s_mask[WIN];            // the NEXT window's candidate masks (its first round is sized before it is staged)
    __shared__ int s_nfit_[2];
    __shared__ uint32_t s_total_[2];
    if (header[1]) return;
    const unsigned long long dbg_t0 = (v.dbg & 32u) ? wall_clock64() : 0ull;
    int dbg_rounds = 0;
    uint32_t dbg_windows = 0, dbg_trips = 0, dbg_entries = 0, dbg_slots = 0, dbg_rows = 0;   // LARA2DGS_DEBUG_FLAGS & 512
    unsigned long long dbg_tp = 0ull;            // phase timers (LARA2DGS_DEBUG_FLAGS & 64), thread 0 only
    uint32_t dbg_ph[5] = {0u, 0u, 0u, 0u, 0u};   // prologue, stage, setup, phase P, phase S2 (shader clocks)
#define DBG_PHASE(k) do { if (v.dbg & 64u) { const unsigned long long t__ = __builtin_readcyclecounter(); dbg_ph[k] += (uint32_t)(t__ - dbg_tp); dbg_tp = t__; } } while (0)
    if (v.dbg & 64u) dbg_tp = __builtin_readcyclecounter();
    // work item = (tile, segment): the full segments first, then every tile's last segment
    const uint32_t n_full = header[3];
    int tile, seg;
    if (header[22]) {       // the ordered union of all work items (bwd_order_kernel)
        if (blockIdx.x >= n_full + (uint32_t)v.tiles) return;
        const uint2 it = bwd_items[blockIdx.x];
        tile = (int)it.x; seg = (int)it.y;
    } else if (blockIdx.x < n_full) {
        const uint2 it = bwd_items[blockIdx.x];
        tile = (int)it.x; seg = (int)it.y;
    } else if (blockIdx.x - n_full < (uint32_t)v.tiles) {
        tile = (int)bwd_order[blockIdx.x - n_full];
        seg = (int)seg_cnt[tile];
    } else {
        return;
    }
    // The tile only needs entries [0, max over its pixels of last_contributor): the forward left that maximum in tile_maxc, so a
    // work item beyond it (most of them once surfaces are opaque) leaves here, before any per-pixel load or barrier.
    const uint2 range = ranges[tile];
    const int seg_lo = seg * L2D_SEG;      // the last segment runs to the end of the list
    const int seg_hi = seg == (int)seg_cnt[tile] ? (int)(range.y - range.x) : seg_lo + L2D_SEG;
    const int total = min((int)tile_maxc[tile], seg_hi);
    const int lo = seg_lo;
    if (total <= lo) return;
    const int tx = tile % v.gx, ty = tile / v.gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = lane >> 2;
    const int lxi = (wave & 1) * 8 + (grp & 3) * 2 + (lane & 1);
    const int lyi = (wave >> 1) * 8 + (grp >> 2) * 2 + ((lane >> 1) & 1);
    const int pxi = tx * TILE + lxi, pyi = ty * TILE + lyi;
    const bool inside = pxi < v.W && pyi < v.H;
    const size_t HW = (size_t)v.H * v.W;
    const size_t pix = inside ? (size_t)pyi * v.W + pxi : 0;
    const float lx = (float)lxi, ly = (float)lyi;
    const float X0 = (float)(tx * TILE), Y0 = (float)(ty * TILE);

    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    const uint32_t median_contributor = (MAPS && inside) ? n_contrib[pix + HW] : 0u;
    float dpix[3] = {0.f, 0.f, 0.f}, dnrm[3] = {0.f, 0.f, 0.f};
    float dL_ddepth = 0.f, dL_daccum = 0.f, dL_dmedian = 0.f, dL_dreg = 0.f;
    float final_D = 0.f, final_D2 = 0.f;
    if (inside) {
        for (int ch = 0; ch < 3; ch++) dpix[ch] = dL_dcolor[ch * HW + pix];
        if constexpr (MAPS) {
            dL_ddepth = dL_dallmap[0 * HW + pix];
            dL_daccum = dL_dallmap[1 * HW + pix];
            for (int ch = 0; ch < 3; ch++) dnrm[ch] = dL_dallmap[(2 + ch) * HW + pix];
            dL_dmedian = dL_dallmap[5 * HW + pix];
            dL_dreg = dL_dallmap[6 * HW + pix];
            final_D = final_T[pix + HW];
            final_D2 = final_T[pix + 2 * HW];
        }
    }
    const float final_A = 1.0f - T_final;
    const float bg_dot_dpixel = v.bg[0] * dpix[0] + v.bg[1] * dpix[1] + v.bg[2] * dpix[2];

    float accum_g = 0.f, last_g = 0.f, last_alpha = 0.f;
    float last_dL_dT = 0.f;

    // bits of the 8x8 block mask that precede this lane's 2x2 block (slab slot ranking)
    const int my_blk = (lyi >> 1) * 8 + (lxi >> 1);
    const uint32_t below_lo = my_blk >= 32 ? 0xffffffffu : (1u << my_blk) - 1u;
    const uint32_t below_hi = my_blk >= 32 ? (1u << (my_blk - 32)) - 1u : 0u;
    const QuadCoef qcoef = quad_coef(lane, lx, ly);
    uint32_t quad_last = last_contributor;  // max over the 2x2 block
    quad_last = max(quad_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)quad_last, 0xB1, 0xf, 0xf, false));
    quad_last = max(quad_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)quad_last, 0x4E, 0xf, 0xf, false));
    if ((lane & 3) == 0) s_ql[my_blk] = quad_last;  // read by wave 0 after the first window's barrier
    if (tid < SF) pool[POOL * SF + tid] = 0.f;

    // A pixel whose walk began above this segment resumes from the forward's checkpoint at seg_hi:
    // with F the running sum of f_k * w_k over entries < seg_hi and T_b the transmittance there,
    // the "what lies behind" recurrences equal (F_final - F) / T_b, entered with last_alpha = 0.
    if ((int)last_contributor > seg_hi) {
        const float *ck = ckpt + ((size_t)seg_base[tile] + (size_t)seg) * (L2D_CKPT_F * 256) + tid;
        const float Tb = ck[0], inv_Tb = 1.0f / Tb;
        T = Tb;
        const float s_alpha = (Tb - T_final) * inv_Tb;
        if constexpr (MAPS) {
            const float s_m1 = (final_D - ck[1 * 256]) * inv_Tb, s_m2 = (final_D2 - ck[2 * 256]) * inv_Tb;
            float sg = (Tb - T_final) * dL_daccum + (final_T[pix + 6 * HW] - ck[6 * 256]) * dL_ddepth;
            for (int ch = 0; ch < 3; ch++)
                sg += (final_T[pix + (3 + ch) * HW] - ck[(3 + ch) * 256]) * dpix[ch] +
                      (final_T[pix + (7 + ch) * HW] - ck[(7 + ch) * 256]) * dnrm[ch];
            accum_g = sg * inv_Tb;
            last_dL_dT = (final_D2 * s_alpha + final_A * s_m2 - 2.0f * final_D * s_m1) * dL_dreg;
        } else {
            (void)s_alpha;
            // (each colour product rounded before it is added, as in the full kernel, where it is fused with its -- zero -- normal
            //  product first: the two kernels then resume a segment from the same bits)
            float sg = 0.f;
            for (int ch = 0; ch < 3; ch++) {
                float p = (final_T[pix + (3 + ch) * HW] - ck[(3 + ch) * 256]) * dpix[ch];
                asm volatile("" : "+v"(p));
                sg += p;
            }
            accum_g = sg * inv_Tb;
        }
    }

    // Windows of WIN entries, back to front; window slot e <-> list position whi - 1 - e.  A window is cut into rounds of as many
    // entries as fit the slot pool.  Per round: the walk (all four waves), a barrier, then -- side by side -- phase S2 on the waves
    // that own the round's entries (two lanes per entry: waves 0, 1 for the usual <= 64 entries) and, on waves 2 and 3, everything
    // the NEXT round needs: its slot table (wave 3) and, when it opens a new window, that window's staged records (64 entries per
    // wave); a second barrier.  Up to round 5 the slot table and the staging had phases of their own in front of the walk (14 % of
    // the workgroup's time, one wave busy out of four) and a third barrier per round.  (The pool is not cleared between rounds:
    // every slot a round allots is written by its quad -- an entry's slab has a slot for exactly the blocks whose quads walk it.)
    // Surfel ids and the forward's candidate masks are fetched two windows ahead by the staging waves (entry tid - 128).
    const int se = tid - 128;
    const bool stager = se >= 0;
    uint32_t id1 = 0u, id2 = 0u;            // the next window's ids / masks, and the one behind it
    uint2 mk1 = make_uint2(0u, 0u), mk2 = mk1;
    // sizes round `s0_` of the window that ends at list position `whi_` (`wcnt_` entries) into table `buf`; `side`: the window is
    // not staged yet, its masks come from s_mask.  One wave: lane l sizes entries 2l and 2l+1, one 64-lane scan covers the 128.
    auto size_round = [&](const int whi_, const int wcnt_, const int s0_, const int buf, const bool side) {
        unsigned long long *const live_ = s_live_[buf];
        uint16_t *const base_ = s_base_[buf];
                // A block needs a slot in an entry's slab only while its quad still walks that entry
                // (list position below the quad's last contributor): block b is live from round entry
                // jmin_b = whi - s0 - quad_last_b on.  Lane b drops its bit at jmin_b, an OR-scan over
                // the entries turns that into each entry's live mask.  Where pixels saturate early
                // (opaque surfaces) the deep entries shrink to a few slots and rounds stay full.
                const int jm = max(whi_ - s0_ - (int)s_ql[lane], 0);
                unsigned long long la = ~0ull, lb = ~0ull;
                if (__ballot(jm > 0) != 0ull) {  // (usually every block is live for the whole round: skip)
                    // (volatile: the words are modified by OTHER lanes' atomics between this lane's store and load)
                    volatile unsigned long long *vlive = live_;
                    vlive[2 * lane] = 0ull;
                    vlive[2 * lane + 1] = 0ull;
                    if (jm < SLAB_CHUNK) atomicOr(&live_[jm], 1ull << lane);
                    __builtin_amdgcn_wave_barrier();
                    la = vlive[2 * lane];
                    lb = la | vlive[2 * lane + 1];
                    unsigned long long ex = lb;  // inclusive OR-scan of the pair masks over the lanes
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t ylo = __shfl_up((uint32_t)ex, d, 64), yhi = __shfl_up((uint32_t)(ex >> 32), d, 64);
                        if (lane >= d) ex |= ((unsigned long long)yhi << 32) | ylo;
                    }
                    const uint32_t plo = __shfl_up((uint32_t)ex, 1, 64), phi = __shfl_up((uint32_t)(ex >> 32), 1, 64);
                    const unsigned long long before = lane ? (((unsigned long long)phi << 32) | plo) : 0ull;
                    la |= before; lb |= before;
                }
                uint32_t c[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int slot = s0_ + 2 * lane + q;
                    unsigned long long mk = 0ull;
                    if (slot < wcnt_) {
                        const uint2 mw = side ? s_mask[slot] : make_uint2(__float_as_uint(rec[3 * WIN + slot].w), __float_as_uint(rec[5 * WIN + slot].z));
                        mk = (((unsigned long long)mw.y << 32) | mw.x) & (q ? lb : la);
                    }
                    live_[2 * lane + q] = mk;
                    c[q] = (uint32_t)__builtin_popcountll(mk);
                }
                uint32_t incl = c[0] + c[1];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += y;
                }
                const uint32_t b0 = incl - c[0] - c[1], b1 = incl - c[1];
                const unsigned long long f0 = __ballot(s0_ + 2 * lane < wcnt_ && b1 <= (uint32_t)POOL);
                const unsigned long long f1 = __ballot(s0_ + 2 * lane + 1 < wcnt_ && incl <= (uint32_t)POOL);
                // entries that fit form a prefix; at least one fits (a slab has at most 64 slots)
                const int L = f1 == ~0ull ? 64 : __builtin_ctzll(~f1);
                if (lane == 0) s_nfit_[buf] = L == 64 ? 128 : 2 * L + (int)((f0 >> L) & 1ull);
                base_[2 * lane] = (uint16_t)b0;
                base_[2 * lane + 1] = (uint16_t)b1;
                if (lane == 63) s_total_[buf] = incl;
    };
    {
        const int wcnt = min(WIN, total - lo);
        if (stager) {
            const uint32_t id0 = total - 1 - se >= lo ? point_list[range.x + total - 1 - se] : 0u;
            const uint2 mk0 = total - 1 - se >= lo ? pair_mask[range.x + total - 1 - se] : make_uint2(0u, 0u);
            id1 = total - WIN - 1 - se >= lo ? point_list[range.x + total - WIN - 1 - se] : 0u;
            mk1 = total - WIN - 1 - se >= lo ? pair_mask[range.x + total - WIN - 1 - se] : make_uint2(0u, 0u);
            id2 = total - 2 * WIN - 1 - se >= lo ? point_list[range.x + total - 2 * WIN - 1 - se] : 0u;
            mk2 = total - 2 * WIN - 1 - se >= lo ? pair_mask[range.x + total - 2 * WIN - 1 - se] : make_uint2(0u, 0u);
            stage_entry_masked<WIN>(geom, id0, mk0, se < wcnt, X0, Y0, rec, s_id_[0], se);
        }
        __syncthreads();
        if (wave == 3) size_round(total, wcnt, 0, 0, false);
        __syncthreads();
        DBG_PHASE(0);
    }
    int cur = 0, wb = 0;        // the slot table / the id list in use
    for (int whi = total; whi > lo; whi -= WIN) {
        const int wcnt = min(WIN, whi - lo);
        const bool more_windows = whi - WIN > lo;
        dbg_windows++;
        if (stager) s_mask[se] = mk1;       // (read by wave 3 in this window's last round, barriers away)
        unsigned long long wm0 = 0ull, wm1 = 0ull;      // this quad's candidate bits over the window's 128 slots (set in the first round)

        // Slab rounds over the window.  An entry's slab has one slot per candidate block of its mask whose quad still walks the
        // entry, in mask-bit order: slot = base + rank(block) with rank = popcount(live mask below the block).  A round takes as
        // many entries as fit the pool, at most 128.
        for (int s0 = 0; s0 < wcnt;) {
            dbg_rounds++;
            const unsigned long long *const s_live = s_live_[cur];
            const uint16_t *const s_base = s_base_[cur];
            const uint32_t *const s_id = s_id_[wb];
            const int nfit = s_nfit_[cur];
            if (v.dbg & 512u) { dbg_entries += (uint32_t)nfit; dbg_slots += nfit < SLAB_CHUNK ? (uint32_t)s_base[nfit] : min(s_total_[cur], (uint32_t)POOL); }

            // ---- phase P: every quad walks its own candidates over the whole round, last list position
            //      first (window slots ascend as list positions descend)
            {
                // The (entry x block) bit matrix of the WINDOW is transposed once, in the window's first round (16 ballots per 64
                // entries, ~130 VALU in all: a fifth of a round's fixed cost when every round redid it for its own entries -- and a
                // window takes two rounds on average); a round takes its entries' bits [s0, s0 + nfit) out of the quad's
                // 128-bit window mask.
                if (s0 == 0) {
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const float4 *mrec = rec + ((wave >> 1) ? 5 : 3) * WIN + 64 * k + lane;    // (slots past wcnt are staged as zeros)
                        const uint32_t bm = __float_as_uint((wave >> 1) ? mrec->z : mrec->w);
                        unsigned long long m = 0ull;
                        if (__ballot((bm & (0x0f0f0f0fu << ((wave & 1) * 4))) != 0u) != 0ull) m = quad_masks(bm, (wave & 1) * 4, grp);
                        if (k == 0) wm0 = m; else wm1 = m;
                    }
                }
                unsigned long long m0 = wm0, m1 = wm1;
                {
                    int sh = s0;
                    if (sh >= 64) { m0 = m1; m1 = 0ull; sh -= 64; }
                    if (sh) { m0 = (m0 >> sh) | (m1 << (64 - sh)); m1 >>= sh; }
                    if (nfit < 64) { m0 &= (1ull << nfit) - 1ull; m1 = 0ull; }
                    else if (nfit < 128) m1 &= (1ull << (nfit - 64)) - 1ull;
                }
                // entries at or beyond the last contributor of all four pixels are not this quad's
                const int jmin = whi - s0 - (int)quad_last;
                if (jmin > 0) {
                    m0 = jmin >= 64 ? 0ull : m0 & (~0ull << jmin);
                    m1 = jmin >= 128 ? 0ull : (jmin > 64 ? m1 & (~0ull << (jmin - 64)) : m1);
                }
                // the round's 128 candidate bits as four 32-bit words, popped lowest first (32-bit find-first-set and
                // clear-lowest-bit: a third of the instructions of the 64-bit pair-of-words bookkeeping)
                uint32_t mm = (uint32_t)m0, q1 = (uint32_t)(m0 >> 32), q2 = (uint32_t)m1, q3 = (uint32_t)(m1 >> 32);
                int jb = 0;
                auto trip = [&]() __attribute__((always_inline)) {
                    dbg_trips++;
                    while (mm == 0u && (q1 | q2 | q3) != 0u) { mm = q1; q1 = q2; q2 = q3; q3 = 0u; jb += 32; }
                    const bool has = mm != 0u;
                    const int j = jb + (has ? __builtin_ctz(mm) : 0);  // entry of this round
                    mm &= mm - 1u;
                    const int ws = s0 + j;                               // window slot
                    const uint32_t contributor = (uint32_t)(whi - 1 - ws);  // 0-based list position
                    const EntryRec ent = load_entry<WIN>(rec, ws);
                    Hit h;
                    float Tw[3], opa;
                    const bool active = eval_rec(ent, lx, ly, h, Tw, opa) && has && contributor < last_contributor;
                    // (no shortcut for a trip without a single active lane: it would be taken by a few per cent of the trips and
                    // costs every trip five register copies to keep the state both paths update -- 397 -> 388 us at init)
                    // From here on all lanes run (the quad sums below need uniform control flow): inactive
                    // lanes keep their state and contribute exact zeros.
                    const float4 r4 = rec[4 * WIN + ws], r5 = rec[5 * WIN + ws];
                    const float nrm[3] = {r4.x, r4.y, r4.z}, rgb[3] = {r4.w, r5.x, r5.y};
                    if constexpr (!MAPS) {
                        // Colour gradient only: no depth, distortion, median or normal chain (dL/dz = 0 for every pair); what is
                        // left is the code below, term for term.  16 sums per (entry, block): rows 0-2 as in the full kernel with
                        // (G da, qG2 ddx, qG2 ddy) in the place of the three dL/dz sums, row 3 = the colour products and the weight.
                        (void)nrm;
                        const float alpha = active ? h.alpha : 0.f;
                        const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                        const float T_new = T * inv_1ma;
                        const float w = alpha * T_new;
                        // (the contractions written out the way the compiler fuses the full kernel's longer expressions, so that the
                        //  two kernels round alike: the green product alone, red and blue fused onto it; the background term as a
                        //  rounded product under the fused T_new * dL/dalpha)
                        const float gval = __builtin_fmaf(rgb[2], dpix[2], __builtin_fmaf(rgb[0], dpix[0], rgb[1] * dpix[1]));
                        const float accum_new = mul_legacy(last_alpha, last_g) + (1.f - last_alpha) * accum_g;
                        float dL_dalpha = gval - accum_new;
                        dL_dalpha = __builtin_fmaf(T_new, dL_dalpha, -((T_final * inv_1ma) * bg_dot_dpixel));
                        T = T_new;
                        accum_g = accum_new;
                        last_g = gval;
                        last_alpha = alpha;
                        const float da = active ? dL_dalpha : 0.f;
                        const float sx = h.sx, sy = h.sy;
                        const float rz3 = (active && h.use3d) ? h.rz : 0.f;
                        const float qG = opa * da * h.G;
                        const float dL_dsx = -mul_legacy(qG, sx), dL_dsy = -mul_legacy(qG, sy);
                        const float dpx = dL_dsx * rz3, dpy = dL_dsy * rz3;
                        const float dpv[3] = {dpx, dpy, -(mul_legacy(dpx, sx) + mul_legacy(dpy, sy))};
                        const float qG2 = h.use3d ? 0.f : -FILTER_INV_SQUARE * qG;
                        const float q3[3] = {h.G * da, qG2 * h.ddx, qG2 * h.ddy};
                        const float h4[4] = {w * dpix[0], w * dpix[1], w * dpix[2], w};
                        float r[4];
                        quad_reduce_scatter16(dpv, q3, h4, qcoef, r, lane);
#pragma unroll
                        for (int i = 0; i < 4; i++) asm volatile("" : "+v"(r[i]));
                        const unsigned long long lv = s_live[has ? j : 0];
                        const int rank = __builtin_popcount((uint32_t)lv & below_lo) + __builtin_popcount((uint32_t)(lv >> 32) & below_hi);
                        float4 *ps = (float4 *)(pool + ((int)s_base[has ? j : 0] + rank) * SF) + (lane & 3);
                        if (has) *ps = make_float4(r[0], r[1], r[2], r[3]);     // slot float 4 (lane & 3) + i = r[i]
                    } else {
                    // An inactive lane runs the recurrences with alpha = 0, which leaves its state where the next active
                    // entry would have put it anyway: T / (1 - 0) = T; the collapsed "what lies behind" recurrence
                    // becomes (accum_new, *, 0) and the next step's 0 * last_g + 1 * accum_new reproduces accum_new
                    // bit for bit; last_dL_dT likewise.  One select (alpha) instead of five state selects; the two
                    // products whose other factor may be garbage on such a lane are legacy multiplies (0 * NaN = 0).
                    const float alpha = active ? h.alpha : 0.f, c_d = h.depth;
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    const float T_new = T * inv_1ma;
                    const float w = alpha * T_new;
                    // The colour, depth, alpha and normal channels share one recurrence
                    // ("what lies behind this entry") and enter dL/dalpha only through
                    // their dot product with the pixel's incoming gradient, so the eight
                    // channel recurrences of the published kernel collapse into one.
                    const float gval = rgb[0] * dpix[0] + rgb[1] * dpix[1] + rgb[2] * dpix[2] +
                                       nrm[0] * dnrm[0] + nrm[1] * dnrm[1] + nrm[2] * dnrm[2] +
                                       c_d * dL_ddepth + dL_daccum;
                    const float accum_new = mul_legacy(last_alpha, last_g) + (1.f - last_alpha) * accum_g;
                    float dL_dalpha = gval - accum_new;
                    float dL_dz = 0.0f;
                    const float inv_cd = __builtin_amdgcn_rcpf(c_d);
                    const float m_d = FAR_N / (FAR_N - NEAR_N) * (1.f - NEAR_N * inv_cd);
                    const float dmd_dd = (FAR_N * NEAR_N) / (FAR_N - NEAR_N) * inv_cd * inv_cd;
                    if (contributor + 1 == median_contributor) dL_dz += dL_dmedian;
                    const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2.f * m_d * final_D) * dL_dreg;
                    dL_dalpha += dL_dweight - last_dL_dT;
                    const float dLdT_new = mul_legacy(alpha, dL_dweight) + (1.f - alpha) * last_dL_dT;
                    const float dL_dmd = 2.0f * (T_new * alpha) * (m_d * final_A - final_D) * dL_dreg;
                    dL_dz += dL_dmd * dmd_dd;
                    dL_dalpha *= T_new;
                    dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;
                    dL_dz += w * dL_ddepth;
                    T = T_new;
                    accum_g = accum_new;
                    last_g = gval;
                    last_alpha = alpha;
                    last_dL_dT = dLdT_new;
                    // this pixel's coefficient-space partials (zeros when inactive) ...
                    // An inactive lane's sx, sy, rz may be inf / NaN (pz = 0); every product they enter has a factor
                    // that IS zero there (dz, qG, dpx, dpy, rz3), and v_mul_legacy_f32 (0 * anything = 0) keeps the
                    // zero -- three selects fewer than zeroing sx, sy, rz themselves.  Active lanes: same products.
                    const float ww = w, da = active ? dL_dalpha : 0.f, dz = active ? dL_dz : 0.f;   // (w = 0 when inactive)
                    const float sx = h.sx, sy = h.sy;
                    const float rz3 = (active && h.use3d) ? h.rz : 0.f;     // the 3-D branch's 1/pz, else 0
                    // depth = s . Tw.xy + Tw.z (the published backward uses this form in both branches)
                    const float qv[3] = {mul_legacy(dz, sx), mul_legacy(dz, sy), dz};
                    const float qG = opa * da * h.G;  // dL/dG * G  (G is finite on every lane)
                    const float dL_dsx = dz * Tw[0] - mul_legacy(qG, sx);
                    const float dL_dsy = dz * Tw[1] - mul_legacy(qG, sy);
                    const float dpx = dL_dsx * rz3, dpy = dL_dsy * rz3;
                    const float dpv[3] = {dpx, dpy, -(mul_legacy(dpx, sx) + mul_legacy(dpy, sy))};
                    const float qG2 = h.use3d ? 0.f : -FILTER_INV_SQUARE * qG;   // the low-pass branch's share
                    // hv: centre (2), normal (3), opacity, colour (3), sum of weights (> 0 marks a block that contributed)
                    const float hv[10] = {qG2 * h.ddx, qG2 * h.ddy, ww * dnrm[0], ww * dnrm[1], ww * dnrm[2], h.G * da,
                                          ww * dpix[0], ww * dpix[1], ww * dpix[2], ww};
                    // ... summed over the quad's 2x2 pixels with a DPP reduce-scatter (each lane ends up with 5-6 of the 22
                    // sums; the pixel-offset moments of dL/dp come out of the butterfly's own partial sums) and parked in the
                    // (entry, block) slot: plain stores, one writer per slot
                    float r[6];
                    quad_reduce_scatter(dpv, qv, hv, qcoef, r, lane);
                    // (pin the sums here: left alone, the compiler sinks the second step's adds into the `if (has)` below,
                    // away from their DPP operand fetches, and pays a v_mov 0 + v_mov_dpp + v_add per value instead of one
                    // v_add_f32_dpp)
#pragma unroll
                    for (int i = 0; i < 6; i++) asm volatile("" : "+v"(r[i]));
                    const unsigned long long lv = s_live[has ? j : 0];
                    const int rank = __builtin_popcount((uint32_t)lv & below_lo) + __builtin_popcount((uint32_t)(lv >> 32) & below_hi);
                    float *ps = pool + ((int)s_base[has ? j : 0] + rank) * SLAB_F + (lane & 3);
                    if (has) {
                        ps[0] = r[0]; ps[4] = r[1]; ps[8] = r[2]; ps[12] = r[3]; ps[16] = r[4];
                        if (!(lane & 2)) ps[20] = r[5];
                    }
                    }   // MAPS
                };
                // (two trips per loop iteration: the five state registers alternate instead of being copied at the loop's top)
                while (__ballot((mm | q1 | q2 | q3) != 0u) != 0ull) {
                    trip();
#ifndef L2D_WALK_NO_UNROLL
                    if (__ballot((mm | q1 | q2 | q3) != 0u) == 0ull) break;
                    trip();
#endif
                }
            }
            __syncthreads();        // the round's slots are written
            DBG_PHASE(3);

            // ---- phase S2: two lanes per entry add up the entry's block slots (no geometry any more); a wave whose 32 entries lie
            //      beyond the round (a round holds 61 entries on average at LaRa's init statistics) skips it
            if ((wave << 5) < nfit && !(v.dbg & 2u)) {
                const int e = tid >> 1, part = tid & 1;
                const int ws = min(s0 + e, WIN - 1);
                const bool has = e < nfit;
                const int cnt = has ? __builtin_popcountll(s_live[e]) : 0;
                // the entry's T rows are needed at the very end: fetch them now, behind the slot loop
                float4 gq0 = make_float4(0.f, 0.f, 0.f, 0.f), gq1 = gq0, gq2 = gq0;
                if (cnt && part == 0) {
                    const float4 *gm = geom + (size_t)s_id[ws] * 5;
                    gq0 = gm[0]; gq1 = gm[1]; gq2 = gm[2];
                }
                // the lane's first slot is LOADED (a lane without one reads the pool's extra, always-zero slot): no 24 zero moves and
                // no adds for the first slot
                constexpr int NG = MAPS ? 22 : 16;
                float g[NG];
                const float4 *ps = (const float4 *)(pool + (int)s_base[e] * SF);
                {
                    const float4 *p0 = part < cnt ? ps + part * (SF / 4) : (const float4 *)(pool + POOL * SF);
#pragma unroll
                    for (int q = 0; q < NG / 4; q++) {
                        const float4 t = p0[q];
                        g[4 * q] = t.x; g[4 * q + 1] = t.y; g[4 * q + 2] = t.z; g[4 * q + 3] = t.w;
                    }
                    if constexpr (MAPS) {
                        const float2 t2 = *(const float2 *)(p0 + 5);
                        g[20] = t2.x; g[21] = t2.y;
                    }
                }
                for (int i = part + 2; i < cnt; i += 2) {
#pragma unroll
                    for (int q = 0; q < NG / 4; q++) {
                        const float4 t = ps[i * (SF / 4) + q];
                        g[4 * q] += t.x; g[4 * q + 1] += t.y; g[4 * q + 2] += t.z; g[4 * q + 3] += t.w;
                    }
                    if constexpr (MAPS) {
                        const float2 t2 = *(const float2 *)(ps + i * (SF / 4) + 5);
                        g[20] += t2.x; g[21] += t2.y;
                    }
                }
#pragma unroll
                for (int k = 0; k < NG; k++) g[k] += dpp_full<0xB1>(g[k]);  // the entry's two lanes
                // (colour-only slots: float 4 l + i = lane l's r[i] -- rows 0-2 transposed, (G da, qG2 ddx, qG2 ddy) where the full
                //  kernel has the dL/dz sums, the colour products and the weight in every lane's last float)
                const bool touched = (MAPS ? g[21] : g[15]) > 0.f;
                if (part == 0 && touched) {
                    const uint32_t p = range.x + (uint32_t)(whi - 1 - ws);
                    const float4 g0 = gq0, g1 = gq1, g2 = gq2;
                    const float Tu[3] = {g0.x, g0.y, g0.z}, Tv[3] = {g0.w, g1.x, g1.y}, Tw[3] = {g1.z, g1.w, g2.x};
                    const float k0[3] = {X0 * Tw[0] - Tu[0], X0 * Tw[1] - Tu[1], X0 * Tw[2] - Tu[2]};
                    const float l0[3] = {Y0 * Tw[0] - Tv[0], Y0 * Tw[1] - Tv[1], Y0 * Tw[2] - Tv[2]};
                    // slot rows 0-2 = per component of dL/dp: (sum, sum lx, sum ly, q) -- see quad_reduce_scatter
                    const float a[3] = {g[0], MAPS ? g[4] : g[1], MAPS ? g[8] : g[2]};
                    const float b[3] = {MAPS ? g[1] : g[4], g[5], MAPS ? g[9] : g[6]};
                    const float cc[3] = {MAPS ? g[2] : g[8], MAPS ? g[6] : g[9], g[10]};
                    const float qd[3] = {MAPS ? g[3] : 0.f, MAPS ? g[7] : 0.f, MAPS ? g[11] : 0.f};       // sums of dz sx, dz sy, dz
                    // A = k0 x l0, B = Tw x l0, C = k0 x Tw ; for y = u x v: dL/du = v x dL/dy, dL/dv = dL/dy x u
                    float t1[3], t2[3], dk0[3], dl0[3], dTw[3];
                    cross3(l0, a, t1); cross3(Tw, cc, t2);
                    for (int i = 0; i < 3; i++) dk0[i] = t1[i] + t2[i];
                    cross3(a, k0, t1); cross3(b, Tw, t2);
                    for (int i = 0; i < 3; i++) dl0[i] = t1[i] + t2[i];
                    cross3(l0, b, t1); cross3(cc, k0, t2);
                    for (int i = 0; i < 3; i++) {
                        dTw[i] = t1[i] + t2[i] + X0 * dk0[i] + Y0 * dl0[i];
                        if constexpr (MAPS) dTw[i] += qd[i];
                    }
                    // one 80-byte gradient row per touched (tile, surfel) pair, written exactly once, at the pair's
                    // SURFEL-MAJOR index (pair_pos[p], recorded by the sort): a surfel's rows are contiguous and
                    // preprocess_bwd streams them; a byte per pair marks the rows that exist
                    const uint32_t q = pair_pos[p];
                    float4 *row = pair_grad + (size_t)q * (GRAD_F / 4);
                    row[0] = make_float4(-dk0[0], -dk0[1], -dk0[2], -dl0[0]);
                    row[1] = make_float4(-dl0[1], -dl0[2], dTw[0], dTw[1]);
                    if constexpr (MAPS) {
                        row[2] = make_float4(dTw[2], g[12], g[13], g[14]);
                        row[3] = make_float4(g[15], g[16], g[17], g[18]);
                        row[4] = make_float4(g[19], g[20], 0.f, 0.f);
                    } else {
                        row[2] = make_float4(dTw[2], g[13], g[14], 0.f);
                        row[3] = make_float4(0.f, 0.f, g[12], g[3]);
                        row[4] = make_float4(g[7], g[11], 0.f, 0.f);
                    }
                    pair_valid[q] = 1;
                    dbg_rows++;
                }
            }
            // ---- beside phase S2, on waves 2 and 3: the next window's records (when this was the window's last round), the next
            //      round's slot table
            const bool last_round = s0 + nfit >= wcnt;
            if (stager && last_round && more_windows) {
                stage_entry_masked<WIN>(geom, id1, mk1, se < min(WIN, whi - WIN - lo), X0, Y0, rec, s_id_[wb ^ 1], se);
                id1 = id2;
                mk1 = mk2;
                id2 = whi - 3 * WIN - 1 - se >= lo ? point_list[range.x + whi - 3 * WIN - 1 - se] : 0u;
                mk2 = whi - 3 * WIN - 1 - se >= lo ? pair_mask[range.x + whi - 3 * WIN - 1 - se] : make_uint2(0u, 0u);
            }
            if (wave == 3) {
                if (!last_round) size_round(whi, wcnt, s0 + nfit, cur ^ 1, false);
                else if (more_windows) size_round(whi - WIN, min(WIN, whi - WIN - lo), 0, cur ^ 1, true);
            }
            __syncthreads();        // round r's slots are summed, round r + 1 is sized (and staged)
            DBG_PHASE(4);
            s0 += nfit;
            cur ^= 1;
        }
        wb ^= 1;
    }
    if ((v.dbg & 64u) && tid == 0) {
        DBG_PHASE(4);
        uint32_t *h = const_cast<uint32_t *>(header);
        for (int k = 0; k < 5; k++) atomicAdd(&h[16 + k], dbg_ph[k] >> 6);  // units of 64 clocks
    }
    if (v.dbg & 512u) {   // walk statistics (tools/bwd_probe.py): work items, windows, rounds, wave-trips, entries and slots of the rounds, rows
        uint32_t *h = const_cast<uint32_t *>(header);
        if (tid == 0) {
            atomicAdd(&h[24], 1u); atomicAdd(&h[25], dbg_windows); atomicAdd(&h[26], (uint32_t)dbg_rounds);
            atomicAdd(&h[28], dbg_entries); atomicAdd(&h[29], dbg_slots);
        }
        if (lane == 0) atomicAdd(&h[27], dbg_trips);
        if (dbg_rows) atomicAdd(&h[30], dbg_rows);
    }
    if ((v.dbg & 32u) && tid == 0) {  // work-group residency in 100 MHz ticks: max, sum, first start, last end
        uint32_t *h = const_cast<uint32_t *>(header);
        const unsigned long long t1 = wall_clock64();
        atomicMax(&h[8], (uint32_t)(t1 - dbg_t0));
        atomicMax((unsigned long long *)&h[12], ((unsigned long long)(uint32_t)(t1 - dbg_t0) << 32) |
                                                    ((unsigned long long)dbg_rounds << 16) | (unsigned long long)blockIdx.x);
        atomicAdd(&h[9], (uint32_t)(t1 - dbg_t0));
        atomicMax(&h[10], ~(uint32_t)dbg_t0);
        atomicMax(&h[11], (uint32_t)t1);
    }
}

// reduce-scatter self-test: in [64][16] (lane major: dp 3, q 3, h 10) -> out [16 quads][24] = the quad's slot; the lanes'
// pixel offsets are those of wave 0 of a workgroup
__global__ void __launch_bounds__(64)
selftest_butterfly_kernel(const float *__restrict__ in, float *__restrict__ out) {
    const int lane = threadIdx.x, q4 = lane & 3, grp = lane >> 2;
    const float lx = (float)((grp & 3) * 2 + (lane & 1)), ly = (float)((grp >> 2) * 2 + ((lane >> 1) & 1));
    float dp[3], q[3], h[10], r[6];
    for (int k = 0; k < 3; k++) { dp[k] = in[lane * 16 + k]; q[k] = in[lane * 16 + 3 + k]; }
    for (int k = 0; k < 10; k++) h[k] = in[lane * 16 + 6 + k];
    quad_reduce_scatter(dp, q, h, quad_coef(lane, lx, ly), r, lane);
    float *dst = out + (lane >> 2) * 24 + q4;
    for (int i = 0; i < 5; i++) dst[4 * i] = r[i];
    if (q4 < 2) dst[20] = r[5];
}

// ------------------------------------------------------------------------------------------------
// backward launch order: dearest work item first
// ------------------------------------------------------------------------------------------------
// tile_scan lists the backward's work items before anything is known about them: the full 512-entry segments in TILE-ID order,
// the tiles' last segments by length.  A single-view launch then lasts as long as the dear items that happen to start late
// (tools/bwd_probe.py, trained-like: 43 ms of workgroup time over 768 slots = 56 us, span 144 us, slowest item 109 us).  The
// forward has walked the same entries in the same 512-entry rounds: its wave-trips per round (seg_cost) are re-used here as the
// items' cost: ALL work items -- the full segments and every tile's last one -- go into one list, ordered by a 64-bucket
// counting sort, dearest first (4 wave-trips per bucket, everything from 252 on in the first).  One workgroup; every item is read
// into registers before the first is written back, so bwd_items is permuted (and extended by the last segments) in place: one
// global round trip, two barriers.  header[22] marks the list as the ordered union (composite_bwd then maps blockIdx.x through it
// alone; a second backward over the same state skips the pass).
// The same launch zero-fills the backward's validity bitmap (workgroups 1..): it replaces the memset launch that stood in front of
// composite_bwd, so the ordering costs no launch.
constexpr int BO_PER_THREAD = 17;     // items per thread: more than 17 408 work items stay in tile_scan's order
__global__ void __launch_bounds__(1024)
bwd_order_kernel(ViewDev v, uint32_t *__restrict__ header, uint2 *__restrict__ bwd_items, const uint32_t *__restrict__ seg_cnt,
                 const uint32_t *__restrict__ seg_cost, const long long sst, char *__restrict__ zero_base, const long long zero_bytes,
                 const long long qst) {
    if (blockIdx.x >= 1) {      // workgroups 1.. : the zero fill of the backward's validity bitmap (it used to be a launch of its own)
        uint4 *z = (uint4 *)l2d_view_ptr(zero_base, qst);
        const long long n16 = zero_bytes / 16;
        for (long long i = (long long)(blockIdx.x - 1) * 1024 + threadIdx.x; i < n16; i += (long long)(gridDim.x - 1) * 1024)
            z[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    header = l2d_view_ptr(header, sst); bwd_items = l2d_view_ptr(bwd_items, sst); seg_cnt = l2d_view_ptr(seg_cnt, sst);
    seg_cost = l2d_view_ptr(seg_cost, sst);
    __shared__ uint32_t bcnt[64];
    if (header[1] || (v.dbg & 256u)) return;      // (overflow; 256: A/B runs that keep tile_scan's order)
    const int tid = threadIdx.x, lane = tid & 63;
    if (header[22]) return;                        // already ordered (the costs are indexed by the ORIGINAL positions)
    // the work items: the full segments as tile_scan listed them, then every tile's last segment (tile, seg_cnt[tile])
    const uint32_t n_full = header[3], n = n_full + (uint32_t)v.tiles;
    if (n > (uint32_t)(BO_PER_THREAD * 1024)) return;
    const uint32_t *cost_last = seg_cost + (v.cap / L2D_SEG + 1u);
    if (tid < 64) bcnt[tid] = 0u;
    __syncthreads();
    uint2 item[BO_PER_THREAD];
    uint32_t bkt[BO_PER_THREAD];
#pragma unroll
    for (int k = 0; k < BO_PER_THREAD; k++) {
        const uint32_t i = (uint32_t)tid + 1024u * k;
        bkt[k] = 64u;
        if (i < n) {
            uint32_t c;
            if (i < n_full) { item[k] = bwd_items[i]; c = seg_cost[i]; }
            else { const uint32_t t = i - n_full; item[k] = make_uint2(t, seg_cnt[t]); c = cost_last[t]; }
            bkt[k] = 63u - min(63u, c >> 2);
            atomicAdd(&bcnt[bkt[k]], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const uint32_t c = bcnt[tid];
        uint32_t x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        bcnt[tid] = x - c;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BO_PER_THREAD; k++)
        if (bkt[k] < 64u) bwd_items[atomicAdd(&bcnt[bkt[k]], 1u)] = item[k];
    if (tid == 0) header[22] = 1u;
}

}  // namespace

int launch_composite_fwd(const ViewDev &v, StateView st, ScratchView sc, float *out_color, float *out_allmap,
                         hipStream_t s, const ViewBatch *vbp) {
    (void)sc;
    ViewBatch vb{};
    if (vbp) vb = *vbp;
    const unsigned nz = vbp ? (unsigned)vb.n : 1u;    // blockIdx.z = view (st / out_* are view 0's)
    {
        L2D_PROF(v.fwd_only ? "composite_fwd_only" : "composite_fwd", s);
        auto kern = v.fwd_only ? composite_fwd_kernel<false> : composite_fwd_kernel<true>;
        hipLaunchKernelGGL(kern, dim3(v.tiles, 1, nz), dim3(256), 0, s, v, st.header, st.ranges, st.point_list,
                           (const float4 *)st.geom, st.tile_order, (const float4 *)st.cullbox, st.final_T, st.n_contrib, st.seg_base,
                           st.seg_cnt, st.ckpt, st.pair_mask, st.tile_maxc, st.seg_cost, out_color, out_allmap, vb);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_bwd_order(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s, const ViewBatch *vbp, void *zero_base, int64_t zero_bytes) {
    (void)sc;
    L2D_PROF("bwd_order", s);
    const int64_t zb = (zero_bytes + 15) / 16 * 16;      // (the region is 256-byte aligned and padded: rounding up stays inside it)
    const unsigned zw = zb > 0 ? (unsigned)((zb / 16 + 8191) / 8192 < 128 ? (zb / 16 + 8191) / 8192 : 128) : 0u;
    hipLaunchKernelGGL(bwd_order_kernel, dim3(1 + zw, 1, vbp ? (unsigned)vbp->n : 1u), dim3(1024), 0, s, v, st.header, st.bwd_items, st.seg_cnt,
                       st.seg_cost, vbp ? vbp->state_stride : 0ll, (char *)zero_base, (long long)zb, vbp ? vbp->scratch_stride : 0ll);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_composite_bwd(const ViewDev &v, StateView st, ScratchView sc, const float *dL_dcolor,
                         const float *dL_dallmap, hipStream_t s, const ViewBatch *vbp) {
    ViewBatch vb{};
    if (vbp) vb = *vbp;
    {
        L2D_PROF(dL_dallmap ? "composite_bwd" : "composite_bwd_color", s);
        // one workgroup per (tile, segment); the count lives on the device (header[3]), so launch
        // the upper bound -- surplus workgroups exit on their first instruction
        const unsigned grid = (unsigned)v.tiles + v.cap / L2D_SEG;
        // dL_dallmap == NULL: the gradient on the seven maps is zero -> the colour-only kernel
        auto kern = dL_dallmap ? composite_bwd_kernel<true> : composite_bwd_kernel<false>;
        hipLaunchKernelGGL(kern, dim3(grid, 1, vbp ? (unsigned)vb.n : 1u), dim3(256), 0, s, v, st.header, st.ranges,
                           st.point_list, (const float4 *)st.geom, st.tile_order,
                           (const float4 *)st.cullbox, st.final_T, st.n_contrib, st.seg_base, st.seg_cnt, st.bwd_order,
                           st.bwd_items, st.ckpt, st.pair_mask, st.tile_maxc, dL_dcolor, dL_dallmap, st.pair_pos, sc.pair_grad,
                           (uint8_t *)sc.pair_valid, vb);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_selftest_butterfly(const float *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL(selftest_butterfly_kernel, dim3(1), dim3(64), 0, s, in, out);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
