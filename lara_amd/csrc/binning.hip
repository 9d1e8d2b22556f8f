// binning.hip -- duplicate-and-tile-bin + per-tile key sort for gfx950.
//
// The reference pipeline (published 2DGS/3DGS rasteriser) is: inclusive scan of tiles_touched ->
// D2H copy of the total (host sync) -> duplicateWithKeys (64-bit key = tile << 32 | depth bits)
// -> device-wide 64-bit LSD radix sort over all D pairs (6 passes, ~24 B moved per pair per pass)
// -> identifyTileRanges.  Its OUTPUT CONTRACT is: per tile, the list of surfel ids ordered by
// (depth bits ascending, then surfel id ascending -- the radix sort is stable and keys are emitted
// in id order), plus [start,end) per tile with tiles laid out in increasing tile id.
//
// MI355X-native formulation with the same output, bit for bit:
//   1. tile population histogram            (atomics, done inside preprocess_fwd)
//   2. exclusive scan over <= a few thousand tiles in ONE workgroup -> ranges directly
//      (no identifyTileRanges pass, no device-wide scan over P, no host sync: the total stays on
//      the device and is checked against the caller's capacity there)
//   3. scatter (depth bits, id) into the tile's segment (order inside a segment is arbitrary)
//   4. per-tile bitonic sort of the 64-bit (depth bits << 32 | id) words in LDS: a tile's list
//      (~1.5k entries at LaRa's sizes) fits the 160 KB LDS many times over, so the sort never
//      touches HBM beyond one read and one write of the segment.
// Traffic: 8 B write + 8 B read + 4 B write per pair instead of ~144 B per pair for 6 radix passes.
#include "common.h"

namespace {

// ---- sorting network shared by the tile-order sort and the per-tile key sort ----------------
// Bitonic network in its "all comparators ascending" form (first step of each merge mirrors the
// block), so virtual +inf padding above n never moves: comparators touching an index >= n are
// skipped.  Works for any n, in LDS or (oversized tiles) directly in the global segment -- one
// workgroup owns a segment, and __syncthreads() orders its own global accesses.
template <typename Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr a, const uint32_t n, const uint32_t m /*pow2 >= n*/) {
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const bool flip = (j == (k >> 1));
            for (uint32_t t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                // t-th comparator: lower index i has bit j clear
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t p = flip ? (i ^ (k - 1)) : (i ^ j);
                if (p < n) {  // i < p always
                    const uint64_t x = a[i], y = a[p];
                    if (y < x) { a[i] = y; a[p] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- 2. exclusive scan of tile counts -> ranges ------------------------------------------------
__global__ void __launch_bounds__(1024)
tile_scan_kernel(ViewDev v, const uint32_t *__restrict__ tile_count, uint2 *__restrict__ ranges,
                 uint32_t *__restrict__ header, uint32_t *__restrict__ tile_order) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s, s_max;
    __shared__ uint32_t bcnt[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    uint32_t max_len = 0;
    for (int base = 0; base < v.tiles; base += 1024) {
        const int i = base + tid;
        const uint32_t c = i < v.tiles ? tile_count[i] : 0u;
        max_len = c > max_len ? c : max_len;
        // inclusive scan inside the wave
        uint32_t x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_sums[wid] = x;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; w++) wave_off += wave_sums[w];
        const uint32_t carry = carry_s;
        const uint32_t incl = carry + wave_off + x;
        if (i < v.tiles) ranges[i] = c ? make_uint2(incl - c, incl) : make_uint2(0u, 0u);
        __syncthreads();
        if (tid == 1023) carry_s = incl;
        __syncthreads();
    }
    // block max of the longest tile list (diagnostic) and the total
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t y = __shfl_xor(max_len, d, 64);
        max_len = y > max_len ? y : max_len;
    }
    __syncthreads();
    if (lane == 0) wave_sums[wid] = max_len;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < 16; w++) m = wave_sums[w] > m ? wave_sums[w] : m;
        const uint32_t total = carry_s;
        header[0] = total;
        header[1] = total > v.cap ? 1u : 0u;
        header[2] = m;
        s_max = m;
    }
    // Launch order of the composite kernels: longest list first (LPT), so that the heavy centre
    // tiles spread over all CUs instead of piling onto the few CUs their ids map to.  A 64-bucket
    // counting sort on the list length is enough (order inside a bucket is irrelevant).
    if (tid < 64) bcnt[tid] = 0;
    __syncthreads();
    const uint32_t denom = s_max + 1u;
    for (int i = tid; i < v.tiles; i += 1024) {
        const uint32_t b = 63u - (uint32_t)(((uint64_t)tile_count[i] * 64u) / denom);
        atomicAdd(&bcnt[b], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the 64 bucket sizes inside wave 0
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
    for (int i = tid; i < v.tiles; i += 1024) {
        const uint32_t b = 63u - (uint32_t)(((uint64_t)tile_count[i] * 64u) / denom);
        tile_order[atomicAdd(&bcnt[b], 1u)] = (uint32_t)i;
    }
}

// ---- 3. scatter (depth bits, id) into the tile segments ----------------------------------------
// Two-level slot reservation: the workgroup counts its pairs per tile in LDS, reserves one
// contiguous run per touched tile with ONE returning device-scope atomic, then hands out slots
// inside the run with LDS atomics.  (Order inside a tile segment is arbitrary by design.)
__global__ void __launch_bounds__(256)
scatter_kernel(ViewDev v, const ushort4 *__restrict__ rect,
               const float4 *__restrict__ geom, const uint2 *__restrict__ ranges,
               uint32_t *__restrict__ tile_fill, uint64_t *__restrict__ keys, const int use_lds) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    ushort4 r = make_ushort4(0, 0, 0, 0);
    uint64_t word = 0;
    if (idx < v.P) {
        r = rect[idx];
        if (r.z > r.x && r.w > r.y) {
            const float depth = geom[(size_t)idx * 5 + 3].w;
            word = ((uint64_t)__float_as_uint(depth) << 32) | (uint32_t)idx;
        }
    }
    if (use_lds) {
        for (int t = threadIdx.x; t < v.tiles; t += blockDim.x) hist[t] = 0;
        __syncthreads();
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) atomicAdd(&hist[y * v.gx + x], 1u);
        __syncthreads();
        for (int t = threadIdx.x; t < v.tiles; t += blockDim.x) {
            const uint32_t c = hist[t];
            if (c) hist[t] = ranges[t].x + __hip_atomic_fetch_add(&tile_fill[t], c, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) {
                const uint32_t slot = atomicAdd(&hist[y * v.gx + x], 1u);
                if (slot < v.cap) keys[slot] = word;
            }
    } else {
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) {
                const int t = y * v.gx + x;
                const uint32_t slot = ranges[t].x + __hip_atomic_fetch_add(&tile_fill[t], 1u, __ATOMIC_RELAXED,
                                                                            __HIP_MEMORY_SCOPE_AGENT);
                if (slot < v.cap) keys[slot] = word;
            }
    }
}

// ---- 4. per-tile sort (network defined above) ------------------------------------------------
__device__ __forceinline__ uint32_t next_pow2(uint32_t n) {
    return n <= 1 ? 1u : 1u << (32 - __clz(n - 1));
}

template <int LDS_ENTRIES, bool SMALL>
__global__ void __launch_bounds__(256)
tile_sort_kernel(ViewDev v, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ header,
                 uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list) {
    __shared__ uint64_t buf[LDS_ENTRIES];
    const uint2 rg = ranges[blockIdx.x];
    uint32_t n = rg.y - rg.x;
    if (header[1]) return;  // capacity overflow: lists are incomplete, outputs get poisoned instead
    if (SMALL ? (n > (uint32_t)LDS_ENTRIES) : (n <= 2048u)) return;  // the other kernel's tile
    if (n == 0) return;
    const uint32_t m = next_pow2(n);
    uint64_t *seg = keys + rg.x;
    if (n <= (uint32_t)LDS_ENTRIES) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) buf[i] = seg[i];
        __syncthreads();
        bitonic_sort(buf, n, m);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) point_list[rg.x + i] = (uint32_t)buf[i];
    } else {
        __syncthreads();
        bitonic_sort(seg, n, m);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) point_list[rg.x + i] = (uint32_t)seg[i];
    }
}

}  // namespace

int launch_binning(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s) {
    {
        L2D_PROF("tile_scan", s);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, v, sc.tile_count, st.ranges, st.header,
                           st.tile_order);
    }
    L2D_CHECK_LAUNCH();
    if (v.P == 0) return LARA2DGS_OK;
    {
        L2D_PROF("scatter", s);
        const int use_lds = v.tiles <= L2D_LDS_HIST_TILES;
        hipLaunchKernelGGL(scatter_kernel, dim3((v.P + 255) / 256), dim3(256),
                           use_lds ? (size_t)v.tiles * 4 : 0, s, v, sc.rect, (const float4 *)st.geom,
                           st.ranges, sc.tile_fill, sc.keys, use_lds);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("tile_sort_small", s);
        hipLaunchKernelGGL((tile_sort_kernel<2048, true>), dim3(v.tiles), dim3(256), 0, s, v, st.ranges,
                           st.header, sc.keys, st.point_list);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("tile_sort_large", s);
        hipLaunchKernelGGL((tile_sort_kernel<8192, false>), dim3(v.tiles), dim3(256), 0, s, v, st.ranges,
                           st.header, sc.keys, st.point_list);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
