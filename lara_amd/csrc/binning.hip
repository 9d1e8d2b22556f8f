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

// how many workgroups share the sort of a list of `len` entries (see tile_sort_kernel)
constexpr int L2D_SORT_PARTS = 2;
__device__ __forceinline__ uint32_t l2d_sort_parts(const uint32_t len) {
    return len <= 4096u || len > 8192u ? 1u : 2u;
}

// ---- 2. exclusive scan of tile counts -> ranges ------------------------------------------------
__global__ void __launch_bounds__(1024)
tile_scan_kernel(ViewDev v, const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ sub_start,
                 uint2 *__restrict__ ranges, uint32_t *__restrict__ header,
                 uint32_t *__restrict__ tile_order, uint32_t *__restrict__ block_tot,
                 uint32_t *__restrict__ seg_base, uint32_t *__restrict__ seg_cnt,
                 uint32_t *__restrict__ bwd_order, uint2 *__restrict__ bwd_items,
                 uint32_t *__restrict__ sort_parts, uint2 *__restrict__ sort_items, const long long sst, const long long qst) {
    tile_count = l2d_view_ptr(tile_count, qst); sub_start = l2d_view_ptr(sub_start, qst); block_tot = l2d_view_ptr(block_tot, qst);
    sort_parts = l2d_view_ptr(sort_parts, qst); sort_items = l2d_view_ptr(sort_items, qst);
    ranges = l2d_view_ptr(ranges, sst); header = l2d_view_ptr(header, sst); tile_order = l2d_view_ptr(tile_order, sst);
    seg_base = l2d_view_ptr(seg_base, sst); seg_cnt = l2d_view_ptr(seg_cnt, sst); bwd_order = l2d_view_ptr(bwd_order, sst);
    bwd_items = l2d_view_ptr(bwd_items, sst);
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s, s_max;
    __shared__ uint32_t bcnt[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (blockIdx.x == 1) {   // the second workgroup: a scan that does not depend on the tiles, run beside the others
        // exclusive scan (in place) of the per-surfel-block pair totals -> surfel-major pair numbering
        if (tid == 0) carry_s = 0;
        __syncthreads();
        const int nblk = (v.P + 255) / 256;
        for (int base = 0; base < nblk; base += 1024) {
            const int i = base + tid;
            const uint32_t c = i < nblk ? block_tot[i] : 0u;
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
            const uint32_t incl = carry_s + wave_off + x;
            if (i < nblk) block_tot[i] = incl - c;
            __syncthreads();
            if (tid == 1023) carry_s = incl;
            __syncthreads();
        }
        return;
    }
    if (tid == 0) carry_s = 0;
    // this kernel is the first to touch the header (words 0-3 are assigned below, the rest are the composite
    // kernels' debug counters): zeroing it here saves the forward a memset launch
    if (tid >= 4 && tid < 64) header[tid] = 0u;
    __syncthreads();
    uint32_t max_len = 0;
    for (int base = 0; base < v.tiles; base += 1024) {
        const int i = base + tid;
        uint32_t sub[L2D_SLICES];
        uint32_t c = 0;
        if (i < v.tiles) {
            const uint4 *tc = (const uint4 *)(tile_count + (size_t)i * L2D_SLICES);
#pragma unroll
            for (int q = 0; q < L2D_SLICES / 4; q++) {
                const uint4 w4 = tc[q];
                sub[4 * q] = w4.x; sub[4 * q + 1] = w4.y; sub[4 * q + 2] = w4.z; sub[4 * q + 3] = w4.w;
            }
#pragma unroll
            for (int q = 0; q < L2D_SLICES; q++) c += sub[q];
        }
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
        if (i < v.tiles) {
            ranges[i] = c ? make_uint2(incl - c, incl) : make_uint2(0u, 0u);
            uint32_t run = incl - c;
            uint4 *ss = (uint4 *)(sub_start + (size_t)i * L2D_SLICES);
#pragma unroll
            for (int q = 0; q < L2D_SLICES / 4; q++) {
                uint4 w4;
                w4.x = run; run += sub[4 * q];
                w4.y = run; run += sub[4 * q + 1];
                w4.z = run; run += sub[4 * q + 2];
                w4.w = run; run += sub[4 * q + 3];
                ss[q] = w4;
            }
        }
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
        if (v.counts_out) {
            // the caller's copy of the pair count, in HOST memory: the reference reads this number with a blocking copy between
            // its scan and its duplicate-with-keys; here the caller spins on word 3 while the rest of the forward runs
            uint32_t *c = v.counts_out + 4 * blockIdx.z;
            __hip_atomic_store(&c[0], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&c[1], total > v.cap ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&c[2], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&c[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // Launch order of the composite kernels: longest list first (LPT), so that the heavy centre
    // tiles spread over all CUs instead of piling onto the few CUs their ids map to.  A 64-bucket
    // counting sort on the list length is enough (order inside a bucket is irrelevant).
    if (tid < 64) bcnt[tid] = 0;
    __syncthreads();
    const uint32_t denom = s_max + 1u;
    for (int i = tid; i < v.tiles; i += 1024) {
        const uint2 rg = ranges[i];
        const uint32_t b = 63u - (uint32_t)(((uint64_t)(rg.y - rg.x) * 64u) / denom);
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
        const uint2 rg = ranges[i];
        const uint32_t b = 63u - (uint32_t)(((uint64_t)(rg.y - rg.x) * 64u) / denom);
        tile_order[atomicAdd(&bcnt[b], 1u)] = (uint32_t)i;
    }
    // Backward work list.  A tile's list is cut into L2D_SEG-entry segments; nb = floor((len-1)/SEG)
    // of them are full and get an entry in bwd_items (and a checkpoint row, seg_base[tile] + s, at
    // their upper boundary); the last, partial one is launched per tile in bwd_order (longest first).
    // (A forward-only call has no such sections: it goes straight to the sort's work list.)
    __syncthreads();
    if (!v.fwd_only) {
    if (tid == 0) carry_s = 0;
    if (tid < 64) bcnt[tid] = 0;
    __syncthreads();
    const bool overflow = header[1] != 0u;  // then the lists are unusable and bwd_items may not fit
    for (int base = 0; base < v.tiles; base += 1024) {
        const int i = base + tid;
        uint32_t len = 0;
        if (i < v.tiles) { const uint2 rg = ranges[i]; len = rg.y - rg.x; }
        const uint32_t nb = len ? (len - 1u) / L2D_SEG : 0u;
        uint32_t x = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_sums[wid] = x;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; w++) wave_off += wave_sums[w];
        const uint32_t incl = carry_s + wave_off + x;
        if (i < v.tiles) {
            seg_base[i] = incl - nb;
            // (the checkpoint slab holds cap / L2D_SEG + 1 rows: every boundary of a frame that fits the capacity has one)
            seg_cnt[i] = nb;
            if (!overflow)
                for (uint32_t q = 0; q < nb; q++) bwd_items[incl - nb + q] = make_uint2((uint32_t)i, q);
            const uint32_t pl = len - nb * L2D_SEG;  // length of the last segment
            atomicAdd(&bcnt[63u - (uint32_t)(((uint64_t)min(pl, (uint32_t)L2D_SEG) * 64u) / (L2D_SEG + 1u))], 1u);
        }
        __syncthreads();
        if (tid == 1023) carry_s = incl;
        __syncthreads();
    }
    if (tid == 0) { seg_base[v.tiles] = carry_s; header[3] = carry_s; }
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
    for (int i = tid; i < v.tiles; i += 1024) {
        const uint2 rg = ranges[i];
        const uint32_t len = rg.y - rg.x;
        const uint32_t pl = len - seg_cnt[i] * L2D_SEG;
        bwd_order[atomicAdd(&bcnt[63u - (uint32_t)(((uint64_t)min(pl, (uint32_t)L2D_SEG) * 64u) / (L2D_SEG + 1u))], 1u)] = (uint32_t)i;
    }
    }   // !fwd_only
    // The sort's work list: a list longer than 2048 entries is shared by several workgroups (tile_sort_kernel), one
    // extra work item per additional part; at most `tiles` items (a tile that finds the list full gets fewer parts).
    __syncthreads();
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < v.tiles; base += 1024) {
        const int i = base + tid;
        uint32_t len = 0;
        if (i < v.tiles) { const uint2 rg = ranges[i]; len = rg.y - rg.x; }
        const uint32_t want = l2d_sort_parts(len) - 1u;
        uint32_t x = want;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_sums[wid] = x;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; w++) wave_off += wave_sums[w];
        const uint32_t incl = carry_s + wave_off + x, first = incl - want;
        if (i < v.tiles) {
            const uint32_t room = first < (uint32_t)v.tiles ? (uint32_t)v.tiles - first : 0u;
            const uint32_t got = want < room ? want : room;
            sort_parts[i] = 1u + got;
            for (uint32_t q = 0; q < got; q++) sort_items[first + q] = make_uint2((uint32_t)i, q + 1u);
        }
        __syncthreads();
        if (tid == 1023) carry_s = incl;
        __syncthreads();
    }
    if (tid == 0) sort_parts[v.tiles] = carry_s < (uint32_t)v.tiles ? carry_s : (uint32_t)v.tiles;
}

// ---- 3. scatter (depth bits, id) into the tile segments ----------------------------------------
// Two-level slot reservation: the workgroup counts its pairs per tile in LDS, reserves one
// contiguous run per touched tile with ONE returning device-scope atomic, then hands out slots
// inside the run with LDS atomics.  (Order inside a tile segment is arbitrary by design.)
__global__ void __launch_bounds__(256)
scatter_kernel(ViewDev v, const uint4 *__restrict__ rect, const uint32_t *__restrict__ sub_start,
               const uint32_t *__restrict__ block_base, uint32_t *__restrict__ pair_base,
               uint32_t *__restrict__ tile_fill, uint64_t *__restrict__ keys, const int use_lds, const long long sst,
               const long long qst) {
    rect = l2d_view_ptr(rect, qst); sub_start = l2d_view_ptr(sub_start, qst); block_base = l2d_view_ptr(block_base, qst);
    tile_fill = l2d_view_ptr(tile_fill, qst); keys = l2d_view_ptr(keys, qst); pair_base = l2d_view_ptr(pair_base, sst);
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int slice = blockIdx.x % L2D_SLICES;
    int rx0 = 0, ry0 = 0, rx1 = 0, ry1 = 0;
    uint64_t word = 0;
    if (idx < v.P) {
        const uint4 r = rect[idx];
        rx0 = r.x & 0xffff; ry0 = r.x >> 16; rx1 = r.y & 0xffff; ry1 = r.y >> 16;
        word = ((uint64_t)r.z << 32) | (uint32_t)idx;
        if (!v.fwd_only) {      // surfel-major pair numbering: where the backward writes a pair's gradient row
            const uint32_t pb = block_base[blockIdx.x] + r.w;
            pair_base[idx] = pb;
            if (idx == v.P - 1) pair_base[v.P] = pb + (uint32_t)(rx1 - rx0) * (uint32_t)(ry1 - ry0);
        }
    }
    if (use_lds) {
        for (int t = threadIdx.x; t < v.tiles; t += blockDim.x) hist[t] = 0;
        __syncthreads();
        for (int y = ry0; y < ry1; y++)
            for (int x = rx0; x < rx1; x++) atomicAdd(&hist[y * v.gx + x], 1u);
        __syncthreads();
        for (int t = threadIdx.x; t < v.tiles; t += blockDim.x) {
            const uint32_t c = hist[t];
            if (c) hist[t] = sub_start[t * L2D_SLICES + slice] +
                             __hip_atomic_fetch_add(&tile_fill[t * L2D_SLICES + slice], c, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int y = ry0; y < ry1; y++)
            for (int x = rx0; x < rx1; x++) {
                const uint32_t slot = atomicAdd(&hist[y * v.gx + x], 1u);
                if (slot < v.cap) keys[slot] = word;
            }
    } else {
        for (int y = ry0; y < ry1; y++)
            for (int x = rx0; x < rx1; x++) {
                const int t = (y * v.gx + x) * L2D_SLICES + slice;
                const uint32_t slot = sub_start[t] + __hip_atomic_fetch_add(&tile_fill[t], 1u, __ATOMIC_RELAXED,
                                                                          __HIP_MEMORY_SCOPE_AGENT);
                if (slot < v.cap) keys[slot] = word;
            }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t n) {
    return n <= 1 ? 1u : 1u << (32 - __clz(n - 1));
}

// ---- 4. per-tile sort -----------------------------------------------------------------------------
// Register-blocked bitonic network: a 256-thread workgroup sorts 256*E keys, thread t holding the
// E consecutive keys [t*E, t*E+E).  Comparators whose partner lives in the same thread are plain
// register compare-exchanges; partners in the same wave are reached with ds_bpermute; only
// partners in another wave (thread distance >= 64) go through LDS (element-major, conflict free).
// Padding keys are ~0 and, the network being all-ascending, stay above every real key.
__device__ __forceinline__ void ce(uint64_t &a, uint64_t &b) {  // a <- min, b <- max
    const bool sw = b < a;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t x, int src_lane) {
    const uint32_t lo = __shfl((uint32_t)x, src_lane, 64), hi = __shfl((uint32_t)(x >> 32), src_lane, 64);
    return ((uint64_t)hi << 32) | lo;
}

// compare-exchange every key with partner thread pt's key (REVERSED: its key E-1-e, the mirror
// step that opens a merge); the lower-numbered thread keeps the minima
template <int E, int NT, bool REVERSED>
__device__ __forceinline__ void xchg(uint64_t (&x)[E], const int t, const int pt, uint64_t *lds) {
    uint64_t o[E];
    if ((t ^ pt) < 64) {  // partner in the same wave (uniform: the xor distance is common to all t)
#pragma unroll
        for (int e = 0; e < E; e++) o[e] = shfl64(x[REVERSED ? E - 1 - e : e], pt & 63);
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) lds[e * NT + t] = x[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; e++) o[e] = lds[(REVERSED ? E - 1 - e : e) * NT + pt];
        __syncthreads();
    }
    const bool keep_min = t < pt;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const bool take = keep_min ? (o[e] < x[e]) : (x[e] < o[e]);
        x[e] = take ? o[e] : x[e];
    }
}

template <int E, int NT>
__device__ __forceinline__ void block_sort(uint64_t (&x)[E], uint64_t *lds) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 2; k <= E; k <<= 1) {  // merges that fit inside a thread
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int p = e ^ (k - 1);
            if (p > e) ce(x[e], x[p]);
        }
#pragma unroll
        for (int j = k >> 2; j >= 1; j >>= 1)
#pragma unroll
            for (int e = 0; e < E; e++)
                if ((e & j) == 0) ce(x[e], x[e | j]);
    }
#pragma unroll 1
    for (int kt = 2; kt <= NT; kt <<= 1) {  // merges of kt threads' worth of keys
        xchg<E, NT, true>(x, t, t ^ (kt - 1), lds);
#pragma unroll 1
        for (int jt = kt >> 2; jt >= 1; jt >>= 1) xchg<E, NT, false>(x, t, t ^ jt, lds);
#pragma unroll
        for (int j = E >> 1; j >= 1; j >>= 1)
#pragma unroll
            for (int e = 0; e < E; e++)
                if ((e & j) == 0) ce(x[e], x[e | j]);
    }
}

// where (surfel id, this tile) sits in the surfel-major pair numbering: recorded per list position, so that the
// backward can write its gradient rows surfel-major (a surfel's rows contiguous: preprocess_bwd streams them)
struct PairMap {
    const uint4 *rect;
    const uint32_t *pair_base;
    uint32_t *pair_pos;
    int tx, ty;
    uint32_t first;  // position of the tile's first entry in the sorted list
    __device__ __forceinline__ void put(const uint32_t id, const uint32_t i) const {
        if (!pair_pos) return;      // forward-only call: nobody will gather gradient rows
        const uint4 r = rect[id];
        const int rx0 = r.x & 0xffff, ry0 = r.x >> 16, rx1 = r.y & 0xffff;
        pair_pos[first + i] = pair_base[id] + (uint32_t)((ty - ry0) * (rx1 - rx0) + (tx - rx0));
    }
};

template <int E, int NT>
__device__ __forceinline__ void sort_tile(const uint64_t *__restrict__ seg, const uint32_t n,
                                          uint32_t *__restrict__ out, uint64_t *lds, const PairMap &pm) {
    uint64_t x[E];
    const uint32_t i0 = threadIdx.x * E;
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = (i0 + e < n) ? seg[i0 + e] : ~0ull;
    block_sort<E, NT>(x, lds);
#pragma unroll
    for (int e = 0; e < E; e++)
        if (i0 + e < n) {
            out[i0 + e] = (uint32_t)x[e];
            pm.put((uint32_t)x[e], i0 + e);
        }
}

// One launch of 512-thread workgroups in longest-list-first order; up to 4096 keys are sorted in 32 KB of LDS (four
// workgroups per CU) by sort_keys().  A longer list (round 1: a second launch of 1024-thread workgroups for those
// few lists lasted 64 us on its own) is cut by KEY RANGE into L2D_SORT_PARTS parts that different workgroups sort
// independently: part k takes the keys whose
// depth falls into the k-th of S equal slices of the list's [min, max] depth interval, knows where its output starts
// (the number of keys in lower slices, counted on its own pass over the segment: no communication between
// workgroups), compacts its keys into LDS in arbitrary order and sorts them.  Concatenated parts = the sorted list,
// whatever the depth distribution (a skewed one only balances worse; a slice beyond the LDS buffer makes part 0 sort
// the whole list in place).  Lists beyond 8192 entries (not reached at LaRa's sizes) take that in-place path too.
constexpr uint32_t L2D_SORT_LDS_KEYS = 4096;   // 32 KB: four 512-thread workgroups per CU, as many as its wave slots hold

// Sort m <= 4096 keys (from the global segment, or already compacted into `lds`) and write ids + pair_pos at `off`.
//
// Bucket sort: the keys are dealt into B ~ m/4 buckets by depth (equal slices of the [min, max] interval -- LaRa's
// depths inside a tile are close to uniform), each bucket is put in order by one thread with an insertion sort on the
// full 64-bit (depth, id) word -- ties come out id-ascending, the reference's stable-sort order, without any
// stability requirement on the passes before -- and the concatenated buckets are the sorted list.  ~60 operations and
// 7 barriers per key against ~55 compare-exchange stages (most of them with a cross-lane or LDS exchange) of the
// bitonic network it replaces; that network remains the fallback for a depth distribution that overfills a bucket
// (a wall of equal depths).
constexpr int L2D_SORT_BUCKETS = 1024, L2D_SORT_BUCKET_MAX = 48;

template <int E>
__device__ __forceinline__ void sort_keys(const uint64_t *src, const uint32_t m, uint32_t *__restrict__ out,
                                          const uint32_t off, uint64_t *lds, uint32_t *bkt, uint32_t *sh, const PairMap &pm) {
    const int t = threadIdx.x, lane = t & 63;
    uint64_t x[E];
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = (t + e * 512 < (int)m) ? src[t + e * 512] : ~0ull;
    if (t == 0) { sh[0] = ~0u; sh[1] = 0u; sh[2] = 0u; }
    const uint32_t B = m <= 256u ? 64u : (m <= 1024u ? 256u : (uint32_t)L2D_SORT_BUCKETS);
    for (uint32_t b = t; b < B; b += 512) bkt[b] = 0u;
    __syncthreads();   // (also: every key of `src` is in registers, `lds` may be overwritten from here on)
    uint32_t lo = ~0u, hi = 0u;
#pragma unroll
    for (int e = 0; e < E; e++)
        if (t + e * 512 < (int)m) {
            const uint32_t d = (uint32_t)(x[e] >> 32);
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
        const uint32_t a = __shfl_xor(lo, k, 64), b = __shfl_xor(hi, k, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if (lane == 0) { atomicMin(&sh[0], lo); atomicMax(&sh[1], hi); }
    __syncthreads();
    const uint32_t dmin = sh[0];
    // bucket = floor((d - dmin) * B / (range + 1)) as a multiply-high by a per-tile constant (monotone in d, < B)
    const uint64_t scale = ((uint64_t)B << 32) / ((uint64_t)(sh[1] - dmin) + 1ull);
    uint32_t bk[E], slot[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        bk[e] = 0u; slot[e] = 0u;
        if (t + e * 512 < (int)m) {
            const uint64_t q = (uint64_t)((uint32_t)(x[e] >> 32) - dmin) * scale;
            bk[e] = (uint32_t)(q >> 32);
            bk[e] = bk[e] < B ? bk[e] : B - 1u;   // (scale is rounded down: the product cannot reach B; belt and braces)
            slot[e] = atomicAdd(&bkt[bk[e]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the B bucket sizes (two per thread), the largest bucket on the side
    {
        const uint32_t c0 = 2 * t < (int)B ? bkt[2 * t] : 0u, c1 = 2 * t + 1 < (int)B ? bkt[2 * t + 1] : 0u;
        uint32_t incl = c0 + c1, big = c0 > c1 ? c0 : c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if (lane >= d) incl += y;
        }
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) { const uint32_t y = __shfl_xor(big, k, 64); big = y > big ? y : big; }
        if (lane == 63) sh[4 + (t >> 6)] = incl;
        if (lane == 0) atomicMax(&sh[2], big);
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < (t >> 6); w++) wave_off += sh[4 + w];
        __syncthreads();
        if (2 * t < (int)B) bkt[2 * t] = wave_off + incl - c0 - c1;
        if (2 * t + 1 < (int)B) bkt[2 * t + 1] = wave_off + incl - c1;
        if (t == 511) bkt[B] = wave_off + incl;   // = m
    }
    __syncthreads();
    if (sh[2] > (uint32_t)L2D_SORT_BUCKET_MAX) {   // uniform: a degenerate depth distribution -> the sorting network
        __syncthreads();
        block_sort<E, 512>(x, lds);
#pragma unroll
        for (int e = 0; e < E; e++)
            if ((uint32_t)(t * E + e) < m) {
                out[off + t * E + e] = (uint32_t)x[e];
                pm.put((uint32_t)x[e], off + t * E + e);
            }
        return;
    }
#pragma unroll
    for (int e = 0; e < E; e++)
        if (t + e * 512 < (int)m) lds[bkt[bk[e]] + slot[e]] = x[e];
    __syncthreads();
    for (uint32_t b = t; b < B; b += 512) {      // one thread per bucket: insertion sort on the 64-bit words
        const uint32_t s0 = bkt[b], s1 = bkt[b + 1];
        for (uint32_t i = s0 + 1; i < s1; i++) {
            const uint64_t key = lds[i];
            uint32_t j = i;
            while (j > s0 && lds[j - 1] > key) { lds[j] = lds[j - 1]; j--; }
            lds[j] = key;
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < m; i += 512) {
        const uint32_t id = (uint32_t)lds[i];
        out[off + i] = id;
        pm.put(id, off + i);
    }
}

template <int DUMMY = 0>
__device__ __forceinline__ void sort_dispatch(const uint64_t *src, const uint32_t m, uint32_t *__restrict__ out,
                                              const uint32_t off, uint64_t *lds, uint32_t *bkt, uint32_t *sh, const PairMap &pm) {
    if (m <= 512u) sort_keys<1>(src, m, out, off, lds, bkt, sh, pm);
    else if (m <= 1024u) sort_keys<2>(src, m, out, off, lds, bkt, sh, pm);
    else if (m <= 2048u) sort_keys<4>(src, m, out, off, lds, bkt, sh, pm);
    else sort_keys<8>(src, m, out, off, lds, bkt, sh, pm);
}

__global__ void __launch_bounds__(512)
tile_sort_kernel(ViewDev v, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ header,
                 const uint32_t *__restrict__ tile_order, uint64_t *__restrict__ keys,
                 uint32_t *__restrict__ point_list, const uint4 *__restrict__ rect,
                 const uint32_t *__restrict__ pair_base, uint32_t *__restrict__ pair_pos,
                 uint32_t *__restrict__ sort_parts, const uint2 *__restrict__ sort_items, const long long sst, const long long qst) {
    ranges = l2d_view_ptr(ranges, sst); header = l2d_view_ptr(header, sst); tile_order = l2d_view_ptr(tile_order, sst);
    point_list = l2d_view_ptr(point_list, sst); pair_base = l2d_view_ptr(pair_base, sst); pair_pos = l2d_view_ptr(pair_pos, sst);
    keys = l2d_view_ptr(keys, qst); rect = l2d_view_ptr(rect, qst); sort_parts = l2d_view_ptr(sort_parts, qst);
    sort_items = l2d_view_ptr(sort_items, qst);
    __shared__ uint64_t lds[L2D_SORT_LDS_KEYS];
    __shared__ uint32_t bkt[L2D_SORT_BUCKETS + 1], sh[16];
    __shared__ uint32_t s_min, s_max, s_mine, s_ticket, s_hist[L2D_SORT_PARTS];
    // Grid: `tiles` extra work items first (tile_scan's list of (tile, part >= 1) for the long lists; the unused ones
    // exit on one load), then part 0 of every tile in longest-list-first order.  (A grid of parts x tiles workgroups
    // that find out for themselves that they are not needed costs 20 us in dependent loads before they exit.)
    int part = 0, tile;
    if ((int)blockIdx.x < v.tiles) {
        if (blockIdx.x >= sort_parts[v.tiles]) return;
        const uint2 it = sort_items[blockIdx.x];
        tile = (int)it.x; part = (int)it.y;
    } else {
        tile = (int)tile_order[(int)blockIdx.x - v.tiles];
    }
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (header[1]) return;  // capacity overflow: lists are incomplete, outputs get poisoned instead
    if (n == 0) return;
    const int S = (int)(sort_parts[tile] & 0xffffu);   // (bits 16+: the parts' arrival tickets, see the fallback below)
    uint64_t *seg = keys + rg.x;
    uint32_t *out = point_list + rg.x;
    const PairMap pm{rect, pair_base, v.fwd_only ? nullptr : pair_pos, tile % v.gx, tile / v.gx, rg.x};
    bool whole_list_in_place = S == 1 && n > L2D_SORT_LDS_KEYS;   // beyond 8192 entries, or no work items left for this tile
    if (S == 1 && !whole_list_in_place) {
        sort_dispatch(seg, n, out, 0u, lds, bkt, sh, pm);
        return;
    }
    if (S > 1) {
        // ---- the S equal slices of the list's depth interval; every part counts all of them (same data, same
        //      result in every workgroup of the tile: no communication needed to agree on the fallback below)
        if (threadIdx.x == 0) { s_min = ~0u; s_max = 0u; s_mine = 0u; }
        if (threadIdx.x < L2D_SORT_PARTS) s_hist[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t lo = ~0u, hi = 0u;
        for (uint32_t i = threadIdx.x; i < n; i += 512) {
            const uint32_t d = (uint32_t)(seg[i] >> 32);
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) {
            const uint32_t a = __shfl_xor(lo, k, 64), b = __shfl_xor(hi, k, 64);
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
        __syncthreads();
        const uint32_t dmin = s_min;
        const uint64_t width = (uint64_t)(s_max - dmin) + 1ull;
        uint32_t cnt[L2D_SORT_PARTS];
#pragma unroll
        for (int b = 0; b < L2D_SORT_PARTS; b++) cnt[b] = 0u;
        for (uint32_t i = threadIdx.x; i < n; i += 512) {   // (the segment is re-read from L2: 8 B per key)
            const int bin = (int)(((uint64_t)((uint32_t)(seg[i] >> 32) - dmin) * (uint64_t)S) / width);
#pragma unroll
            for (int b = 0; b < L2D_SORT_PARTS; b++) cnt[b] += bin == b ? 1u : 0u;
        }
#pragma unroll
        for (int b = 0; b < L2D_SORT_PARTS; b++) {
#pragma unroll
            for (int k = 32; k > 0; k >>= 1) cnt[b] += __shfl_xor(cnt[b], k, 64);
            if ((threadIdx.x & 63) == 0 && cnt[b]) atomicAdd(&s_hist[b], cnt[b]);
        }
        __syncthreads();
        uint32_t off = 0, biggest = 0;
        for (int b = 0; b < S; b++) {
            off += b < part ? s_hist[b] : 0u;
            biggest = s_hist[b] > biggest ? s_hist[b] : biggest;
        }
        if (biggest > L2D_SORT_LDS_KEYS) {
            // A slice that does not fit the LDS (a wall of equal depths): ONE workgroup sorts the whole list in place
            // instead.  That mutates `seg`, which the tile's other parts read for the passes above, and the parts are
            // unordered workgroups -- so the LAST part to arrive here does it: each takes a ticket once its own reads
            // of `seg` are done (they fed the reductions above), nobody writes before all S tickets are taken, hence
            // every part derived min / max / histogram from the untouched segment and took this same branch.
            if (threadIdx.x == 0) {
                __threadfence();
                s_ticket = __hip_atomic_fetch_add(&sort_parts[tile], 0x10000u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) >> 16;
            }
            __syncthreads();
            if (s_ticket != (uint32_t)S - 1u) return;
            whole_list_in_place = true;
        } else {
            const uint32_t m = s_hist[part];
            if (m == 0) return;
            for (uint32_t i0 = 0; i0 < n; i0 += 512) {      // (uniform trip count: the ballot below needs the whole wave)
                const uint32_t i = i0 + threadIdx.x;
                const uint64_t key = i < n ? seg[i] : 0ull;
                const int bin = i < n ? (int)(((uint64_t)((uint32_t)(key >> 32) - dmin) * (uint64_t)S) / width) : -1;
                // one LDS atomic per wave, slots inside the wave by rank (not one same-address atomic per key; the order is arbitrary anyway: the sort follows
                const unsigned long long mine = __ballot(bin == part);
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0 && mine) base = atomicAdd(&s_mine, (uint32_t)__builtin_popcountll(mine));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (bin == part) lds[base + (uint32_t)__builtin_popcountll(mine & ((1ull << (threadIdx.x & 63)) - 1ull))] = key;
            }
            __syncthreads();
            sort_dispatch(lds, m, out, off, lds, bkt, sh, pm);
            return;
        }
    }
    // generic network run directly on the global segment (lists beyond 8192 entries, degenerate depth distributions)
    bitonic_sort(seg, n, next_pow2(n));
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        out[i] = (uint32_t)seg[i];
        pm.put((uint32_t)seg[i], i);
    }
}

// ---- the fine pass's lists as a FILTER of the coarse pass's (lara2dgs_forward_views_subset) ------------------------------------
// LaRa's fine pass (network.py:502-525) renders a subset of the surfels the coarse pass has just rendered, from the same cameras,
// with the same geometry -- only the colours differ.  Its per-tile lists are therefore the coarse lists with the dropped surfels
// taken out and the ids renumbered: the sort key is (depth bits, id), the subset's numbering is monotone in the coarse one, so a
// STABLE filter of a coarse list IS the fine list, tie order included.  The subset's preprocess still runs (its colours, and with
// them the records, are its own; its per-tile counts give tile_scan the ranges), scatter + sort do not: a per-surfel pass writes the
// surfel-major pair numbering scatter would have written, a per-tile pass compacts.
__global__ void __launch_bounds__(256)
pair_base_kernel(ViewDev v, const uint4 *__restrict__ rect, const uint32_t *__restrict__ block_base, uint32_t *__restrict__ pair_base,
                 const long long sst, const long long qst) {
    rect = l2d_view_ptr(rect, qst); block_base = l2d_view_ptr(block_base, qst); pair_base = l2d_view_ptr(pair_base, sst);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= v.P) return;
    const uint4 r = rect[idx];
    const uint32_t pb = block_base[blockIdx.x] + r.w;
    pair_base[idx] = pb;
    if (idx == v.P - 1) {
        const int rx0 = r.x & 0xffff, ry0 = r.x >> 16, rx1 = r.y & 0xffff, ry1 = r.y >> 16;
        pair_base[v.P] = pb + (uint32_t)(rx1 - rx0) * (uint32_t)(ry1 - ry0);
    }
}

// one workgroup per (tile, view): the coarse list in order, kept entries to the fine list at start + rank (ballot ranks inside a wave,
// wave totals through LDS in wave order: the compaction is stable); a kept entry's pair index in the fine pass's surfel-major
// numbering = its surfel's first pair there + the tile's offset inside the surfel's rectangle, which the coarse pair map knows
__global__ void __launch_bounds__(256)
subset_compact_kernel(ViewDev v, const uint32_t *__restrict__ header, const uint2 *__restrict__ ranges, uint32_t *__restrict__ point_list,
                      const uint32_t *__restrict__ pair_base, uint32_t *__restrict__ pair_pos, const long long sst,
                      const uint2 *__restrict__ c_ranges, const uint32_t *__restrict__ c_point_list, const uint32_t *__restrict__ c_pair_base,
                      const uint32_t *__restrict__ c_pair_pos, const long long csst, const int32_t *__restrict__ inv) {
    header = l2d_view_ptr(header, sst); ranges = l2d_view_ptr(ranges, sst); point_list = l2d_view_ptr(point_list, sst);
    pair_base = l2d_view_ptr(pair_base, sst); pair_pos = l2d_view_ptr(pair_pos, sst);
    c_ranges = l2d_view_ptr(c_ranges, csst); c_point_list = l2d_view_ptr(c_point_list, csst);
    c_pair_base = l2d_view_ptr(c_pair_base, csst); c_pair_pos = l2d_view_ptr(c_pair_pos, csst);
    __shared__ uint32_t wtot[2][4];
    if (header[1]) return;      // capacity overflow: the outputs get poisoned instead
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint2 cr = c_ranges[tile];
    const uint32_t n = cr.y - cr.x;
    uint32_t out = ranges[tile].x;
    // (a trip's dependent chain -- id, its row in the subset, the two pair numbers -- is started one trip ahead: the kernel is a
    // latency chain per tile, 6 trips of 256 entries at LaRa's list lengths; one barrier per trip, the wave totals double-buffered)
    uint32_t id_n = tid < n ? c_point_list[cr.x + tid] : 0u;
    int fid_n = tid < n ? inv[id_n] : -1;
    int buf = 0;
    for (uint32_t base = 0; base < n; base += 256, buf ^= 1) {      // (uniform trip count: the ballots need whole waves)
        const uint32_t i = base + tid, id = id_n;
        const int fid = fid_n;
        uint32_t off = 0, pbf = 0;
        if (fid >= 0 && !v.fwd_only) { off = c_pair_pos[cr.x + i] - c_pair_base[id]; pbf = pair_base[fid]; }
        if (base + 256 < n) {
            id_n = i + 256 < n ? c_point_list[cr.x + i + 256] : 0u;
            fid_n = i + 256 < n ? inv[id_n] : -1;
        }
        const unsigned long long kept = __ballot(fid >= 0);
        if (lane == 0) wtot[buf][wave] = (uint32_t)__builtin_popcountll(kept);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { before += w < wave ? wtot[buf][w] : 0u; all += wtot[buf][w]; }
        if (fid >= 0) {
            const uint32_t pos = out + before + (uint32_t)__builtin_popcountll(kept & ((1ull << lane) - 1ull));
            point_list[pos] = (uint32_t)fid;
            if (!v.fwd_only) pair_pos[pos] = pbf + off;
        }
        out += all;
    }
}

}  // namespace

int launch_binning_subset(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s, const ViewBatch *vb, StateView cst,
                          long long coarse_stride, const int32_t *inv) {
    const unsigned nz = vb ? (unsigned)vb->n : 1u;
    const long long sst = vb ? vb->state_stride : 0, qst = vb ? vb->scratch_stride : 0;
    {
        L2D_PROF("tile_scan", s);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(v.P > 0 ? 2 : 1, 1, nz), dim3(1024), 0, s, v, sc.tile_count, sc.sub_start,
                           st.ranges, st.header, st.tile_order, sc.block_tot, st.seg_base, st.seg_cnt, st.bwd_order,
                           st.bwd_items, sc.sort_parts, sc.sort_items, sst, qst);
    }
    L2D_CHECK_LAUNCH();
    if (v.P == 0) return LARA2DGS_OK;
    if (!v.fwd_only) {
        L2D_PROF("subset_pair_base", s);
        hipLaunchKernelGGL(pair_base_kernel, dim3((v.P + 255) / 256, 1, nz), dim3(256), 0, s, v, sc.rect, sc.block_tot, st.pair_base, sst, qst);
    }
    {
        L2D_PROF("subset_compact", s);
        hipLaunchKernelGGL(subset_compact_kernel, dim3(v.tiles, 1, nz), dim3(256), 0, s, v, st.header, st.ranges, st.point_list,
                           st.pair_base, st.pair_pos, sst, cst.ranges, cst.point_list, cst.pair_base, cst.pair_pos, coarse_stride, inv);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int launch_binning(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s, const ViewBatch *vb) {
    // vb != nullptr: the binning of ALL views of a multi-view call in three launches (blockIdx.z = view; st / sc are view 0's)
    const unsigned nz = vb ? (unsigned)vb->n : 1u;
    const long long sst = vb ? vb->state_stride : 0, qst = vb ? vb->scratch_stride : 0;
    {
        L2D_PROF("tile_scan", s);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(v.P > 0 ? 2 : 1, 1, nz), dim3(1024), 0, s, v, sc.tile_count, sc.sub_start,
                           st.ranges, st.header, st.tile_order, sc.block_tot, st.seg_base, st.seg_cnt, st.bwd_order,
                           st.bwd_items, sc.sort_parts, sc.sort_items, sst, qst);
    }
    L2D_CHECK_LAUNCH();
    if (v.P == 0) return LARA2DGS_OK;
    {
        L2D_PROF("scatter", s);
        const int use_lds = v.tiles <= L2D_LDS_HIST_TILES;
        hipLaunchKernelGGL(scatter_kernel, dim3((v.P + 255) / 256, 1, nz), dim3(256),
                           use_lds ? (size_t)v.tiles * 4 : 0, s, v, sc.rect, sc.sub_start, sc.block_tot,
                           st.pair_base, sc.tile_fill, sc.keys, use_lds, sst, qst);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("tile_sort", s);
        hipLaunchKernelGGL(tile_sort_kernel, dim3(v.tiles * 2, 1, nz), dim3(512), 0, s, v, st.ranges, st.header, st.tile_order,
                           sc.keys, st.point_list, sc.rect, st.pair_base, st.pair_pos, sc.sort_parts, sc.sort_items, sst, qst);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
