// common.h -- shared declarations of the gfx950 surfel rasteriser kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lara2dgs.h"

#define TILE 16            // tile edge in pixels: the binning contract (tile ids / ranges are bit-exact)
#define NEAR_N 0.2f
#define FAR_N 100.0f
#define FILTER_SIZE 0.707106f
#define FILTER_INV_SQUARE 2.0f
#define CUTOFF 3.0f
#define GEOM_F 20          // floats per surfel record in the state buffer
#define L2D_SLICES 16            // per-tile counters are split 16 ways to shorten same-address atomic chains
#define L2D_LDS_HIST_TILES 8192  // per-workgroup LDS tile histogram up to this many tiles (32 KB)
#define GRAD_F 20          // floats per surfel in the backward accumulator (18 used)
#ifndef L2D_SEG
#define L2D_SEG 512
#endif                    // backward work unit: a tile's list is cut into segments of this many entries (1024, round 6: the per-view
                          // backward 351 -> 403 us, trained-like 90 -> 150 us -- the launch's tail --, the 8-view step unchanged)
#define L2D_CKPT_F 10      // floats per pixel in a segment-boundary checkpoint / in the per-pixel finals
#define L2D_MAX_VIEWS 8    // cameras of one batched preprocess launch (a multi-view call with more views issues several)

// Everything a kernel needs to know about the view, passed by value (lands in SGPRs / kernarg).
struct ViewDev {
    int P, deg, M, H, W, gx, gy, tiles;
    float scale_modifier;
    unsigned cap;  // capacity in (tile, surfel) pairs
    unsigned dbg;  // LARA2DGS_DEBUG_FLAGS (perf experiments only; 0 in production)
    int cull_transparent;  // opt-in: surfels with opacity < 1/255 are culled in preprocess (lara2dgs_view.prefiltered bit 1)
    int fwd_only;          // lara2dgs_view.forward_only: nothing is kept for a backward
    const float *bg, *viewmatrix, *projmatrix, *campos;
    uint32_t *counts_out;  // host-visible uint32[4] of view 0 (view z: + 4 z), or null: lara2dgs_view.counts_out
};

// A multi-view call carves the views' `state` / `scratch` buffers out of one allocation each at a constant stride, and all
// views agree in every size: the binning and composite kernels of ALL views are then ONE launch each -- blockIdx.z = view, every
// state / scratch / output pointer advanced by view * stride (0 and gridDim.z = 1 for a single view).  Only the background
// colours are separate tensors per view.
struct ViewBatch {
    int n;                      // views of this launch (<= L2D_MAX_VIEWS)
    long long state_stride, scratch_stride;   // bytes between consecutive views' buffers
    const float *bg[L2D_MAX_VIEWS];
};
#ifdef __HIPCC__
// (pointer arithmetic on the pointer itself: through an integer the compiler loses the kernel argument's address space and
// every access of the kernel becomes a flat_load / flat_store with a 64-bit VGPR address instead of global_load with an SGPR
// base -- measured: composite_bwd 411 -> 431 us)
template <class T>
__device__ __forceinline__ T *l2d_view_ptr(T *p, const long long stride_bytes) {
    return (T *)((const char *)p + (long long)blockIdx.z * stride_bytes);
}
#endif

struct StateView {  // typed pointers into the caller's `state` buffer
    uint32_t *header;
    float4 *geom;         // [P][5] float4
    float4 *cullbox;      // [P] minx,maxx,miny,maxy: conservative box of {alpha >= 1/255}
    uint32_t *point_list; // [cap]
    uint2 *ranges;        // [tiles]
    uint32_t *tile_order; // [tiles] tile ids, longest list first (work-balanced launch order)
    uint32_t *pair_base;  // [P+1] first pair index of each surfel (exclusive scan of tiles touched)
    uint32_t *pair_pos;   // [cap] position in the sorted list -> pair index in surfel-major numbering
    float *final_T;       // [L2D_CKPT_F][HW] end-of-walk T, M1, M2, C(3), D, N(3)
    uint32_t *n_contrib;  // [2][HW]
    uint32_t *seg_base;   // [tiles+1] exclusive scan of interior segment boundaries per tile
    uint32_t *seg_cnt;    // [tiles] interior segment boundaries of the tile's list = (len - 1) / L2D_SEG
    uint32_t *bwd_order;  // [tiles] tile ids by length of their last (partial) segment, longest first
    uint2 *bwd_items;     // [cap/L2D_SEG+1 + tiles] (tile, segment) of every full segment; once ordered (header[22]): of EVERY work item
    float *ckpt;          // [cap/L2D_SEG+1][L2D_CKPT_F][256] per-pixel prefix sums at segment boundaries
    uint2 *pair_mask;     // [cap] per list position: the forward's 64-bit candidate mask (lo, hi) of the entry
    uint32_t *tile_maxc;  // [tiles] max over the tile's pixels of the last contributor (list position + 1), from the forward
    uint32_t *seg_cost;   // [cap/L2D_SEG+1 + tiles] forward wave-trips per backward work item (full segments, then last segments)
};

struct ScratchView {
    uint32_t *tile_count;  // [tiles][L2D_SLICES]  population per (tile, surfel-block slice)
    uint32_t *tile_fill;   // [tiles][L2D_SLICES]  scatter cursors
    uint32_t *sub_start;   // [tiles][L2D_SLICES]  first slot of each (tile, slice) sub-segment
    uint32_t *sort_parts;  // [tiles+1]  workgroups sharing a tile's sort (1 for short lists); [tiles] = number of sort_items
    uint2 *sort_items;     // [tiles]  (tile, part >= 1): the extra sort workgroups of the long lists
    uint4 *rect;           // [P] tile rectangle (4 x u16 in .x,.y) + depth bits (.z)
    uint64_t *keys;        // [cap]  (depth bits << 32) | surfel id, grouped per tile, unsorted
    uint32_t *block_tot;   // [ceil(P/256)] pairs per surfel block, then (in place) their exclusive scan
    float4 *pair_grad;     // [cap][5] backward: per (tile, surfel) gradient rows, surfel-major (aliases the forward region)
    uint32_t *pair_valid;  // [cap] bytes, backward: byte q != 0 <=> gradient row q (surfel-major) was written
};

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Checkpoint rows: the worst case.  A tile of `len` entries has (len - 1) / L2D_SEG interior boundaries, so a frame of
// D <= cap pairs never needs more than cap / L2D_SEG rows (round 5: the slab used to be half of that, and a frame beyond half
// the capacity ran some tiles unsegmented; with the capacity following the measured pair count -- rasterizer.py -- frames
// between cap / 2 and cap are the normal case, and 20 bytes per pair of capacity buy them the segmented backward).
__host__ __device__ static inline int64_t l2d_ckpt_slots(int64_t cap) { return cap / L2D_SEG + 1; }

// fwd_only: the sections only a backward reads have size 0 (offset = the next section's); the kernels of such a call never
// touch them.
static inline void state_layout(int P, int H, int W, int64_t cap, int fwd_only, lara2dgs_state_layout *L) {
    const int64_t tiles = (int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const int64_t HW = (int64_t)H * W;
    const int64_t k = fwd_only ? 0 : 1;     // multiplies the size of every backward-only section
    int64_t o = 0;
    L->header = o;      o = align_up(o + 256, 256);
    L->geom = o;        o = align_up(o + (int64_t)P * GEOM_F * 4, 256);
    L->cullbox = o;     o = align_up(o + (int64_t)P * 16, 256);
    L->point_list = o;  o = align_up(o + cap * 4, 256);
    L->ranges = o;      o = align_up(o + tiles * 8, 256);
    L->tile_order = o;  o = align_up(o + tiles * 4, 256);
    L->pair_base = o;   o = align_up(o + k * ((int64_t)P + 1) * 4, 256);
    L->pair_pos = o;    o = align_up(o + k * cap * 4, 256);
    L->final_T = o;     o = align_up(o + k * L2D_CKPT_F * HW * 4, 256);
    L->n_contrib = o;   o = align_up(o + k * 2 * HW * 4, 256);
    const int64_t nseg = cap / L2D_SEG + 1;
    L->seg_base = o;    o = align_up(o + k * (tiles + 1) * 4, 256);
    L->seg_cnt = o;     o = align_up(o + k * tiles * 4, 256);
    L->bwd_order = o;   o = align_up(o + k * tiles * 4, 256);
    L->bwd_items = o;   o = align_up(o + k * (nseg + tiles) * 8, 256);   // (+ tiles: the ordered list also holds every tile's last segment)
    L->ckpt = o;        o = align_up(o + k * l2d_ckpt_slots(cap) * L2D_CKPT_F * 256 * 4, 256);
    L->pair_mask = o;   o = align_up(o + k * cap * 8, 256);
    L->tile_maxc = o;   o = align_up(o + k * tiles * 4, 256);
    L->seg_cost = o;    o = align_up(o + k * (nseg + tiles) * 4, 256);
    L->total = o;
}

struct ScratchLayout { int64_t tile_count, tile_fill, sub_start, sort_parts, sort_items, rect, keys, block_tot, pair_grad, pair_valid, total; };
static inline void scratch_layout(int P, int H, int W, int64_t cap, int fwd_only, ScratchLayout *L) {
    const int64_t tiles = (int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    int64_t o = 0;
    L->tile_count = o;  o = align_up(o + tiles * 4 * L2D_SLICES, 256);
    L->tile_fill = o;   o = align_up(o + tiles * 4 * L2D_SLICES, 256);
    L->sub_start = o;   o = align_up(o + tiles * 4 * L2D_SLICES, 256);
    L->sort_parts = o;  o = align_up(o + (tiles + 1) * 4, 256);
    L->sort_items = o;  o = align_up(o + tiles * 8, 256);
    const int64_t fwd0 = o;
    L->rect = o;        o = align_up(o + (int64_t)P * 16, 256);
    L->keys = o;        o = align_up(o + cap * 8, 256);
    L->block_tot = o;   o = align_up(o + (((int64_t)P + 255) / 256 + 1) * 4, 256);
    const int64_t fwd_end = o;
    L->pair_grad = fwd0;  // backward reuses the forward-only region
    L->pair_valid = align_up(fwd0 + cap * GRAD_F * 4, 256);
    const int64_t bwd_end = align_up(L->pair_valid + cap + 256, 256);
    L->total = (fwd_only || fwd_end > bwd_end) ? fwd_end : bwd_end;
}

// ---- launchers (one per .hip translation unit) -------------------------------------------------
int launch_preprocess_fwd(const ViewDev &v, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          const float *rotations, const float *transmat_precomp, StateView st,
                          ScratchView sc, int32_t *radii, hipStream_t s);
int launch_preprocess_fwd_views(const ViewDev &v, int n, const ViewDev *views, const float *means3D, const float *shs,
                                const float *colors_precomp, const float *opacities, const float *scales,
                                const float *rotations, const float *transmat_precomp, const StateView *st,
                                const ScratchView *sc, int32_t *const *radii, hipStream_t s);
int launch_binning(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s, const ViewBatch *vb = nullptr);
// the lists of a subset of the surfels a coarse call has binned, by filtering its lists (binning.hip); cst = view 0 of the coarse state
int launch_binning_subset(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s, const ViewBatch *vb, StateView cst,
                          long long coarse_stride, const int32_t *inv);
int launch_composite_fwd(const ViewDev &v, StateView st, ScratchView sc, float *out_color, float *out_allmap,
                         hipStream_t s, const ViewBatch *vb = nullptr);
// re-orders the backward's work items by what they cost the forward AND zero-fills [zero_base, zero_base + zero_bytes) (per view at
// the scratch stride): the one launch in front of composite_bwd
int launch_bwd_order(const ViewDev &v, StateView st, ScratchView sc, hipStream_t s, const ViewBatch *vb, void *zero_base, int64_t zero_bytes);
int launch_composite_bwd(const ViewDev &v, StateView st, ScratchView sc, const float *dL_dcolor,
                         const float *dL_dallmap, hipStream_t s, const ViewBatch *vb = nullptr);
int launch_preprocess_bwd(const ViewDev &v, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *scales, const float *rotations,
                          const float *transmat_precomp, const int32_t *radii, StateView st,
                          ScratchView sc, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs,
                          float *dL_dcolors, float *dL_dopacities, float *dL_dscales,
                          float *dL_drotations, float *dL_dtransmat, hipStream_t s);
int launch_preprocess_bwd_views(const ViewDev &v, int n, const ViewDev *views, int accumulate, const float *means3D,
                                const float *shs, const float *colors_precomp, const float *scales, const float *rotations,
                                const float *transmat_precomp, const int32_t *const *radii, const StateView *st,
                                const ScratchView *sc, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs,
                                float *dL_dcolors, float *dL_dopacities, float *dL_dscales, float *dL_drotations,
                                float *dL_dtransmat, hipStream_t s);
int launch_selftest_butterfly(const float *in, float *out, hipStream_t s);
int launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present,
                        hipStream_t s);

void l2d_set_hip_error(hipError_t e);

// optional event bracketing of a launch (see lara2dgs_profile_enable)
struct L2dProfScope {
    int slot;
    hipStream_t s;
    L2dProfScope(const char *name, hipStream_t stream);
    ~L2dProfScope();
};
#define L2D_PROF(name, stream) L2dProfScope prof_scope__(name, stream)
#define L2D_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) {                            \
            l2d_set_hip_error(e__);                         \
            return LARA2DGS_E_LAUNCH;                       \
        }                                                   \
    } while (0)
