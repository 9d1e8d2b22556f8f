/*
 * lara2dgs.h -- C ABI of the MI355X-native 2D-Gaussian-surfel rasteriser (liblara2dgs.so).
 *
 * This is the drop-in boundary for the one hot path of autonomousvision/LaRa that is native code:
 * the `diff_surfel_rasterization._C` extension that `lightning/renderer_2dgs.py:7-10` imports and
 * `renderer_2dgs.py:209-218` calls through `GaussianRasterizer`.  The reference's native module
 * (pybind, torch::Tensor signatures; sources absent from the snapshot, .gitmodules:1-3) exposes
 * three entry points; each function below names the one it replaces.  Signatures are plain
 * pointers and sizes -- no torch types -- so the library is bindable from ctypes / cffi / any FFI
 * (see INTEGRATION.md for the binding a LaRa maintainer adds).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated otherwise; all float data is fp32,
 *     contiguous, 16-byte aligned (a torch allocation is);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the calls return
 *     without synchronising (the reference blocks on a D2H copy of `num_rendered` once per view,
 *     between its scan and its duplicate-with-keys; here the caller may have the same number
 *     WRITTEN TO HOST MEMORY by the scan kernel -- `counts_out` -- and read it while the rest of the
 *     forward is still running);
 *   - the caller owns every allocation: outputs, the `state` buffer that forward hands to
 *     backward (the reference's geomBuffer/binningBuffer/imgBuffer), and a transient `scratch`;
 *   - the library keeps no global state and no settings (bar the optional profiling log) and is re-entrant across
 *     streams;
 *   - functions return 0 on success or a negative LARA2DGS_E_* code; nothing throws.
 *   - binning capacity: the number of (tile, surfel) pairs D is data dependent and only known on the
 *     device.  The caller passes `capacity`; if a view needs more, the kernels write the real D into
 *     header[0], raise the `overflow` word header[1] AND poison the outputs with NaN (loud in the data).
 *     Nothing else was harmed: the caller repeats the SAME call with buffers sized for the reported D.
 *     The Python operator does exactly that BEFORE IT RETURNS -- it sizes a call from the counts recent
 *     calls of the same size reported (2 x their maximum), waits for `counts_out` (written by the scan,
 *     a fifth of the way into the forward) and repeats a call that did not fit, so that no consumer ever
 *     sees the poisoned outputs; the reference resizes its buffers after a blocking read of num_rendered
 *     and can never fail on D either (lara_amd/rasterizer.py "workspace policy").
 *   - forward-only calls (`forward_only` = 1; the reference's inference callers evaluation.py:129 and
 *     tools/meshExtractor.py:85 run under no_grad and never call the backward): the forward leaves out
 *     everything it keeps for the backward -- candidate masks, segment checkpoints, per-pixel finals,
 *     contributor counts, the work-item lists -- and `state` shrinks to the sorted lists + the surfel
 *     records (sizes from the same queries with forward_only = 1).  Images, radii and the integer
 *     surface (point_list, ranges) are the training-mode forward's, bit for bit.
 */
#ifndef LARA2DGS_H
#define LARA2DGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LARA2DGS_ABI_VERSION 10

#define LARA2DGS_OK 0
#define LARA2DGS_E_INVALID (-1)   /* bad argument (null pointer, negative size, sh_degree > 3 ...) */
#define LARA2DGS_E_LAUNCH (-2)    /* a HIP launch / API call failed; see lara2dgs_last_hip_error() */
#define LARA2DGS_E_UNSUPPORTED (-3)

/* The 12 fields of GaussianRasterizationSettings (renderer_2dgs.py:124-137) plus sizes.
 * bg / viewmatrix / projmatrix / campos stay DEVICE tensors exactly as the reference passes them
 * (viewmatrix = w2c^T, projmatrix = w2c^T P^T, campos = -c2w[:3,3]; lightning/utils.py:39-48). */
typedef struct lara2dgs_view {
    int32_t P;              /* number of surfels */
    int32_t sh_degree;      /* active degree 0..3 */
    int32_t sh_coeffs;      /* coefficients stored per surfel = shs.shape[1] (0 with colors_precomp) */
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t prefiltered;    /* bit 0: as the reference passes it (unused by LaRa, ignored); bit 1 (opt-in, not in the
                             * reference): cull surfels whose opacity is below 1/255 -- alpha = min(0.99, opacity * G)
                             * can then never pass the 1/255 test, so images and gradients are unchanged while such
                             * surfels leave the binning, sort and composite (their radii read 0) */
    int32_t debug;
    int32_t forward_only;   /* 1: no backward will follow (inference): see "forward-only calls" above */
    int64_t capacity;       /* max (tile, surfel) pairs the state/scratch buffers were sized for */
    const float *bg;         /* [3]  */
    const float *viewmatrix; /* [16] */
    const float *projmatrix; /* [16] */
    const float *campos;     /* [3]  */
    uint32_t *counts_out;    /* optional (NULL = off).  uint32[4] in HOST memory the device can write (hipHostMalloc /
                              * a pinned torch tensor): the scan kernel stores [0] = num_rendered D, [1] = overflow
                              * (D > capacity), [2] = longest tile list, then -- after a system-scope release -- [3] = 1.
                              * The caller clears [3] before the call and may spin on it: D is known a fifth of the way
                              * into the forward, while the scatter / sort / composite kernels are still running.  This
                              * replaces the reference's blocking D2H copy of num_rendered.  In a multi-view call view i
                              * must pass views[0].counts_out + 4 * i (or every view NULL). */
} lara2dgs_view;

/* Byte offsets of the sections of the `state` buffer (for tests, debugging and tooling; the
 * integer sections are the bit-exact parity surface: SURVEY.md section 8a R2-R7). */
typedef struct lara2dgs_state_layout {
    int64_t header;      /* uint32[16]: [0]=num_rendered, [1]=overflow flag, [2]=max tile list length,
                          * [3]=number of full segments (interior boundaries) */
    int64_t geom;        /* float[P][20]: Tu(3) Tv(3) Tw(3) xy(2) opacity normal(3) depth rgb(3) clamp-bits */
    int64_t cullbox;     /* float[P][4]: min x, max x, min y, max y of the pixels a surfel can reach with
                          * alpha >= 1/255 (conservative; lets the composite skip 8x8 quadrants) */
    int64_t point_list;  /* uint32[capacity]: surfel ids, per tile, sorted by (depth bits, id) */
    int64_t ranges;      /* uint32[tiles][2]: [start, end) per 16x16 tile, (0,0) when empty */
    int64_t tile_order;  /* uint32[tiles]: tile ids sorted by list length, longest first: the composite
                          * kernels' workgroup -> tile map (load balance across the 256 CUs) */
    int64_t pair_base;   /* uint32[P+1]: first (tile, surfel) pair of each surfel, surfel-major order */
    int64_t pair_pos;    /* uint32[capacity]: position in point_list -> the (tile, surfel) pair's index in surfel-major
                          * numbering (pair_base[id] + tile offset); the backward writes its per-pair gradient rows
                          * there, so that a surfel's rows are contiguous and summed without atomics */
    int64_t final_T;     /* float[10][H][W]: end-of-walk T, M1, M2, colour(3), depth, normal(3) sums */
    int64_t n_contrib;   /* uint32[2][H][W]: last contributor, median contributor */
    int64_t seg_base;    /* uint32[tiles+1]: exclusive scan of floor((len-1)/512) = interior boundaries when a
                          * tile's list is cut into 512-entry segments (the backward's unit of work) */
    int64_t seg_cnt;     /* uint32[tiles]: floor((len-1)/512) per tile, the same numbers un-scanned */
    int64_t bwd_order;   /* uint32[tiles]: tile ids by length of the last (partial) segment, longest first */
    int64_t bwd_items;   /* uint32[capacity/512+1 + tiles][2]: (tile, segment) of every full segment; after the
                          * backward's ordering pass (header[22] = 1): of every work item, last segments included, dearest first */
    int64_t ckpt;        /* float[capacity/512+1][10][256]: the ten per-pixel running sums as the forward walk
                          * crosses a segment boundary; lets segments of one tile run on different CUs */
    int64_t pair_mask;   /* uint64[capacity]: per list position, the forward's candidate mask of the entry over the tile's
                          * 8x8 grid of 2x2 pixel blocks (bit gy*8+gx); the backward reads it instead of scan-converting
                          * the surfel's footprint a second time */
    int64_t tile_maxc;   /* uint32[tiles]: the largest last-contributor position over the tile's pixels, written by the forward
                          * composite: a backward work item (tile, segment) beyond it exits on its first load */
    int64_t seg_cost;    /* uint32[capacity/512+1 + tiles]: what each backward work item cost the FORWARD (wave-trips of its
                          * 512-entry round): [seg_base[tile] + s] for the full segments, [capacity/512+1 + tile] for a tile's
                          * last one; the forward re-orders bwd_items / bwd_order by it, dearest first */
    int64_t total;
} lara2dgs_state_layout;

int lara2dgs_abi_version(void);
const char *lara2dgs_error_string(int code);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int lara2dgs_last_hip_error(void);

/* Sizes of the two caller-owned buffers.  Replaces the resize-callback scheme of the reference's
 * `rasterize_gaussians` (geomBuffer / binningBuffer / imgBuffer).  forward_only = the view's flag: the sections only a
 * backward reads then have size 0 (their offsets equal the next section's) and the scratch buffer ends with the forward's
 * own arrays. */
int64_t lara2dgs_state_bytes(int32_t P, int32_t H, int32_t W, int64_t capacity, int32_t forward_only);
int64_t lara2dgs_scratch_bytes(int32_t P, int32_t H, int32_t W, int64_t capacity, int32_t forward_only);
int lara2dgs_get_state_layout(int32_t P, int32_t H, int32_t W, int64_t capacity, int32_t forward_only,
                              lara2dgs_state_layout *out);

/* Replaces `_C.rasterize_gaussians(bg, means3D, colors_precomp, opacities, scales, rotations,
 * scale_modifier, transMat_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, sh, degree,
 * campos, prefiltered, debug)` as called from GaussianRasterizer.forward
 * (renderer_2dgs.py:209-218).
 *   means3D [P,3]; exactly one of shs [P,M,3] / colors_precomp [P,3]; opacities [P];
 *   either scales [P,2] + rotations [P,4] (w,x,y,z) or transmat_precomp [P,9]; unused = NULL.
 *   out_color [3,H,W]; out_allmap [7,H,W] (depth-sum, alpha, normal xyz, median depth,
 *   distortion: renderer_2dgs.py:226-242); out_radii int32 [P]. */
int lara2dgs_forward(const lara2dgs_view *view, const float *means3D, const float *shs,
                     const float *colors_precomp, const float *opacities, const float *scales,
                     const float *rotations, const float *transmat_precomp, float *out_color,
                     float *out_allmap, int32_t *out_radii, void *state, void *scratch,
                     void *stream);

/* Replaces `_C.rasterize_gaussians_backward(...)`.  `state` is the buffer a forward with forward_only = 0 filled (a view
 * with forward_only = 1 is LARA2DGS_E_INVALID here); the backward WRITES to it (it
 * re-orders the work-item list `bwd_items` in place and sets header[22]): one backward at a time per state buffer.
 * dL_dallmap may be NULL = no gradient on any of the seven maps (LaRa's fine pass, lightning/loss.py:35-47: the loss reads its
 * image only): the compositing backward then runs its colour-only form -- same gradients as seven planes of zeros, bit for bit.
 * Gradient outputs (any may be NULL when the corresponding input was NULL):
 *   dL_dmeans3D [P,3], dL_dmeans2D [P,3], dL_dshs [P,M,3], dL_dcolors [P,3], dL_dopacities [P],
 *   dL_dscales [P,2], dL_drotations [P,4], dL_dtransmat [P,9].  They are fully overwritten. */
int lara2dgs_backward(const lara2dgs_view *view, const float *means3D, const float *shs,
                      const float *colors_precomp, const float *scales, const float *rotations,
                      const float *transmat_precomp, const int32_t *radii, const float *dL_dcolor,
                      const float *dL_dallmap, void *state, void *scratch,
                      float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs, float *dL_dcolors,
                      float *dL_dopacities, float *dL_dscales, float *dL_drotations,
                      float *dL_dtransmat, void *stream);


/* ---- multi-view calls (not in the reference, whose Python loop issues one call per view:
 * lightning/network.py:486-497, :516-525) ------------------------------------------------------------------------
 * One call rasterises the SAME surfels from n_views cameras.  `views` is an array of n_views records that agree in
 * everything but the four camera pointers (and bg).  Outputs are stacked: out_color [n,3,H,W], out_allmap [n,7,H,W],
 * out_radii [n,P]; view i's saved state lives at state + i * state_stride (state_stride >= lara2dgs_state_bytes,
 * 256-byte multiple) -- per-camera state carved from one allocation; `scratch` holds n_views transient buffers of
 * scratch_stride bytes each (>= lara2dgs_scratch_bytes).  Every kernel of the call is ONE launch over the cameras on `stream`
 * (workgroup z index = view, chunks of 8): the surfels' inputs are read from HBM once for the n cameras, and one view's tail
 * of long tile lists is filled by the next view's workgroups.  Per-view results are the per-view call's, bit for bit. */
int lara2dgs_forward_views(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                           const float *shs, const float *colors_precomp, const float *opacities,
                           const float *scales, const float *rotations, const float *transmat_precomp,
                           float *out_color, float *out_allmap, int32_t *out_radii, void *state,
                           int64_t state_stride, void *scratch, int64_t scratch_stride, void *stream);

/* The same call for a SUBSET of the surfels an earlier multi-view call has rendered, from the same cameras at the same image size and
 * with the same geometry (means, scales, rotations, opacities: the subset's rows are copies of the earlier call's) -- LaRa's fine
 * pass (lightning/network.py:502-525: `x[mask]` of the coarse pass's Gaussians with refined SH coefficients).  The subset's
 * per-tile lists are then the earlier call's lists with the dropped surfels taken out and the ids renumbered -- the sort key is
 * (depth bits, id) and the subset's numbering is monotone -- so this entry point FILTERS those lists (stable compaction per tile)
 * instead of scattering and sorting again.  Everything it leaves in `state` -- records, point_list, ranges, the pair map -- equals
 * lara2dgs_forward_views' on the same inputs bit for bit; so do the images.
 *   coarse_state / coarse_state_stride / coarse_capacity / coarse_P / coarse_forward_only: the earlier call's state buffer (still
 *       alive: a training step keeps it for the backward) and the arguments it was laid out with; a subset call that keeps state for a
 *       backward (views[0].forward_only == 0) needs an earlier call that did too (the pair map is derived from its pair map);
 *   inv: int32 [coarse_P] on the device: the subset's row of every earlier surfel, -1 for the dropped ones; rows ascending.
 * views[i] must be the earlier call's view i (same cameras); P = the subset's surfel count. */
typedef struct lara2dgs_subset {
    const void *coarse_state;
    int64_t coarse_state_stride;
    int64_t coarse_capacity;
    int32_t coarse_P;
    int32_t coarse_forward_only;
    const int32_t *inv;
} lara2dgs_subset;
int lara2dgs_forward_views_subset(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                                  const float *shs, const float *colors_precomp, const float *opacities,
                                  const float *scales, const float *rotations, const float *transmat_precomp,
                                  float *out_color, float *out_allmap, int32_t *out_radii, void *state,
                                  int64_t state_stride, void *scratch, int64_t scratch_stride,
                                  const lara2dgs_subset *subset, void *stream);

/* Offsets (in floats) of the gradient arrays inside one flat gradient buffer; -1 = absent. */
typedef struct lara2dgs_grad_layout {
    int64_t means3D, means2D, shs, colors, opacities, scales, rotations, transmat, total;
} lara2dgs_grad_layout;
int lara2dgs_get_grad_layout(int32_t P, int32_t sh_coeffs, int32_t has_shs, int32_t has_colors,
                             int32_t has_scale_rot, int32_t has_transmat, lara2dgs_grad_layout *out);

/* Backward of lara2dgs_forward_views: dL_dcolor [n,3,H,W], dL_dallmap [n,7,H,W] or NULL (= zero, as above), radii [n,P].  grad_out
 * ([layout.total] floats; every gradient array in it is fully overwritten) receives the gradients summed over the
 * views in view order -- the sum over a scene's views that autograd otherwise forms with n-1 accumulation kernels per
 * input, and bit-reproducible: each view's per-pair gradient rows stay in its own scratch buffer and ONE per-surfel launch
 * folds the n views' contributions in registers, in view order.  Writes to `state` like lara2dgs_backward. */
int lara2dgs_backward_views(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                            const float *shs, const float *colors_precomp, const float *scales,
                            const float *rotations, const float *transmat_precomp, const int32_t *radii,
                            const float *dL_dcolor, const float *dL_dallmap, void *state,
                            int64_t state_stride, void *scratch, int64_t scratch_stride,
                            float *grad_out, void *stream);

/* Replaces `_C.mark_visible(means3D, viewmatrix, projmatrix)` (GaussianRasterizer.markVisible).
 * present: uint8 [P]. */
int lara2dgs_mark_visible(int32_t P, const float *means3D, const float *viewmatrix,
                          const float *projmatrix, uint8_t *present, void *stream);

/* Optional per-kernel timing (bench.py's roofline leg; not part of the reference surface).  THE ONE process-wide switch of this
 * library -- it changes no result, it is off by default, and tests/test_abi_cpu.py::test_library_exports_no_setters names it as the
 * single exported symbol that reads like a switch.
 * When enabled (process-wide), every kernel the library launches is bracketed by HIP
 * events recorded on the launch stream.  lara2dgs_profile_collect synchronises those events,
 * writes up to `max_entries` records (kernel name -> `names`, NUL-separated, at most `names_len`
 * bytes; duration in milliseconds -> `ms`), clears the log and returns the number written. */
int lara2dgs_profile_enable(int on);
/* Device self-tests of building blocks whose correctness rests on gfx950 lane semantics.
 * which = 0: DPP quad reduce-scatter of the backward composite; in = float[64][22] (lane major),
 * out = float[16][22]: out[q][k] = sum over the 4 lanes of quad q of in[.][k]. */
int lara2dgs_selftest(int which, const float *in, float *out, void *stream);
int lara2dgs_profile_collect(char *names, int names_len, float *ms, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* LARA2DGS_H */
