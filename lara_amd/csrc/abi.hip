// abi.hip -- extern "C" entry points of liblara2dgs.so (see include/lara2dgs.h).
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local int g_last_hip_error = 0;
void l2d_set_hip_error(hipError_t e) { g_last_hip_error = (int)e; }

// ---- optional per-kernel timing --------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <map>
#include <utility>
#include <vector>
namespace {
// Process-wide (PyTorch runs backward on its autograd thread, not on the caller's).
struct ProfRec { const char *name; hipEvent_t a, b; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_log;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {  // call with g_prof_mu held
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

L2dProfScope::L2dProfScope(const char *name, hipStream_t stream) : slot(-1), s(stream) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, prof_event(), prof_event()};
    (void)hipEventRecord(r.a, s);
    g_prof_log.push_back(r);
    slot = (int)g_prof_log.size() - 1;
}
L2dProfScope::~L2dProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof_log.size()) (void)hipEventRecord(g_prof_log[slot].b, s);
}

namespace {

bool make_view(const lara2dgs_view *view, ViewDev &v) {
    if (!view) return false;
    if (view->P < 0 || view->image_height <= 0 || view->image_width <= 0) return false;
    if (view->sh_degree < 0 || view->sh_degree > 3) return false;
    if (view->capacity < 0 || view->capacity > 0xffffffffll) return false;
    if (!view->bg || !view->viewmatrix || !view->projmatrix || !view->campos) return false;
    v.P = view->P;
    v.deg = view->sh_degree;
    v.M = view->sh_coeffs;
    v.H = view->image_height;
    v.W = view->image_width;
    v.gx = (v.W + TILE - 1) / TILE;
    v.gy = (v.H + TILE - 1) / TILE;
    v.tiles = v.gx * v.gy;
    if (v.gx > 65535 || v.gy > 65535) return false;
    v.scale_modifier = view->scale_modifier;
    v.cull_transparent = (view->prefiltered >> 1) & 1;
    v.cap = (unsigned)view->capacity;
    v.fwd_only = view->forward_only != 0;
    v.counts_out = view->counts_out;
    {
        static const unsigned dbg = getenv("LARA2DGS_DEBUG_FLAGS") ? (unsigned)strtoul(getenv("LARA2DGS_DEBUG_FLAGS"), nullptr, 0) : 0u;
        v.dbg = dbg;
    }
    v.bg = view->bg;
    v.viewmatrix = view->viewmatrix;
    v.projmatrix = view->projmatrix;
    v.campos = view->campos;
    return true;
}

StateView carve_state(const ViewDev &v, void *state) {
    lara2dgs_state_layout L;
    state_layout(v.P, v.H, v.W, v.cap, v.fwd_only, &L);
    char *b = (char *)state;
    StateView s;
    s.header = (uint32_t *)(b + L.header);
    s.geom = (float4 *)(b + L.geom);
    s.cullbox = (float4 *)(b + L.cullbox);
    s.point_list = (uint32_t *)(b + L.point_list);
    s.ranges = (uint2 *)(b + L.ranges);
    s.tile_order = (uint32_t *)(b + L.tile_order);
    s.pair_base = (uint32_t *)(b + L.pair_base);
    s.pair_pos = (uint32_t *)(b + L.pair_pos);
    s.final_T = (float *)(b + L.final_T);
    s.n_contrib = (uint32_t *)(b + L.n_contrib);
    s.seg_base = (uint32_t *)(b + L.seg_base);
    s.seg_cnt = (uint32_t *)(b + L.seg_cnt);
    s.bwd_order = (uint32_t *)(b + L.bwd_order);
    s.bwd_items = (uint2 *)(b + L.bwd_items);
    s.ckpt = (float *)(b + L.ckpt);
    s.pair_mask = (uint2 *)(b + L.pair_mask);
    s.tile_maxc = (uint32_t *)(b + L.tile_maxc);
    s.seg_cost = (uint32_t *)(b + L.seg_cost);
    return s;
}

ScratchView carve_scratch(const ViewDev &v, void *scratch, ScratchLayout &L) {
    scratch_layout(v.P, v.H, v.W, v.cap, v.fwd_only, &L);
    char *b = (char *)scratch;
    ScratchView s;
    s.tile_count = (uint32_t *)(b + L.tile_count);
    s.tile_fill = (uint32_t *)(b + L.tile_fill);
    s.sub_start = (uint32_t *)(b + L.sub_start);
    s.sort_parts = (uint32_t *)(b + L.sort_parts);
    s.sort_items = (uint2 *)(b + L.sort_items);
    s.rect = (uint4 *)(b + L.rect);
    s.keys = (uint64_t *)(b + L.keys);
    s.block_tot = (uint32_t *)(b + L.block_tot);
    s.pair_grad = (float4 *)(b + L.pair_grad);
    s.pair_valid = (uint32_t *)(b + L.pair_valid);
    return s;
}

}  // namespace

extern "C" {

int lara2dgs_abi_version(void) { return LARA2DGS_ABI_VERSION; }

const char *lara2dgs_error_string(int code) {
    switch (code) {
    case LARA2DGS_OK: return "ok";
    case LARA2DGS_E_INVALID: return "invalid argument";
    case LARA2DGS_E_LAUNCH: return "HIP launch/API failure (see lara2dgs_last_hip_error)";
    case LARA2DGS_E_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
    }
}

int lara2dgs_last_hip_error(void) { return g_last_hip_error; }

int64_t lara2dgs_state_bytes(int32_t P, int32_t H, int32_t W, int64_t capacity, int32_t forward_only) {
    if (P < 0 || H <= 0 || W <= 0 || capacity < 0) return LARA2DGS_E_INVALID;
    lara2dgs_state_layout L;
    state_layout(P, H, W, capacity, forward_only != 0, &L);
    return L.total;
}

int64_t lara2dgs_scratch_bytes(int32_t P, int32_t H, int32_t W, int64_t capacity, int32_t forward_only) {
    if (P < 0 || H <= 0 || W <= 0 || capacity < 0) return LARA2DGS_E_INVALID;
    ScratchLayout L;
    scratch_layout(P, H, W, capacity, forward_only != 0, &L);
    return L.total;
}

int lara2dgs_get_state_layout(int32_t P, int32_t H, int32_t W, int64_t capacity, int32_t forward_only,
                              lara2dgs_state_layout *out) {
    if (!out || P < 0 || H <= 0 || W <= 0 || capacity < 0) return LARA2DGS_E_INVALID;
    state_layout(P, H, W, capacity, forward_only != 0, out);
    return LARA2DGS_OK;
}

int lara2dgs_forward(const lara2dgs_view *view, const float *means3D, const float *shs,
                     const float *colors_precomp, const float *opacities, const float *scales,
                     const float *rotations, const float *transmat_precomp, float *out_color,
                     float *out_allmap, int32_t *out_radii, void *state, void *scratch,
                     void *stream) {
    ViewDev v;
    if (!make_view(view, v)) return LARA2DGS_E_INVALID;
    if (!out_color || !out_allmap || !state || !scratch) return LARA2DGS_E_INVALID;
    if (v.P > 0) {
        if (!means3D || !opacities || !out_radii) return LARA2DGS_E_INVALID;
        if ((shs == nullptr) == (colors_precomp == nullptr)) return LARA2DGS_E_INVALID;
        const bool has_sr = scales && rotations;
        if (has_sr == (transmat_precomp != nullptr)) return LARA2DGS_E_INVALID;
        if (shs && v.M < (v.deg + 1) * (v.deg + 1)) return LARA2DGS_E_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    StateView st = carve_state(v, state);
    ScratchLayout SL;
    ScratchView sc = carve_scratch(v, scratch, SL);
    // tile_count + tile_fill start at zero (the header is initialised by tile_scan, the first kernel to use it)
    hipError_t e = hipMemsetAsync(sc.tile_count, 0, (size_t)(SL.sub_start - SL.tile_count), s);
    if (e != hipSuccess) { l2d_set_hip_error(e); return LARA2DGS_E_LAUNCH; }
    int rc = launch_preprocess_fwd(v, means3D, shs, colors_precomp, opacities, scales, rotations,
                                   transmat_precomp, st, sc, out_radii, s);
    if (rc) return rc;
    rc = launch_binning(v, st, sc, s);
    if (rc) return rc;
    return launch_composite_fwd(v, st, sc, out_color, out_allmap, s);
}

int lara2dgs_backward(const lara2dgs_view *view, const float *means3D, const float *shs,
                      const float *colors_precomp, const float *scales, const float *rotations,
                      const float *transmat_precomp, const int32_t *radii, const float *dL_dcolor,
                      const float *dL_dallmap, void *state, void *scratch,
                      float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs, float *dL_dcolors,
                      float *dL_dopacities, float *dL_dscales, float *dL_drotations,
                      float *dL_dtransmat, void *stream) {
    ViewDev v;
    if (!make_view(view, v)) return LARA2DGS_E_INVALID;
    if (!dL_dcolor || !state || !scratch) return LARA2DGS_E_INVALID;      // (dL_dallmap NULL = zero: the colour-only backward)
    if (v.fwd_only) return LARA2DGS_E_INVALID;      // a forward-only call kept nothing for a backward
    if (v.P == 0) return LARA2DGS_OK;
    if (!means3D || !radii || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacities) return LARA2DGS_E_INVALID;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return LARA2DGS_E_INVALID;
    if (shs && !dL_dshs) return LARA2DGS_E_INVALID;
    if (colors_precomp && !dL_dcolors) return LARA2DGS_E_INVALID;
    const bool has_sr = scales && rotations;
    if (has_sr == (transmat_precomp != nullptr)) return LARA2DGS_E_INVALID;
    if (has_sr && (!dL_dscales || !dL_drotations)) return LARA2DGS_E_INVALID;
    if (transmat_precomp && !dL_dtransmat) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    StateView st = carve_state(v, state);
    ScratchLayout SL;
    ScratchView sc = carve_scratch(v, scratch, SL);
    // one launch: the validity bitmap zeroed + the work items ordered dearest first (by what they cost the forward)
    int rc = launch_bwd_order(v, st, sc, s, nullptr, sc.pair_valid, SL.total - SL.pair_valid);
    if (rc) return rc;
    rc = launch_composite_bwd(v, st, sc, dL_dcolor, dL_dallmap, s);
    if (rc) return rc;
    return launch_preprocess_bwd(v, means3D, shs, colors_precomp, scales, rotations, transmat_precomp,
                                 radii, st, sc, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors,
                                 dL_dopacities, dL_dscales, dL_drotations, dL_dtransmat, s);
}

int lara2dgs_mark_visible(int32_t P, const float *means3D, const float *viewmatrix,
                          const float *projmatrix, uint8_t *present, void *stream) {
    (void)projmatrix;  // kept for signature parity with the reference; the test only needs view z
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return LARA2DGS_E_INVALID;
    return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

int lara2dgs_selftest(int which, const float *in, float *out, void *stream) {
    if (which != 0 || !in || !out) return LARA2DGS_E_INVALID;
    return launch_selftest_butterfly(in, out, (hipStream_t)stream);
}

int lara2dgs_profile_enable(int on) {
    g_prof_on.store(on != 0);
    return LARA2DGS_OK;
}

int lara2dgs_profile_collect(char *names, int names_len, float *ms, int max_entries) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0, pos = 0;
    for (size_t i = 0; i < g_prof_log.size(); i++) {
        ProfRec &r = g_prof_log[i];
        if (n < max_entries && names && ms) {
            const int len = (int)strlen(r.name) + 1;
            if (pos + len <= names_len) {
                float t = 0.f;
                (void)hipEventSynchronize(r.b);
                (void)hipEventElapsedTime(&t, r.a, r.b);
                memcpy(names + pos, r.name, len);
                pos += len;
                ms[n++] = t;
            }
        }
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_log.clear();
    return n;
}

}  // extern "C"

// ---- multi-view calls ---------------------------------------------------------------------------------------------
namespace {
#define HIP_TRY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { l2d_set_hip_error(e__); return LARA2DGS_E_LAUNCH; } } while (0)

bool views_agree(int n, const lara2dgs_view *views) {
    for (int i = 1; i < n; i++) {
        const lara2dgs_view &a = views[0], &b = views[i];
        if (a.P != b.P || a.sh_degree != b.sh_degree || a.sh_coeffs != b.sh_coeffs || a.image_height != b.image_height ||
            a.image_width != b.image_width || a.capacity != b.capacity || a.prefiltered != b.prefiltered ||
            a.scale_modifier != b.scale_modifier || a.debug != b.debug || a.forward_only != b.forward_only) return false;   // (the batched kernels run all cameras with views[0]'s scalars)
        if (b.counts_out != (a.counts_out ? a.counts_out + 4 * i : nullptr)) return false;   // (... and view 0's counts pointer + 4 z)
    }
    return true;
}

void grad_layout(int64_t P, int M, bool sh, bool col, bool sr, bool tm, lara2dgs_grad_layout *L) {
    int64_t o = 0;
    auto sec = [&](bool on, int64_t n) { if (!on) return (int64_t)-1; const int64_t at = o; o = (o + n + 3) / 4 * 4; return at; };
    L->means3D = sec(true, P * 3);
    L->means2D = sec(true, P * 3);
    L->shs = sec(sh, P * M * 3);
    L->colors = sec(col, P * 3);
    L->opacities = sec(true, P);
    L->scales = sec(sr, P * 2);
    L->rotations = sec(sr, P * 4);
    L->transmat = sec(tm, P * 9);
    L->total = o;
}

// zero `bytes` (a multiple of 16) at base + v * stride for v < gridDim.y: one launch for all views of a call.  (hipMemset2DAsync
// over the scratch stride takes 70 us per call on this runtime; eight hipMemsetAsync calls 8 launches of 5 us.)
__global__ void __launch_bounds__(256) zero_strided_kernel(char *__restrict__ base, const int64_t stride, const int64_t bytes) {
    uint4 *p = (uint4 *)(base + (int64_t)blockIdx.y * stride);
    const int64_t n = bytes / 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

inline void zero_strided(void *base, int64_t stride, int64_t bytes, int n_views, hipStream_t s) {
    const int64_t n = (bytes + 15) / 16;
    const unsigned gx = (unsigned)(n < 256 * 64 ? (n + 255) / 256 : 64);
    hipLaunchKernelGGL(zero_strided_kernel, dim3(gx ? gx : 1, (unsigned)n_views), dim3(256), 0, s, (char *)base, stride, (bytes + 15) / 16 * 16);
}

bool strides_ok(const lara2dgs_view &v0, int64_t state_stride, int64_t scratch_stride) {
    return state_stride % 256 == 0 && scratch_stride % 256 == 0 &&
           state_stride >= lara2dgs_state_bytes(v0.P, v0.image_height, v0.image_width, v0.capacity, v0.forward_only) &&
           scratch_stride >= lara2dgs_scratch_bytes(v0.P, v0.image_height, v0.image_width, v0.capacity, v0.forward_only);
}
}  // namespace

extern "C" {

int lara2dgs_get_grad_layout(int32_t P, int32_t sh_coeffs, int32_t has_shs, int32_t has_colors,
                             int32_t has_scale_rot, int32_t has_transmat, lara2dgs_grad_layout *out) {
    if (!out || P < 0 || sh_coeffs < 0) return LARA2DGS_E_INVALID;
    grad_layout(P, sh_coeffs, has_shs != 0, has_colors != 0, has_scale_rot != 0, has_transmat != 0, out);
    return LARA2DGS_OK;
}

// Every kernel of a multi-view call is ONE launch over the cameras on the caller's stream (workgroup z index = view; chunks of
// L2D_MAX_VIEWS): the surfels' inputs come from HBM once for the n cameras, and one view's tail of long tile lists is filled by
// the next view's workgroups.  (Rounds 2-4 also kept a per-view path dealt to side streams -- "lanes" -- with two process-wide
// setters; it measured slower in every configuration once the kernels were batched and is gone: the library keeps no settings.)
static int forward_views_impl(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                           const float *shs, const float *colors_precomp, const float *opacities,
                           const float *scales, const float *rotations, const float *transmat_precomp,
                           float *out_color, float *out_allmap, int32_t *out_radii, void *state,
                           int64_t state_stride, void *scratch, int64_t scratch_stride, void *stream,
                           const lara2dgs_subset *subset) {
    if (n_views <= 0 || !views || !state || !scratch) return LARA2DGS_E_INVALID;
    if (subset) {
        // the subset call's lists come out of the coarse call's: same cameras and image, a coarse state that holds what is read
        if (!subset->coarse_state || !subset->inv || subset->coarse_P < views[0].P || subset->coarse_capacity < 0 ||
            subset->coarse_capacity > 0xffffffffll || subset->coarse_state_stride % 256) return LARA2DGS_E_INVALID;
        if (!views[0].forward_only && subset->coarse_forward_only) return LARA2DGS_E_INVALID;   // (the pair map is derived from the coarse one)
        if (subset->coarse_state_stride < lara2dgs_state_bytes(subset->coarse_P, views[0].image_height, views[0].image_width,
                                                                subset->coarse_capacity, subset->coarse_forward_only)) return LARA2DGS_E_INVALID;
    }
    if (!views_agree(n_views, views)) return LARA2DGS_E_INVALID;
    const lara2dgs_view &v0 = views[0];
    if (!strides_ok(v0, state_stride, scratch_stride)) return LARA2DGS_E_INVALID;
    hipStream_t caller = (hipStream_t)stream;
    const int64_t HW = (int64_t)v0.image_height * v0.image_width;
    if (v0.P == 0) {    // nothing to bin: every view is its background (the per-view entry point knows how)
        int rc = LARA2DGS_OK;
        for (int i = 0; i < n_views && rc == LARA2DGS_OK; i++)
            rc = lara2dgs_forward(&views[i], means3D, shs, colors_precomp, opacities, scales, rotations, transmat_precomp,
                                  out_color + i * 3 * HW, out_allmap + i * 7 * HW, out_radii,
                                  (char *)state + i * state_stride, (char *)scratch + i * scratch_stride, stream);
        return rc;
    }
    if (!out_color || !out_allmap || !means3D || !opacities || !out_radii) return LARA2DGS_E_INVALID;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return LARA2DGS_E_INVALID;
    if ((scales && rotations) == (transmat_precomp != nullptr)) return LARA2DGS_E_INVALID;
    std::vector<ViewDev> vd(n_views);
    std::vector<StateView> st(n_views);
    std::vector<ScratchView> sc(n_views);
    std::vector<int32_t *> rad(n_views);
    for (int i = 0; i < n_views; i++) {
        if (!make_view(&views[i], vd[i])) return LARA2DGS_E_INVALID;
        if (shs && vd[i].M < (vd[i].deg + 1) * (vd[i].deg + 1)) return LARA2DGS_E_INVALID;
        st[i] = carve_state(vd[i], (char *)state + i * state_stride);
        ScratchLayout SL;
        sc[i] = carve_scratch(vd[i], (char *)scratch + i * scratch_stride, SL);
        rad[i] = out_radii + (int64_t)i * v0.P;
        if (i == 0)   // the views agree in (P, H, W, capacity): one layout, one strided fill for all of them
            zero_strided(sc[0].tile_count, scratch_stride, SL.sub_start - SL.tile_count, n_views, caller);
    }
    int rc = LARA2DGS_OK;
    for (int i0 = 0; i0 < n_views && rc == LARA2DGS_OK; i0 += L2D_MAX_VIEWS) {
        ViewBatch vb{};
        vb.n = n_views - i0 < L2D_MAX_VIEWS ? n_views - i0 : L2D_MAX_VIEWS;
        vb.state_stride = state_stride; vb.scratch_stride = scratch_stride;
        for (int k = 0; k < vb.n; k++) vb.bg[k] = vd[i0 + k].bg;
        rc = launch_preprocess_fwd_views(vd[i0], vb.n, &vd[i0], means3D, shs, colors_precomp, opacities, scales, rotations,
                                         transmat_precomp, &st[i0], &sc[i0], &rad[i0], caller);
        if (rc == LARA2DGS_OK && !subset) rc = launch_binning(vd[i0], st[i0], sc[i0], caller, &vb);
        if (rc == LARA2DGS_OK && subset) {
            ViewDev cv = vd[i0];      // the coarse call's view i0: its surfel count, capacity and mode decide its state's layout
            cv.P = subset->coarse_P; cv.cap = (unsigned)subset->coarse_capacity; cv.fwd_only = subset->coarse_forward_only != 0;
            const StateView cst = carve_state(cv, (char *)const_cast<void *>(subset->coarse_state) + (int64_t)i0 * subset->coarse_state_stride);
            rc = launch_binning_subset(vd[i0], st[i0], sc[i0], caller, &vb, cst, subset->coarse_state_stride, subset->inv);
        }
        if (rc == LARA2DGS_OK)
            rc = launch_composite_fwd(vd[i0], st[i0], sc[i0], out_color + (int64_t)i0 * 3 * HW, out_allmap + (int64_t)i0 * 7 * HW,
                                      caller, &vb);
    }
    return rc;
}

int lara2dgs_forward_views(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                           const float *shs, const float *colors_precomp, const float *opacities,
                           const float *scales, const float *rotations, const float *transmat_precomp,
                           float *out_color, float *out_allmap, int32_t *out_radii, void *state,
                           int64_t state_stride, void *scratch, int64_t scratch_stride, void *stream) {
    return forward_views_impl(n_views, views, means3D, shs, colors_precomp, opacities, scales, rotations, transmat_precomp, out_color,
                              out_allmap, out_radii, state, state_stride, scratch, scratch_stride, stream, nullptr);
}

int lara2dgs_forward_views_subset(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                                  const float *shs, const float *colors_precomp, const float *opacities,
                                  const float *scales, const float *rotations, const float *transmat_precomp,
                                  float *out_color, float *out_allmap, int32_t *out_radii, void *state,
                                  int64_t state_stride, void *scratch, int64_t scratch_stride,
                                  const lara2dgs_subset *subset, void *stream) {
    if (!subset) return LARA2DGS_E_INVALID;
    return forward_views_impl(n_views, views, means3D, shs, colors_precomp, opacities, scales, rotations, transmat_precomp, out_color,
                              out_allmap, out_radii, state, state_stride, scratch, scratch_stride, stream, subset);
}

int lara2dgs_backward_views(int32_t n_views, const lara2dgs_view *views, const float *means3D,
                            const float *shs, const float *colors_precomp, const float *scales,
                            const float *rotations, const float *transmat_precomp, const int32_t *radii,
                            const float *dL_dcolor, const float *dL_dallmap, void *state,
                            int64_t state_stride, void *scratch, int64_t scratch_stride,
                            float *grad_out, void *stream) {
    if (n_views <= 0 || !views || !state || !scratch || !grad_out) return LARA2DGS_E_INVALID;
    if (!views_agree(n_views, views)) return LARA2DGS_E_INVALID;
    const lara2dgs_view &v0 = views[0];
    if (v0.forward_only) return LARA2DGS_E_INVALID;      // a forward-only call kept nothing for a backward
    if (!strides_ok(v0, state_stride, scratch_stride)) return LARA2DGS_E_INVALID;
    if (!dL_dcolor) return LARA2DGS_E_INVALID;      // (dL_dallmap NULL = zero: the colour-only backward)
    const bool has_sh = shs != nullptr, has_col = colors_precomp != nullptr, has_sr = scales && rotations,
               has_tm = transmat_precomp != nullptr;
    lara2dgs_grad_layout G;
    grad_layout(v0.P, v0.sh_coeffs, has_sh, has_col, has_sr, has_tm, &G);
    hipStream_t caller = (hipStream_t)stream;
    if (v0.P == 0 || G.total == 0) return LARA2DGS_OK;
    const int64_t HW = (int64_t)v0.image_height * v0.image_width;
    auto at = [&](float *base, int64_t off) { return off < 0 ? (float *)nullptr : base + off; };
    if (!means3D || !radii) return LARA2DGS_E_INVALID;
    if (has_sh == has_col || has_sr == has_tm) return LARA2DGS_E_INVALID;
    std::vector<ViewDev> vd(n_views);
    std::vector<StateView> st(n_views);
    std::vector<ScratchView> sc(n_views);
    std::vector<const int32_t *> rad(n_views);
    std::vector<ScratchLayout> SL(n_views);
    for (int i = 0; i < n_views; i++) {
        if (!make_view(&views[i], vd[i])) return LARA2DGS_E_INVALID;
        st[i] = carve_state(vd[i], (char *)state + i * state_stride);
        sc[i] = carve_scratch(vd[i], (char *)scratch + i * scratch_stride, SL[i]);
        rad[i] = radii + (int64_t)i * v0.P;
    }
    // Per chunk of views: the ordering of every view's work items (which also zero-fills the validity bitmaps), composite_bwd --
    // each view's gradient rows stay in its own scratch buffer -- and ONE preprocess_bwd launch that walks every surfel through
    // the views in view order and writes the summed gradients: no per-view gradient tensors, no summation pass.
    int rc = LARA2DGS_OK;
    for (int i0 = 0; i0 < n_views && rc == LARA2DGS_OK; i0 += L2D_MAX_VIEWS) {
        ViewBatch vb{};
        vb.n = n_views - i0 < L2D_MAX_VIEWS ? n_views - i0 : L2D_MAX_VIEWS;
        vb.state_stride = state_stride; vb.scratch_stride = scratch_stride;
        for (int k = 0; k < vb.n; k++) vb.bg[k] = vd[i0 + k].bg;
        rc = launch_bwd_order(vd[i0], st[i0], sc[i0], caller, &vb, sc[i0].pair_valid, SL[i0].total - SL[i0].pair_valid);
        if (rc == LARA2DGS_OK)
            rc = launch_composite_bwd(vd[i0], st[i0], sc[i0], dL_dcolor + (int64_t)i0 * 3 * HW, dL_dallmap ? dL_dallmap + (int64_t)i0 * 7 * HW : nullptr, caller, &vb);
        if (rc == LARA2DGS_OK)
            rc = launch_preprocess_bwd_views(vd[i0], vb.n, &vd[i0], i0 > 0, means3D, shs, colors_precomp, scales, rotations,
                                             transmat_precomp, &rad[i0], &st[i0], &sc[i0], at(grad_out, G.means3D),
                                             at(grad_out, G.means2D), at(grad_out, G.shs), at(grad_out, G.colors),
                                             at(grad_out, G.opacities), at(grad_out, G.scales), at(grad_out, G.rotations),
                                             at(grad_out, G.transmat), caller);
    }
    return rc;
}

}  // extern "C"
