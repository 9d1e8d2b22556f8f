// abi.hip -- extern "C" entry points of liblara2dgs.so (see include/lara2dgs.h).
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local int g_last_hip_error = 0;
void l2d_set_hip_error(hipError_t e) { g_last_hip_error = (int)e; }

// ---- optional per-kernel timing --------------------------------------------------------------
#include <atomic>
#include <mutex>
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
    state_layout(v.P, v.H, v.W, v.cap, &L);
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
    return s;
}

ScratchView carve_scratch(const ViewDev &v, void *scratch, ScratchLayout &L) {
    scratch_layout(v.P, v.H, v.W, v.cap, &L);
    char *b = (char *)scratch;
    ScratchView s;
    s.tile_count = (uint32_t *)(b + L.tile_count);
    s.tile_fill = (uint32_t *)(b + L.tile_fill);
    s.sub_start = (uint32_t *)(b + L.sub_start);
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

int64_t lara2dgs_state_bytes(int32_t P, int32_t H, int32_t W, int64_t capacity) {
    if (P < 0 || H <= 0 || W <= 0 || capacity < 0) return LARA2DGS_E_INVALID;
    lara2dgs_state_layout L;
    state_layout(P, H, W, capacity, &L);
    return L.total;
}

int64_t lara2dgs_scratch_bytes(int32_t P, int32_t H, int32_t W, int64_t capacity) {
    if (P < 0 || H <= 0 || W <= 0 || capacity < 0) return LARA2DGS_E_INVALID;
    ScratchLayout L;
    scratch_layout(P, H, W, capacity, &L);
    return L.total;
}

int lara2dgs_get_state_layout(int32_t P, int32_t H, int32_t W, int64_t capacity,
                              lara2dgs_state_layout *out) {
    if (!out || P < 0 || H <= 0 || W <= 0 || capacity < 0) return LARA2DGS_E_INVALID;
    state_layout(P, H, W, capacity, out);
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
    return launch_composite_fwd(v, st, out_color, out_allmap, s);
}

int lara2dgs_backward(const lara2dgs_view *view, const float *means3D, const float *shs,
                      const float *colors_precomp, const float *scales, const float *rotations,
                      const float *transmat_precomp, const int32_t *radii, const float *dL_dcolor,
                      const float *dL_dallmap, const void *state, void *scratch,
                      float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs, float *dL_dcolors,
                      float *dL_dopacities, float *dL_dscales, float *dL_drotations,
                      float *dL_dtransmat, void *stream) {
    ViewDev v;
    if (!make_view(view, v)) return LARA2DGS_E_INVALID;
    if (!dL_dcolor || !dL_dallmap || !state || !scratch) return LARA2DGS_E_INVALID;
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
    StateView st = carve_state(v, const_cast<void *>(state));
    ScratchLayout SL;
    ScratchView sc = carve_scratch(v, scratch, SL);
    hipError_t e = hipMemsetAsync(sc.pair_valid, 0, (size_t)(SL.total - SL.pair_valid), s);
    if (e != hipSuccess) { l2d_set_hip_error(e); return LARA2DGS_E_LAUNCH; }
    int rc = launch_composite_bwd(v, st, sc, dL_dcolor, dL_dallmap, s);
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
