/*
 * oracle/surfel_oracle.c -- CPU ORACLE of the 2D-Gaussian-surfel tile rasteriser.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (lara_amd/,
 * diff_surfel_rasterization/) may import, link or call this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and
 * only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED.  The algorithm LaRa calls lives in the third-party module
 * `hbb1/diff-surfel-rasterization` (git submodule of the reference,
 * /root/reference/.gitmodules:1-3, installed by environment.yml:48).  The
 * submodule directory is EMPTY in the reference snapshot and carries no commit
 * pin, and the reference holds no golden vectors or tests for this boundary
 * (SURVEY.md section 4 / section 8c).  This file therefore restates the *published* 2DGS
 * rasteriser algorithm (Huang et al., "2D Gaussian Splatting for Geometrically
 * Accurate Radiance Fields", SIGGRAPH 2024, and the public repository's
 * forward/backward passes) and anchors it on the reference's own call site:
 *   - argument set / return arity: lightning/renderer_2dgs.py:119-139,209-218
 *   - 7-channel `allmap` layout:   lightning/renderer_2dgs.py:226-242
 *   - camera matrix conventions:   lightning/utils.py:5-48
 * plus self-made known-answer tests (tests/test_oracle_kat.py) and an fp64
 * autograd restatement (oracle/autograd_ref.py) that checks the hand-derived
 * backward below.
 *
 * Floating point: everything is IEEE fp32 in a fixed operation order, compiled
 * with -ffp-contract=off, so that the INTEGER results of the pipeline (radii,
 * tile rectangles, tiles_touched, sort keys, sorted lists, tile ranges) are a
 * bit-exact reference for the HIP path (which is compiled the same way for
 * those stages).  Upstream is built by nvcc with fmad contraction on, so no CPU
 * program can be bit-identical to it; this operation order is our canon.
 *
 * Deviations from upstream, all deliberate and listed in DESIGN.md:
 *   - backward preprocess uses the real W,H (upstream re-derives them as
 *     int(focal*tan*2), which can round to W-1),
 *   - backward preprocess honours scale_modifier (upstream hard-codes 1.0;
 *     LaRa always passes 1.0, renderer_2dgs.py:119),
 *   - quaternion normalisation uses 1/sqrt (upstream: rsqrtf, 2 ulp, not
 *     reproducible); LaRa passes unit quaternions (renderer_2dgs.py:114,189),
 *   - "no median" is stored as median_contributor = 0 (upstream stores
 *     (uint)(-1.0f), which saturates to 0 on NVIDIA hardware),
 *   - per-Gaussian gradients are accumulated in double and rounded once
 *     (upstream: fp32 atomics in arbitrary order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16
#define BLOCK_Y 16
#define NEAR_N 0.2f
#define FAR_N 100.0f
#define FILTER_SIZE 0.707106f
#define FILTER_INV_SQUARE 2.0f
#define CUTOFF 3.0f

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    int32_t P;          /* number of surfels */
    int32_t sh_degree;  /* active SH degree 0..3 */
    int32_t sh_coeffs;  /* coefficients stored per surfel (shs.shape[1]) */
    int32_t H, W;
    float tan_fovx, tan_fovy; /* unused by the arithmetic, kept for the call-site contract */
    float scale_modifier;
    float bg[3];
    float viewmatrix[16]; /* as stored by lightning/utils.py:39 (w2c transposed, row-vector convention) */
    float projmatrix[16]; /* lightning/utils.py:47 */
    float campos[3];      /* lightning/utils.py:48 */
} OracleCfg;

typedef struct {
    int32_t P, H, W, tiles_x, tiles_y;
    int64_t num_rendered;
    int64_t blended_pairs; /* (pixel, splat) pairs the composite actually blended: the path's useful work */
    float *transMats;      /* [P,9]  Tu,Tv,Tw */
    float *normal_opacity; /* [P,4] */
    float *rgb;            /* [P,3] */
    float *means2D;        /* [P,2] */
    float *depths;         /* [P] */
    int32_t *radii;        /* [P] */
    uint32_t *tiles_touched; /* [P] */
    uint32_t *rect;        /* [P,4] min.x min.y max.x max.y */
    uint8_t *clamped;      /* [P,3] */
    uint32_t *point_offsets; /* [P] inclusive scan */
    uint64_t *keys_sorted; /* [D] */
    uint32_t *point_list;  /* [D] */
    uint32_t *ranges;      /* [tiles,2] */
    float *final_T;        /* [3,H,W] : T, M1, M2 */
    uint32_t *n_contrib;   /* [2,H,W] : last_contributor, median_contributor */
} OracleState;

/* ---- small helpers ------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
/* float -> int as GPUs do it (cvt.rzi.s32 / v_cvt_i32_f32): saturating, NaN -> 0.  A plain C cast
 * is undefined out of range (x86 yields INT_MIN), which would change tile rectangles of surfels
 * whose projection blows up. */
static inline int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int)f;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* p (row vector) times the stored 4x4: x' = m0 x + m4 y + m8 z + m12 ... */
static inline void point4x3(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void vec4x3(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2];
}
static inline void vec4x3T(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
    o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
    o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}

/* rotation matrix from quaternion (w,x,y,z); R[r][c] */
static void quat_to_rotmat(const float *q, float R[3][3]) {
    float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    float s = 1.0f / sqrtf(n2);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
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

/* Pm = projmatrix (4x4, row-vector convention) times ndc2pix (4x3):
 * pixel = ((ndc + 1) * W - 1) / 2.  Pm[a][j], j = x*w, y*w, w               */
static void build_Pm(const OracleCfg *c, float Pm[4][3]) {
    const float hw = (float)c->W / 2.0f, hh = (float)c->H / 2.0f;
    const float cw = (float)(c->W - 1) / 2.0f, ch = (float)(c->H - 1) / 2.0f;
    for (int a = 0; a < 4; a++) {
        const float *row = c->projmatrix + 4 * a;
        Pm[a][0] = row[0] * hw + row[3] * cw;
        Pm[a][1] = row[1] * hh + row[3] * ch;
        Pm[a][2] = row[3];
    }
}

/* T(i,j) = sum_a Mrow_i[a] * Pm[a][j]; rows: L0 (u axis), L1 (v axis), centre */
static void compute_transmat(const OracleCfg *c, const float Pm[4][3], const float *p_orig,
                             const float *scale, const float *rot, float Tm[3][3], float normal[3],
                             float R[3][3]) {
    quat_to_rotmat(rot, R);
    const float sx = c->scale_modifier * scale[0], sy = c->scale_modifier * scale[1];
    float L0[3] = {R[0][0] * sx, R[1][0] * sx, R[2][0] * sx};
    float L1[3] = {R[0][1] * sy, R[1][1] * sy, R[2][1] * sy};
    float L2[3] = {R[0][2], R[1][2], R[2][2]};
    for (int j = 0; j < 3; j++) {
        Tm[0][j] = L0[0] * Pm[0][j] + L0[1] * Pm[1][j] + L0[2] * Pm[2][j];
        Tm[1][j] = L1[0] * Pm[0][j] + L1[1] * Pm[1][j] + L1[2] * Pm[2][j];
        Tm[2][j] = p_orig[0] * Pm[0][j] + p_orig[1] * Pm[1][j] + p_orig[2] * Pm[2][j] + Pm[3][j];
    }
    vec4x3(c->viewmatrix, L2, normal);
}

/* 3-sigma bounding box of the projected surfel.  T0,T1,T3 = x*w, y*w, w columns */
static int compute_aabb(const float T0[3], const float T1[3], const float T3[3], float cutoff,
                        float pt[2], float ext[2]) {
    const float t[3] = {cutoff * cutoff, cutoff * cutoff, -1.0f};
    float distance = T3[0] * T3[0] * t[0] + T3[1] * T3[1] * t[1] + T3[2] * T3[2] * t[2];
    if (distance == 0.0f) return 0;
    const float inv = 1.0f / distance;
    float f[3] = {inv * t[0], inv * t[1], inv * t[2]};
    pt[0] = f[0] * T0[0] * T3[0] + f[1] * T0[1] * T3[1] + f[2] * T0[2] * T3[2];
    pt[1] = f[0] * T1[0] * T3[0] + f[1] * T1[1] * T3[1] + f[2] * T1[2] * T3[2];
    float t0 = f[0] * T0[0] * T0[0] + f[1] * T0[1] * T0[1] + f[2] * T0[2] * T0[2];
    float t1 = f[0] * T1[0] * T1[0] + f[1] * T1[1] * T1[1] + f[2] * T1[2] * T1[2];
    float h0 = pt[0] * pt[0] - t0, h1 = pt[1] * pt[1] - t1;
    ext[0] = sqrtf(fmaxf(1e-4f, h0));
    ext[1] = sqrtf(fmaxf(1e-4f, h1));
    return 1;
}

static void get_rect(const float p[2], int max_radius, int gx, int gy, uint32_t r[4]) {
    r[0] = (uint32_t)imin(gx, imax(0, f2i_sat((p[0] - max_radius) / BLOCK_X)));
    r[1] = (uint32_t)imin(gy, imax(0, f2i_sat((p[1] - max_radius) / BLOCK_Y)));
    r[2] = (uint32_t)imin(gx, imax(0, f2i_sat((p[0] + max_radius + BLOCK_X - 1) / BLOCK_X)));
    r[3] = (uint32_t)imin(gy, imax(0, f2i_sat((p[1] + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

static void sh_to_rgb(const OracleCfg *c, const float *pos, const float *sh, float rgb[3],
                      uint8_t clamped[3]) {
    float dir[3] = {pos[0] - c->campos[0], pos[1] - c->campos[1], pos[2] - c->campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    float x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    for (int ch = 0; ch < 3; ch++) {
        const float *s = sh + ch; /* sh[k*3+ch] */
        float r = SH_C0 * s[0];
        if (c->sh_degree > 0) {
            r = r - SH_C1 * y * s[3] + SH_C1 * z * s[6] - SH_C1 * x * s[9];
            if (c->sh_degree > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * s[12] + SH_C2[1] * yz * s[15] +
                    SH_C2[2] * (2.0f * zz - xx - yy) * s[18] + SH_C2[3] * xz * s[21] +
                    SH_C2[4] * (xx - yy) * s[24];
                if (c->sh_degree > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * s[27] + SH_C3[1] * xy * z * s[30] +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * s[33] +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[36] +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * s[39] +
                        SH_C3[5] * z * (xx - yy) * s[42] + SH_C3[6] * x * (xx - 3.0f * yy) * s[45];
                }
            }
        }
        r += 0.5f;
        clamped[ch] = (r < 0.0f);
        rgb[ch] = fmaxf(r, 0.0f);
    }
}

/* ---- stable sort of (key, value) pairs: bottom-up merge sort -------------- */
static void sort_pairs(uint64_t *k, uint32_t *v, int64_t n) {
    if (n < 2) return;
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *ka = k, *kb = k2;
    uint32_t *va = v, *vb = v2;
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (ka[j] < ka[i]) { kb[o] = ka[j]; vb[o++] = va[j++]; }
                else { kb[o] = ka[i]; vb[o++] = va[i++]; }
            }
            while (i < mid) { kb[o] = ka[i]; vb[o++] = va[i++]; }
            while (j < hi) { kb[o] = ka[j]; vb[o++] = va[j++]; }
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != k) { memcpy(k, ka, sizeof(uint64_t) * n); memcpy(v, va, sizeof(uint32_t) * n); }
    free(k2); free(v2);
}

void oracle_free(OracleState *s) {
    if (!s) return;
    free(s->transMats); free(s->normal_opacity); free(s->rgb); free(s->means2D); free(s->depths);
    free(s->radii); free(s->tiles_touched); free(s->rect); free(s->clamped); free(s->point_offsets);
    free(s->keys_sorted); free(s->point_list); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

/* ---- per-pixel splat evaluation shared by forward and backward ------------ */
typedef struct {
    float sx, sy, rho3d, rho2d, dx, dy, depth, G, alpha, pz;
    float kx, ky, kz, lx, ly, lz;
} Hit;

/* returns 0 if the list entry is skipped for this pixel */
static inline int eval_splat(const float *Tr, const float *xy, float opa, float pxf, float pyf, Hit *h) {
    const float *Tu = Tr, *Tv = Tr + 3, *Tw = Tr + 6;
    h->kx = pxf * Tw[0] - Tu[0]; h->ky = pxf * Tw[1] - Tu[1]; h->kz = pxf * Tw[2] - Tu[2];
    h->lx = pyf * Tw[0] - Tv[0]; h->ly = pyf * Tw[1] - Tv[1]; h->lz = pyf * Tw[2] - Tv[2];
    float px = h->ky * h->lz - h->kz * h->ly;
    float py = h->kz * h->lx - h->kx * h->lz;
    float pz = h->kx * h->ly - h->ky * h->lx;
    if (pz == 0.0f) return 0;
    h->pz = pz;
    h->sx = px / pz; h->sy = py / pz;
    h->rho3d = h->sx * h->sx + h->sy * h->sy;
    h->dx = xy[0] - pxf; h->dy = xy[1] - pyf;
    h->rho2d = FILTER_INV_SQUARE * (h->dx * h->dx + h->dy * h->dy);
    float rho = fminf(h->rho3d, h->rho2d);
    h->depth = (h->rho3d <= h->rho2d) ? (h->sx * Tw[0] + h->sy * Tw[1]) + Tw[2] : Tw[2];
    if (h->depth < NEAR_N) return 0;
    float power = -0.5f * rho;
    if (power > 0.0f) return 0;
    h->G = expf(power);
    h->alpha = fminf(0.99f, opa * h->G);
    if (h->alpha < 1.0f / 255.0f) return 0;
    return 1;
}

/* =========================================================================
 * FORWARD
 * inputs: means3D [P,3], shs [P,M,3] or NULL, colors_precomp [P,3] or NULL,
 *         opacities [P], scales [P,2], rotations [P,4] (w,x,y,z),
 *         transMat_precomp [P,9] or NULL
 * outputs: out_color [3,H,W], out_others [7,H,W], radii [P]
 * ========================================================================= */
OracleState *oracle_forward(const OracleCfg *c, const float *means3D, const float *shs,
                            const float *colors_precomp, const float *opacities,
                            const float *scales, const float *rotations,
                            const float *transMat_precomp, float *out_color, float *out_others,
                            int32_t *radii_out) {
    const int P = c->P, W = c->W, H = c->H;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    OracleState *s = (OracleState *)calloc(1, sizeof(OracleState));
    s->P = P; s->H = H; s->W = W; s->tiles_x = gx; s->tiles_y = gy;
    const size_t Pn = P > 0 ? (size_t)P : 1;
    s->transMats = (float *)calloc(Pn * 9, 4);
    s->normal_opacity = (float *)calloc(Pn * 4, 4);
    s->rgb = (float *)calloc(Pn * 3, 4);
    s->means2D = (float *)calloc(Pn * 2, 4);
    s->depths = (float *)calloc(Pn, 4);
    s->radii = (int32_t *)calloc(Pn, 4);
    s->tiles_touched = (uint32_t *)calloc(Pn, 4);
    s->rect = (uint32_t *)calloc(Pn * 4, 4);
    s->clamped = (uint8_t *)calloc(Pn * 3, 1);
    s->point_offsets = (uint32_t *)calloc(Pn, 4);
    s->ranges = (uint32_t *)calloc((size_t)gx * gy * 2, 4);
    s->final_T = (float *)calloc((size_t)3 * H * W, 4);
    s->n_contrib = (uint32_t *)calloc((size_t)2 * H * W, 4);

    float Pm[4][3];
    build_Pm(c, Pm);

    /* ---- preprocess ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        const float *p_orig = means3D + 3 * idx;
        float p_view[3];
        point4x3(c->viewmatrix, p_orig, p_view);
        if (p_view[2] <= 0.2f) continue;
        float Tm[3][3], normal[3], R[3][3];
        if (transMat_precomp == NULL) {
            compute_transmat(c, Pm, p_orig, scales + 2 * idx, rotations + 4 * idx, Tm, normal, R);
        } else {
            const float *tp = transMat_precomp + 9 * idx;
            for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) Tm[i][j] = tp[3 * j + i];
            normal[0] = 0.f; normal[1] = 0.f; normal[2] = 1.f;
        }
        float T0[3] = {Tm[0][0], Tm[1][0], Tm[2][0]};
        float T1[3] = {Tm[0][1], Tm[1][1], Tm[2][1]};
        float T3[3] = {Tm[0][2], Tm[1][2], Tm[2][2]};
        float cosv = -(p_view[0] * normal[0] + p_view[1] * normal[1] + p_view[2] * normal[2]);
        if (cosv == 0.0f) continue;
        float mult = cosv > 0.0f ? 1.0f : -1.0f;
        normal[0] *= mult; normal[1] *= mult; normal[2] *= mult;

        float pt[2], ext[2];
        if (!compute_aabb(T0, T1, T3, CUTOFF, pt, ext)) continue;
        float radius = ceilf(fmaxf(fmaxf(ext[0], ext[1]), CUTOFF * FILTER_SIZE));
        uint32_t r[4];
        get_rect(pt, f2i_sat(radius), gx, gy, r);
        if ((r[2] - r[0]) * (r[3] - r[1]) == 0) continue;

        /* transMats are stored only for visible surfels that reach this point (upstream stores
         * them before the cull tests; no later stage reads the culled ones).                   */
        float *To = s->transMats + 9 * idx;
        To[0] = T0[0]; To[1] = T0[1]; To[2] = T0[2];
        To[3] = T1[0]; To[4] = T1[1]; To[5] = T1[2];
        To[6] = T3[0]; To[7] = T3[1]; To[8] = T3[2];
        if (colors_precomp == NULL) {
            sh_to_rgb(c, p_orig, shs + (size_t)idx * c->sh_coeffs * 3, s->rgb + 3 * idx,
                      s->clamped + 3 * idx);
        } else {
            for (int ch = 0; ch < 3; ch++) s->rgb[3 * idx + ch] = colors_precomp[3 * idx + ch];
        }
        s->depths[idx] = p_view[2];
        s->radii[idx] = f2i_sat(radius);
        s->means2D[2 * idx] = pt[0]; s->means2D[2 * idx + 1] = pt[1];
        s->normal_opacity[4 * idx + 0] = normal[0];
        s->normal_opacity[4 * idx + 1] = normal[1];
        s->normal_opacity[4 * idx + 2] = normal[2];
        s->normal_opacity[4 * idx + 3] = opacities[idx];
        s->tiles_touched[idx] = (r[3] - r[1]) * (r[2] - r[0]);
        memcpy(s->rect + 4 * idx, r, 16);
    }
    if (radii_out) memcpy(radii_out, s->radii, sizeof(int32_t) * P);

    /* ---- inclusive scan, duplicate with keys, stable sort, tile ranges ---- */
    uint64_t acc = 0;
    for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = (uint32_t)acc; }
    const int64_t D = (int64_t)acc;
    s->num_rendered = D;
    s->keys_sorted = (uint64_t *)malloc(sizeof(uint64_t) * (D > 0 ? D : 1));
    s->point_list = (uint32_t *)malloc(sizeof(uint32_t) * (D > 0 ? D : 1));
    for (int idx = 0; idx < P; idx++) {
        if (s->radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0 : s->point_offsets[idx - 1];
        const uint32_t *r = s->rect + 4 * idx;
        for (uint32_t y = r[1]; y < r[3]; y++)
            for (uint32_t x = r[0]; x < r[2]; x++) {
                uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                key <<= 32;
                key |= f2u(s->depths[idx]);
                s->keys_sorted[off] = key;
                s->point_list[off] = (uint32_t)idx;
                off++;
            }
    }
    sort_pairs(s->keys_sorted, s->point_list, D);
    for (int64_t i = 0; i < D; i++) {
        uint32_t t = (uint32_t)(s->keys_sorted[i] >> 32);
        if (i == 0) s->ranges[2 * t] = 0;
        else {
            uint32_t tp = (uint32_t)(s->keys_sorted[i - 1] >> 32);
            if (t != tp) { s->ranges[2 * tp + 1] = (uint32_t)i; s->ranges[2 * t] = (uint32_t)i; }
        }
        if (i == D - 1) s->ranges[2 * t + 1] = (uint32_t)D;
    }

    /* ---- per-pixel front-to-back composite ---- */
    const size_t HW = (size_t)H * W;
    int64_t blended_total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : blended_total)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const size_t pix = (size_t)pyi * W + pxi;
                const float pxf = (float)pxi, pyf = (float)pyi;
                float T = 1.0f, C[3] = {0, 0, 0}, N[3] = {0, 0, 0};
                float Dd = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
                uint32_t contributor = 0, last_contributor = 0, median_contributor = 0;
                for (uint32_t i = r0; i < r1; i++) {
                    contributor++;
                    const uint32_t id = s->point_list[i];
                    const float *no = s->normal_opacity + 4 * id;
                    Hit h;
                    if (!eval_splat(s->transMats + 9 * id, s->means2D + 2 * id, no[3], pxf, pyf, &h))
                        continue;
                    float test_T = T * (1 - h.alpha);
                    if (test_T < 0.0001f) break; /* done */
                    float w = h.alpha * T;
                    float A = 1 - T;
                    float m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / h.depth);
                    distortion += (m * m * A + M2 - 2 * m * M1) * w;
                    Dd += h.depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                    if (T > 0.5f) { median_depth = h.depth; median_contributor = contributor; }
                    for (int ch = 0; ch < 3; ch++) N[ch] += no[ch] * w;
                    for (int ch = 0; ch < 3; ch++) C[ch] += s->rgb[3 * id + ch] * w;
                    T = test_T;
                    last_contributor = contributor;
                    blended_total++;
                }
                s->final_T[pix] = T;
                s->final_T[pix + HW] = M1;
                s->final_T[pix + 2 * HW] = M2;
                s->n_contrib[pix] = last_contributor;
                s->n_contrib[pix + HW] = median_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = C[ch] + T * c->bg[ch];
                out_others[0 * HW + pix] = Dd;
                out_others[1 * HW + pix] = 1 - T;
                for (int ch = 0; ch < 3; ch++) out_others[(2 + ch) * HW + pix] = N[ch];
                out_others[5 * HW + pix] = median_depth;
                out_others[6 * HW + pix] = distortion;
            }
    }
    s->blended_pairs = blended_total;
    return s;
}

/* =========================================================================
 * BACKWARD
 * lowpass_depth_quirk: 1 = published behaviour (the screen-space low-pass branch
 * propagates dL_dz * (s.x, s.y, 1) into Tw although its forward depth is Tw.z);
 * 0 = analytically exact (0,0,1) -- used only for the autograd cross-check.
 * ========================================================================= */
static inline void atomic_add_d(double *p, double v) {
#pragma omp atomic
    *p += v;
}

void oracle_backward(const OracleCfg *c, const OracleState *s, const float *means3D,
                     const float *shs, const float *colors_precomp, const float *scales,
                     const float *rotations, const float *transMat_precomp,
                     const float *dL_dpixels, const float *dL_dothers, int lowpass_depth_quirk,
                     float *dL_dmeans3D, float *dL_dmeans2D_out, float *dL_dshs,
                     float *dL_dcolors_out, float *dL_dopacity, float *dL_dscales,
                     float *dL_drots, float *dL_dtransMat_out) {
    const int P = c->P, W = c->W, H = c->H, gx = s->tiles_x, gy = s->tiles_y;
    const size_t HW = (size_t)H * W;
    const size_t Pn = P > 0 ? (size_t)P : 1;
    double *gT = (double *)calloc(Pn * 9, 8);   /* dL_dtransMat */
    double *gM2 = (double *)calloc(Pn * 2, 8);  /* dL_dmean2D */
    double *gN = (double *)calloc(Pn * 3, 8);   /* dL_dnormal */
    double *gO = (double *)calloc(Pn, 8);       /* dL_dopacity */
    double *gC = (double *)calloc(Pn * 3, 8);   /* dL_dcolor */

#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = s->ranges[2 * tile];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const size_t pix = (size_t)pyi * W + pxi;
                const float pxf = (float)pxi, pyf = (float)pyi;
                const float T_final = s->final_T[pix];
                float T = T_final;
                const uint32_t last_contributor = s->n_contrib[pix];
                const uint32_t median_contributor = s->n_contrib[pix + HW];
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3];
                for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpixels[ch * HW + pix];
                const float dL_ddepth = dL_dothers[0 * HW + pix];
                const float dL_daccum = dL_dothers[1 * HW + pix];
                float dL_dnormal2D[3];
                for (int ch = 0; ch < 3; ch++) dL_dnormal2D[ch] = dL_dothers[(2 + ch) * HW + pix];
                const float dL_dmedian_depth = dL_dothers[5 * HW + pix];
                const float dL_dreg = dL_dothers[6 * HW + pix];
                float last_depth = 0, last_normal[3] = {0, 0, 0};
                float accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = {0, 0, 0};
                const float final_D = s->final_T[pix + HW], final_D2 = s->final_T[pix + 2 * HW];
                const float final_A = 1 - T_final;
                float last_dL_dT = 0, last_alpha = 0, last_color[3] = {0, 0, 0};
                float bg_dot_dpixel = 0;
                for (int ch = 0; ch < 3; ch++) bg_dot_dpixel += c->bg[ch] * dL_dpixel[ch];

                /* back to front over the entries [r0, r0 + last_contributor) */
                for (uint32_t contributor = last_contributor; contributor-- > 0;) {
                    const uint32_t id = s->point_list[r0 + contributor];
                    const float *Tr = s->transMats + 9 * id;
                    const float *Tw = Tr + 6;
                    const float *no = s->normal_opacity + 4 * id;
                    Hit h;
                    if (!eval_splat(Tr, s->means2D + 2 * id, no[3], pxf, pyf, &h)) continue;
                    const float alpha = h.alpha, G = h.G, c_d = h.depth;
                    T = T / (1.f - alpha);
                    const float w = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 3; ch++) {
                        const float col = s->rgb[3 * id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = col;
                        dL_dalpha += (col - accum_rec[ch]) * dL_dpixel[ch];
                        atomic_add_d(&gC[3 * id + ch], (double)(w * dL_dpixel[ch]));
                    }
                    float dL_dz = 0.0f, dL_dweight = 0.0f;
                    const float m_d = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / c_d);
                    const float dmd_dd = (FAR_N * NEAR_N) / ((FAR_N - NEAR_N) * c_d * c_d);
                    if (contributor + 1 == median_contributor) dL_dz += dL_dmedian_depth;
                    dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                    dL_dalpha += dL_dweight - last_dL_dT;
                    last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                    const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                    dL_dz += dL_dmd * dmd_dd;

                    accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                    accum_alpha_rec = last_alpha * 1.0f + (1.f - last_alpha) * accum_alpha_rec;
                    dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
                    for (int ch = 0; ch < 3; ch++) {
                        accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                        last_normal[ch] = no[ch];
                        dL_dalpha += (no[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
                        atomic_add_d(&gN[3 * id + ch], (double)(alpha * T * dL_dnormal2D[ch]));
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                    const float dL_dG = no[3] * dL_dalpha;
                    dL_dz += alpha * T * dL_ddepth;

                    if (h.rho3d <= h.rho2d) {
                        const float dL_dsx = dL_dG * -G * h.sx + dL_dz * Tw[0];
                        const float dL_dsy = dL_dG * -G * h.sy + dL_dz * Tw[1];
                        const float dsx_pz = dL_dsx / h.pz, dsy_pz = dL_dsy / h.pz;
                        const float dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * h.sx + dsy_pz * h.sy);
                        /* dL_dk = cross(l, dL_dp); dL_dl = cross(dL_dp, k) */
                        const float dkx = h.ly * dpz - h.lz * dpy;
                        const float dky = h.lz * dpx - h.lx * dpz;
                        const float dkz = h.lx * dpy - h.ly * dpx;
                        const float dlx = dpy * h.kz - dpz * h.ky;
                        const float dly = dpz * h.kx - dpx * h.kz;
                        const float dlz = dpx * h.ky - dpy * h.kx;
                        atomic_add_d(&gT[9 * id + 0], (double)(-dkx));
                        atomic_add_d(&gT[9 * id + 1], (double)(-dky));
                        atomic_add_d(&gT[9 * id + 2], (double)(-dkz));
                        atomic_add_d(&gT[9 * id + 3], (double)(-dlx));
                        atomic_add_d(&gT[9 * id + 4], (double)(-dly));
                        atomic_add_d(&gT[9 * id + 5], (double)(-dlz));
                        atomic_add_d(&gT[9 * id + 6], (double)(pxf * dkx + pyf * dlx + dL_dz * h.sx));
                        atomic_add_d(&gT[9 * id + 7], (double)(pxf * dky + pyf * dly + dL_dz * h.sy));
                        atomic_add_d(&gT[9 * id + 8], (double)(pxf * dkz + pyf * dlz + dL_dz * 1.0f));
                    } else {
                        const float dG_ddelx = -G * FILTER_INV_SQUARE * h.dx;
                        const float dG_ddely = -G * FILTER_INV_SQUARE * h.dy;
                        atomic_add_d(&gM2[2 * id + 0], (double)(dL_dG * dG_ddelx));
                        atomic_add_d(&gM2[2 * id + 1], (double)(dL_dG * dG_ddely));
                        if (lowpass_depth_quirk) {
                            atomic_add_d(&gT[9 * id + 6], (double)(h.sx * dL_dz));
                            atomic_add_d(&gT[9 * id + 7], (double)(h.sy * dL_dz));
                        }
                        atomic_add_d(&gT[9 * id + 8], (double)dL_dz);
                    }
                    atomic_add_d(&gO[id], (double)(G * dL_dalpha));
                }
            }
    }

    /* ---- backward of preprocess ---- */
    float Pm[4][3];
    build_Pm(c, Pm);
    if (dL_dmeans3D) memset(dL_dmeans3D, 0, sizeof(float) * 3 * Pn);
    if (dL_dmeans2D_out) memset(dL_dmeans2D_out, 0, sizeof(float) * 3 * Pn);
    if (dL_dshs) memset(dL_dshs, 0, sizeof(float) * 3 * Pn * (size_t)(c->sh_coeffs > 0 ? c->sh_coeffs : 1));
    if (dL_dcolors_out) memset(dL_dcolors_out, 0, sizeof(float) * 3 * Pn);
    if (dL_dscales) memset(dL_dscales, 0, sizeof(float) * 2 * Pn);
    if (dL_drots) memset(dL_drots, 0, sizeof(float) * 4 * Pn);
    if (dL_dtransMat_out) memset(dL_dtransMat_out, 0, sizeof(float) * 9 * Pn);
    for (int i = 0; i < P; i++) dL_dopacity[i] = (float)gO[i];

#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(s->radii[idx] > 0)) continue;
        float dT[9];
        for (int k = 0; k < 9; k++) dT[k] = (float)gT[9 * idx + k];
        const float *Tr = s->transMats + 9 * idx; /* Tu, Tv, Tw */
        const float m2x = (float)gM2[2 * idx], m2y = (float)gM2[2 * idx + 1];
        if (m2x != 0.0f || m2y != 0.0f) {
            /* through the AABB-centre formula: centre = sum(f * T0 * T3), f = t / dot(t, T3*T3) */
            const float t[3] = {9.0f, 9.0f, -1.0f};
            const float *T0 = Tr, *T1 = Tr + 3, *T3 = Tr + 6;
            float d = t[0] * T3[0] * T3[0] + t[1] * T3[1] * T3[1] + t[2] * T3[2] * T3[2];
            float f[3] = {t[0] * (1.0f / d), t[1] * (1.0f / d), t[2] * (1.0f / d)};
            float dL_dT3[3], dL_df[3];
            for (int k = 0; k < 3; k++) {
                dT[0 + k] += m2x * f[k] * T3[k];
                dT[3 + k] += m2y * f[k] * T3[k];
                dL_dT3[k] = m2x * f[k] * T0[k] + m2y * f[k] * T1[k];
                dL_df[k] = m2x * T0[k] * T3[k] + m2y * T1[k] * T3[k];
            }
            float dL_dd = (dL_df[0] * f[0] + dL_df[1] * f[1] + dL_df[2] * f[2]) * (-1.0f / d);
            for (int k = 0; k < 3; k++) {
                dL_dT3[k] += dL_dd * (t[k] * T3[k] * 2.0f);
                dT[6 + k] += dL_dT3[k];
            }
        }
        if (transMat_precomp != NULL) {
            if (dL_dtransMat_out) for (int k = 0; k < 9; k++) dL_dtransMat_out[9 * idx + k] = dT[k];
        } else {
            const float *p_orig = means3D + 3 * idx;
            float Tm[3][3], normal[3], R[3][3];
            compute_transmat(c, Pm, p_orig, scales + 2 * idx, rotations + 4 * idx, Tm, normal, R);
            /* dL_dMrow_i[a] = sum_j dL_dT(i,j) Pm[a][j];  dT[3*j+i] = dL_dT(i,j) */
            float dM[3][3];
            for (int i = 0; i < 3; i++)
                for (int a = 0; a < 3; a++)
                    dM[i][a] = dT[0 + i] * Pm[a][0] + dT[3 + i] * Pm[a][1] + dT[6 + i] * Pm[a][2];
            float dn[3] = {(float)gN[3 * idx], (float)gN[3 * idx + 1], (float)gN[3 * idx + 2]};
            float dtn[3];
            vec4x3T(c->viewmatrix, dn, dtn);
            float p_view[3];
            point4x3(c->viewmatrix, p_orig, p_view);
            float cosv = -(p_view[0] * normal[0] + p_view[1] * normal[1] + p_view[2] * normal[2]);
            float mult = cosv > 0.0f ? 1.0f : -1.0f;
            dtn[0] *= mult; dtn[1] *= mult; dtn[2] *= mult;
            const float sx = c->scale_modifier * scales[2 * idx], sy = c->scale_modifier * scales[2 * idx + 1];
            /* V(r,c) = dL/dR(r,c) */
            float V[3][3];
            for (int r = 0; r < 3; r++) { V[r][0] = dM[0][r] * sx; V[r][1] = dM[1][r] * sy; V[r][2] = dtn[r]; }
            dL_dscales[2 * idx + 0] = c->scale_modifier * (dM[0][0] * R[0][0] + dM[0][1] * R[1][0] + dM[0][2] * R[2][0]);
            dL_dscales[2 * idx + 1] = c->scale_modifier * (dM[1][0] * R[0][1] + dM[1][1] * R[1][1] + dM[1][2] * R[2][1]);
            const float *q = rotations + 4 * idx;
            float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            float sn = 1.0f / sqrtf(n2);
            float w = q[0] * sn, x = q[1] * sn, y = q[2] * sn, z = q[3] * sn;
            dL_drots[4 * idx + 0] = 2.f * (x * (V[2][1] - V[1][2]) + y * (V[0][2] - V[2][0]) + z * (V[1][0] - V[0][1]));
            dL_drots[4 * idx + 1] = 2.f * (-2.f * x * (V[1][1] + V[2][2]) + y * (V[1][0] + V[0][1]) + z * (V[2][0] + V[0][2]) + w * (V[2][1] - V[1][2]));
            dL_drots[4 * idx + 2] = 2.f * (x * (V[1][0] + V[0][1]) - 2.f * y * (V[0][0] + V[2][2]) + z * (V[2][1] + V[1][2]) + w * (V[0][2] - V[2][0]));
            dL_drots[4 * idx + 3] = 2.f * (x * (V[2][0] + V[0][2]) + y * (V[2][1] + V[1][2]) - 2.f * z * (V[0][0] + V[1][1]) + w * (V[1][0] - V[0][1]));
            dL_dmeans3D[3 * idx + 0] = dM[2][0];
            dL_dmeans3D[3 * idx + 1] = dM[2][1];
            dL_dmeans3D[3 * idx + 2] = dM[2][2];
        }

        if (colors_precomp == NULL) {
            const float *pos = means3D + 3 * idx;
            const float *sh = shs + (size_t)idx * c->sh_coeffs * 3;
            float *dsh = dL_dshs + (size_t)idx * c->sh_coeffs * 3;
            float dir_o[3] = {pos[0] - c->campos[0], pos[1] - c->campos[1], pos[2] - c->campos[2]};
            float len = sqrtf(dir_o[0] * dir_o[0] + dir_o[1] * dir_o[1] + dir_o[2] * dir_o[2]);
            float x = dir_o[0] / len, y = dir_o[1] / len, z = dir_o[2] / len;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++)
                dRGB[ch] = (float)gC[3 * idx + ch] * (s->clamped[3 * idx + ch] ? 0.0f : 1.0f);
            float ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
                const float *sc = sh + ch;
                float *ds = dsh + ch;
                const float g = dRGB[ch];
                float dx = 0, dy = 0, dz = 0;
                ds[0] = SH_C0 * g;
                if (c->sh_degree > 0) {
                    ds[3] = -SH_C1 * y * g; ds[6] = SH_C1 * z * g; ds[9] = -SH_C1 * x * g;
                    dx = -SH_C1 * sc[9]; dy = -SH_C1 * sc[3]; dz = SH_C1 * sc[6];
                    if (c->sh_degree > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        ds[12] = SH_C2[0] * xy * g; ds[15] = SH_C2[1] * yz * g;
                        ds[18] = SH_C2[2] * (2.f * zz - xx - yy) * g;
                        ds[21] = SH_C2[3] * xz * g; ds[24] = SH_C2[4] * (xx - yy) * g;
                        dx += SH_C2[0] * y * sc[12] + SH_C2[2] * 2.f * -x * sc[18] + SH_C2[3] * z * sc[21] + SH_C2[4] * 2.f * x * sc[24];
                        dy += SH_C2[0] * x * sc[12] + SH_C2[1] * z * sc[15] + SH_C2[2] * 2.f * -y * sc[18] + SH_C2[4] * 2.f * -y * sc[24];
                        dz += SH_C2[1] * y * sc[15] + SH_C2[2] * 2.f * 2.f * z * sc[18] + SH_C2[3] * x * sc[21];
                        if (c->sh_degree > 2) {
                            ds[27] = SH_C3[0] * y * (3.f * xx - yy) * g;
                            ds[30] = SH_C3[1] * xy * z * g;
                            ds[33] = SH_C3[2] * y * (4.f * zz - xx - yy) * g;
                            ds[36] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            ds[39] = SH_C3[4] * x * (4.f * zz - xx - yy) * g;
                            ds[42] = SH_C3[5] * z * (xx - yy) * g;
                            ds[45] = SH_C3[6] * x * (xx - 3.f * yy) * g;
                            dx += SH_C3[0] * sc[27] * 3.f * 2.f * xy + SH_C3[1] * sc[30] * yz +
                                  SH_C3[2] * sc[33] * -2.f * xy + SH_C3[3] * sc[36] * -3.f * 2.f * xz +
                                  SH_C3[4] * sc[39] * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * sc[42] * 2.f * xz + SH_C3[6] * sc[45] * 3.f * (xx - yy);
                            dy += SH_C3[0] * sc[27] * 3.f * (xx - yy) + SH_C3[1] * sc[30] * xz +
                                  SH_C3[2] * sc[33] * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * sc[36] * -3.f * 2.f * yz + SH_C3[4] * sc[39] * -2.f * xy +
                                  SH_C3[5] * sc[42] * -2.f * yz + SH_C3[6] * sc[45] * -3.f * 2.f * xy;
                            dz += SH_C3[1] * sc[30] * xy + SH_C3[2] * sc[33] * 4.f * 2.f * yz +
                                  SH_C3[3] * sc[36] * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * sc[39] * 4.f * 2.f * xz + SH_C3[5] * sc[42] * (xx - yy);
                        }
                    }
                }
                ddir[0] += dx * g; ddir[1] += dy * g; ddir[2] += dz * g;
            }
            /* through dir = dir_o / |dir_o| */
            float sum2 = dir_o[0] * dir_o[0] + dir_o[1] * dir_o[1] + dir_o[2] * dir_o[2];
            float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            float dmx = ((sum2 - dir_o[0] * dir_o[0]) * ddir[0] - dir_o[1] * dir_o[0] * ddir[1] - dir_o[2] * dir_o[0] * ddir[2]) * invsum32;
            float dmy = (-dir_o[0] * dir_o[1] * ddir[0] + (sum2 - dir_o[1] * dir_o[1]) * ddir[1] - dir_o[2] * dir_o[1] * ddir[2]) * invsum32;
            float dmz = (-dir_o[0] * dir_o[2] * ddir[0] - dir_o[1] * dir_o[2] * ddir[1] + (sum2 - dir_o[2] * dir_o[2]) * ddir[2]) * invsum32;
            dL_dmeans3D[3 * idx + 0] += dmx;
            dL_dmeans3D[3 * idx + 1] += dmy;
            dL_dmeans3D[3 * idx + 2] += dmz;
        } else if (dL_dcolors_out) {
            for (int ch = 0; ch < 3; ch++) dL_dcolors_out[3 * idx + ch] = (float)gC[3 * idx + ch];
        }
        /* screen-space gradient handed back for densification heuristics (published behaviour):
         * overwritten with dL/dTu.z, dL/dTv.z scaled to NDC                                     */
        if (dL_dmeans2D_out) {
            const float depth = Tr[8];
            dL_dmeans2D_out[3 * idx + 0] = (float)gT[9 * idx + 2] * depth * 0.5f * (float)W;
            dL_dmeans2D_out[3 * idx + 1] = (float)gT[9 * idx + 5] * depth * 0.5f * (float)H;
        }
    }
    free(gT); free(gM2); free(gN); free(gO); free(gC);
}

/* frustum test used by GaussianRasterizer.markVisible */
void oracle_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present) {
    for (int i = 0; i < P; i++) {
        float pv[3];
        point4x3(viewmatrix, means3D + 3 * i, pv);
        present[i] = pv[2] > 0.2f;
    }
}

/* accessors for ctypes */
int64_t oracle_num_rendered(const OracleState *s) { return s->num_rendered; }
int64_t oracle_blended_pairs(const OracleState *s) { return s->blended_pairs; }
const void *oracle_state_ptr(const OracleState *s, int which) {
    switch (which) {
    case 0: return s->transMats; case 1: return s->normal_opacity; case 2: return s->rgb;
    case 3: return s->means2D; case 4: return s->depths; case 5: return s->radii;
    case 6: return s->tiles_touched; case 7: return s->rect; case 8: return s->clamped;
    case 9: return s->point_offsets; case 10: return s->keys_sorted; case 11: return s->point_list;
    case 12: return s->ranges; case 13: return s->final_T; case 14: return s->n_contrib;
    default: return NULL;
    }
}
