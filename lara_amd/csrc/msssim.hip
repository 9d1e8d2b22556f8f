// msssim.hip -- the MS-SSIM term of LaRa's loss (lightning/loss.py:15, :42-45) on gfx950: forward and backward.
//
// Reference: `pytorch_msssim.MS_SSIM(data_range=1.0, size_average=True, channel=3)` [package absent from this image; restated from
// its published algorithm in lara_amd/loss.py: ms_ssim, which tests/test_loss_cpu.py holds to an independent float64 restatement;
// this file is held to that torch restatement].  Five scales; per scale an 11-tap sigma-1.5 Gaussian 'valid' filter of x, y, x^2,
// y^2, xy, the contrast-structure map cs = (2 sigma_xy + C2) / (sigma_x^2 + sigma_y^2 + C2) and (last scale) the SSIM map
// ((2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1)) cs, their means per (image, channel); 2 x 2 average pooling between scales.
//
// torch runs this as banded matrix products / pooling / ~60 elementwise kernels per image set and direction: 31 ms per training
// step for the coarse and the fine image of 4 scenes x 8 views @512^2 -- more than the raster's backward.  Here, per scale:
//   msssim_pool      x_{l+1}, y_{l+1} = avgpool2(x_l, y_l)                           (level 0 reads the caller's layouts in place)
//   msssim_maps<0>   tile 32 x 32 of the filtered domain from a 42 x 42 input tile in LDS, separable filter of the five maps
//                    (horizontal into LDS, vertical in registers), cs / ssim, per-workgroup partial sums      -> means (forward)
//   msssim_maps<1>   the same arithmetic, but leaves the three per-position gradients dL/dmu_x, dL/d(x^2 filtered), dL/d(xy filtered)
//   msssim_back      dx = filter^T(g_mu) + 2 x filter^T(g_xx) + y filter^T(g_xy) + 1/4 dx_{l+1}[pooled position]
// The means' combination (relu, powers, product, mean over images and channels: [5, N C] numbers) stays in torch, with autograd:
// the backward here receives dL/d(mean) per (scale, image, channel).  No atomics anywhere: reproducible.
#include "common.h"
#include "../../include/lara_loss.h"

namespace {

constexpr int MS_LEVELS = 5, MS_WIN = 11, MS_T = 32, MS_IN = MS_T + MS_WIN - 1;   // 42

struct MsWin { float w[MS_WIN]; };

struct MsView {       // value(n, c, y, x) = p[n sN + c sC + y sY + (x / Wv) sV + (x % Wv) sX]; a batch holds < 2^31 elements (checked)
    float *p;
    int sN, sC, sY, sV, sX;
    int Wv;
};
__device__ __forceinline__ int ms_at(const MsView &v, const int n, const int c, const int y, const int x) {
    const int xv = x / v.Wv, xr = x - xv * v.Wv;
    return n * v.sN + c * v.sC + y * v.sY + xv * v.sV + xr * v.sX;
}

// the same for x >= x_base where v_base = x_base / Wv was divided once (per tile): at most a few view crossings inside a tile
__device__ __forceinline__ int ms_at_from(const MsView &v, const int n, const int c, const int y, const int x, const int v_base,
                                          const int x_of_v_base) {
    int xv = v_base, xr = x - x_of_v_base;
    while (xr >= v.Wv) { xr -= v.Wv; xv++; }
    return n * v.sN + c * v.sC + y * v.sY + xv * v.sV + xr * v.sX;
}

// out[i][j] = 1/4 sum over a, b of in[2i + a - py][2j + b - px] (zero outside): avg_pool2d(kernel 2, padding (H % 2, W % 2))
__global__ void __launch_bounds__(256)
msssim_pool_kernel(const MsView X, const MsView Y, const int C, const int H, const int W, const int Ho, const int Wo,
                   float *__restrict__ Xo, float *__restrict__ Yo) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    if (i >= Ho || j >= Wo) return;
    const int py = H & 1, px = W & 1;
    for (int c = 0; c < C; c++) {      // (all channels of a pixel by one thread: the channel-last level 0 is read line by line)
    const int nc = n * C + c;
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int y = 2 * i + a - py, x = 2 * j + b - px;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                sx += X.p[ms_at(X, n, c, y, x)];
                sy += Y.p[ms_at(Y, n, c, y, x)];
            }
        }
    const size_t o = ((size_t)nc * Ho + i) * Wo + j;
    Xo[o] = 0.25f * sx;
    Yo[o] = 0.25f * sy;
    }
}

// MODE 0: per-workgroup partial sums of the ssim and cs maps -> partial[nc][tile][2]
// MODE 1: the gradient maps G[3][nc][Hb][Wb] = dL/d(mu_x), dL/d(filtered x^2), dL/d(filtered x y) given dmeans[nc][2] = dL/d(mean
//         ssim), dL/d(mean cs)
// Both filter passes are register-blocked: a thread produces FOUR adjacent outputs from 14 inputs it reads once (16-byte LDS reads
// along the row; one column, four rows down the column), a third of the LDS reads of one output per thread -- the first version
// spent more issue slots on ds_read_b32 than on the filter's multiply-adds (level 0: 375 us per launch; this form: see DESIGN 3.15).
constexpr int MS_SW = MS_IN + 2, MS_HW = MS_T + 4;     // LDS row lengths (floats): 44 and 36, multiples of 4 (aligned float4 rows)
template <int MODE>
__global__ void __launch_bounds__(256)
msssim_maps_kernel(const MsView X, const MsView Y, const int C, const int H, const int W, const MsWin win, const float C1,
                   const float C2, float *__restrict__ partial, const float *__restrict__ dmeans, float *__restrict__ G) {
    __shared__ __attribute__((aligned(16))) float sx[MS_IN][MS_SW], sy[MS_IN][MS_SW];
    __shared__ __attribute__((aligned(16))) float hh[5][MS_IN][MS_HW];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int ty0 = blockIdx.y * MS_T, tx0 = blockIdx.x * MS_T;
    const int Hb = H - (MS_WIN - 1), Wb = W - (MS_WIN - 1);
    const int vb = tx0 / X.Wv, vbx = vb * X.Wv;      // (X, Y and their gradient share Wv)
    // One workgroup takes the tile of ALL channels of an image, one after the other: in the renderer's channel-last layout a
    // channel plane uses 4 of every 12 bytes of a cache line, and a grid over (image, channel) fetched every line three times from
    // HBM (1 GB per level-0 launch: that, not the filter, was the first version's 375 us); here the second and third channel hit L2.
    for (int c = 0; c < C; c++) {
    const int nc = n * C + c, NC = gridDim.z * C;
    if (c) __syncthreads();
    {   // all of a thread's loads in flight before the first is stored (a rolled loop paid the memory latency once per element:
        // 8 round trips per channel and tile -- THAT was the first version's 375 us at level 0, not the filter)
        constexpr int NL = (MS_IN * MS_SW + 255) / 256;
        float lx[NL], ly[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int idx = tid + 256 * k, r = idx / MS_SW, cc = idx - r * MS_SW, y = ty0 + r, x = tx0 + cc;
            const bool in = idx < MS_IN * MS_SW && y < H && x < W;
            lx[k] = in ? X.p[ms_at_from(X, n, c, y, x, vb, vbx)] : 0.f;
            ly[k] = in ? Y.p[ms_at_from(Y, n, c, y, x, vb, vbx)] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int idx = tid + 256 * k, r = idx / MS_SW, cc = idx - r * MS_SW;
            if (idx < MS_IN * MS_SW) { sx[r][cc] = lx[k]; sy[r][cc] = ly[k]; }
        }
    }
    __syncthreads();
    for (int item = tid; item < MS_IN * (MS_T / 4); item += 256) {      // along the row: 4 outputs from 14 inputs
        const int r = item >> 3, c0 = (item & 7) * 4;
        float xv[16], yv[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 a = *(const float4 *)&sx[r][c0 + 4 * q], b = *(const float4 *)&sy[r][c0 + 4 * q];
            xv[4 * q] = a.x; xv[4 * q + 1] = a.y; xv[4 * q + 2] = a.z; xv[4 * q + 3] = a.w;
            yv[4 * q] = b.x; yv[4 * q + 1] = b.y; yv[4 * q + 2] = b.z; yv[4 * q + 3] = b.w;
        }
        float o[5][4];
#pragma unroll
        for (int m = 0; m < 5; m++)
#pragma unroll
            for (int e = 0; e < 4; e++) o[m][e] = 0.f;
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const float x1 = xv[k], y1 = yv[k], xx = x1 * x1, yy = y1 * y1, xy = x1 * y1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int t = k - e;
                if (t >= 0 && t < MS_WIN) {
                    const float w = win.w[t];
                    o[0][e] += w * x1; o[1][e] += w * y1; o[2][e] += w * xx; o[3][e] += w * yy; o[4][e] += w * xy;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 5; m++) *(float4 *)&hh[m][r][c0] = make_float4(o[m][0], o[m][1], o[m][2], o[m][3]);
    }
    __syncthreads();
    float s_ssim = 0.f, s_cs = 0.f;
    float gS = 0.f, gC = 0.f;
    if (MODE == 1) {
        const float inv = 1.0f / ((float)Hb * (float)Wb);
        gS = dmeans[2 * nc] * inv;
        gC = dmeans[2 * nc + 1] * inv;
    }
    {                                                                   // down the column: 4 outputs from 14 rows, per map
        const int cc = tid & 31, r0 = (tid >> 5) * 4;
        float v[5][4];
#pragma unroll
        for (int m = 0; m < 5; m++) {
            float hv[14];
#pragma unroll
            for (int k = 0; k < 14; k++) hv[k] = hh[m][r0 + k][cc];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < MS_WIN; t++) a += win.w[t] * hv[e + t];
                v[m][e] = a;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int r = r0 + e;
            const bool valid = ty0 + r < Hb && tx0 + cc < Wb;
            const float mu1 = v[0][e], mu2 = v[1][e];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float A2 = 2.0f * (v[4][e] - mu12) + C2, B2 = (v[2][e] - mu1_sq) + (v[3][e] - mu2_sq) + C2;
            const float A1 = 2.0f * mu12 + C1, B1 = mu1_sq + mu2_sq + C1;
            // (v_rcp_f32 + one Newton step: the full division sequence was a tenth of the kernel's instructions)
            float iB2 = __builtin_amdgcn_rcpf(B2), iB1 = __builtin_amdgcn_rcpf(B1);
            iB2 = iB2 * (2.0f - B2 * iB2); iB1 = iB1 * (2.0f - B1 * iB1);
            const float cs = A2 * iB2, L = A1 * iB1;
            if (MODE == 0) {
                s_cs += valid ? cs : 0.f;
                s_ssim += valid ? L * cs : 0.f;
            } else if (valid) {
                const float g_cs = gC + gS * L, g_L = gS * cs;
                const float dcs_dmu1 = (2.0f * mu1 * A2 - 2.0f * mu2 * B2) * (iB2 * iB2);
                const float dL_dmu1 = (2.0f * mu2 * B1 - 2.0f * mu1 * A1) * (iB1 * iB1);
                const size_t plane = (size_t)Hb * Wb, o = (size_t)nc * plane + (size_t)(ty0 + r) * Wb + (tx0 + cc);
                const size_t all = (size_t)NC * plane;
                G[o] = g_L * dL_dmu1 + g_cs * dcs_dmu1;
                G[all + o] = g_cs * (-A2 * (iB2 * iB2));
                G[2 * all + o] = g_cs * (2.0f * iB2);
            }
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { s_ssim += __shfl_xor(s_ssim, d, 64); s_cs += __shfl_xor(s_cs, d, 64); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = s_ssim; red[1][tid >> 6] = s_cs; }
        __syncthreads();
        if (tid < 2) {
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x, tiles = (size_t)gridDim.x * gridDim.y;
            partial[((size_t)nc * tiles + tile) * 2 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
        }
    }
    }   // channels
}

// means[nc][2] = (sum over the tiles, in tile order) / count
__global__ void __launch_bounds__(64)
msssim_means_kernel(const float *__restrict__ partial, const int tiles, const float inv_count, float *__restrict__ means) {
    const int nc = blockIdx.x, lane = threadIdx.x;
    float a = 0.f, b = 0.f;
    for (int t = lane; t < tiles; t += 64) {
        a += partial[((size_t)nc * tiles + t) * 2];
        b += partial[((size_t)nc * tiles + t) * 2 + 1];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if (lane == 0) { means[2 * nc] = a * inv_count; means[2 * nc + 1] = b * inv_count; }
}

// dx[p] = sum_ij w_i w_j (G0 + 2 x G1 + y G2)[p - (i, j)] + 1/4 dnext[(p + pad) / 2]; tile 32 x 32 of the level's image; the same
// register blocking as the forward filter
__global__ void __launch_bounds__(256)
msssim_back_kernel(const MsView X, const MsView Y, const MsView DX, const int C, const int H, const int W, const MsWin win,
                   const float *__restrict__ G, const float *__restrict__ dnext, const int Hn, const int Wn) {
    __shared__ __attribute__((aligned(16))) float sg[3][MS_IN][MS_SW];
    __shared__ __attribute__((aligned(16))) float th[3][MS_IN][MS_HW];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int ty0 = blockIdx.y * MS_T, tx0 = blockIdx.x * MS_T;
    const int Hb = H - (MS_WIN - 1), Wb = W - (MS_WIN - 1);
    const size_t plane = (size_t)Hb * Wb, all = (size_t)gridDim.z * C * plane;
    for (int c = 0; c < C; c++) {      // all channels of the tile in one workgroup (see msssim_maps_kernel)
    const int nc = n * C + c;
    if (c) __syncthreads();
    // filtered-domain positions q = p - 10 .. p: local (r, cc) <-> q = (ty0 - 10 + r, tx0 - 10 + cc)
    {
        constexpr int NL = (MS_IN * MS_SW + 255) / 256;
        float l0[NL], l1[NL], l2[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int idx = tid + 256 * k, r = idx / MS_SW, cc = idx - r * MS_SW, qy = ty0 - (MS_WIN - 1) + r, qx = tx0 - (MS_WIN - 1) + cc;
            const bool in = idx < MS_IN * MS_SW && cc < MS_IN && (unsigned)qy < (unsigned)Hb && (unsigned)qx < (unsigned)Wb;
            const size_t o = (size_t)nc * plane + (size_t)(in ? qy : 0) * Wb + (in ? qx : 0);
            l0[k] = in ? G[o] : 0.f;
            l1[k] = in ? G[all + o] : 0.f;
            l2[k] = in ? G[2 * all + o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int idx = tid + 256 * k, r = idx / MS_SW, cc = idx - r * MS_SW;
            if (idx < MS_IN * MS_SW) { sg[0][r][cc] = l0[k]; sg[1][r][cc] = l1[k]; sg[2][r][cc] = l2[k]; }
        }
    }
    __syncthreads();
    for (int item = tid; item < MS_IN * (MS_T / 4); item += 256) {      // along the row: p column c0 + e gathers q columns c0 + e + 10 - t
        const int r = item >> 3, c0 = (item & 7) * 4;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float gv[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 a = *(const float4 *)&sg[m][r][c0 + 4 * q];
                gv[4 * q] = a.x; gv[4 * q + 1] = a.y; gv[4 * q + 2] = a.z; gv[4 * q + 3] = a.w;
            }
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < MS_WIN; t++) a += win.w[t] * gv[e + (MS_WIN - 1) - t];
                o[e] = a;
            }
            *(float4 *)&th[m][r][c0] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    const int py = H & 1, px = W & 1;
    const int cc = tid & 31, r0 = (tid >> 5) * 4, x = tx0 + cc;
    float out[3][4];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        float tv[14];
#pragma unroll
        for (int k = 0; k < 14; k++) tv[k] = th[m][r0 + k][cc];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < MS_WIN; t++) a += win.w[t] * tv[e + (MS_WIN - 1) - t];
            out[m][e] = a;
        }
    }
    if (x < W) {
        const int vb = x / X.Wv, vbx = vb * X.Wv;
        float xv[4], yv[4], dn[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int y = ty0 + r0 + e;
            const bool in = y < H;
            xv[e] = in ? X.p[ms_at_from(X, n, c, y, x, vb, vbx)] : 0.f;
            yv[e] = in ? Y.p[ms_at_from(Y, n, c, y, x, vb, vbx)] : 0.f;
            dn[e] = (in && dnext) ? dnext[((size_t)nc * Hn + ((y + py) >> 1)) * Wn + ((x + px) >> 1)] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int y = ty0 + r0 + e;
            if (y < H) DX.p[ms_at_from(DX, n, c, y, x, vb, vbx)] = out[0][e] + 2.0f * xv[e] * out[1][e] + yv[e] * out[2][e] + 0.25f * dn[e];
        }
    }
    }   // channels
}

struct MsPlan {
    int H[MS_LEVELS], W[MS_LEVELS];
    size_t x[MS_LEVELS], y[MS_LEVELS], dx[MS_LEVELS];   // float offsets of the planar pyramid levels 1..4 (level 0: the caller's)
    size_t G, partial, total;
};
bool ms_plan(int N, int C, int H, int W, MsPlan &P) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return false;
    if ((H < W ? H : W) <= (MS_WIN - 1) * 16) return false;     // four 2x downsamplings must leave more than a window
    if ((long long)N * C * H * W >= (1ll << 31)) return false;    // 32-bit element offsets in the kernels
    size_t o = 0;
    const size_t NC = (size_t)N * C;
    P.H[0] = H; P.W[0] = W;
    for (int l = 1; l < MS_LEVELS; l++) { P.H[l] = (P.H[l - 1] + (P.H[l - 1] & 1)) / 2; P.W[l] = (P.W[l - 1] + (P.W[l - 1] & 1)) / 2; }
    P.x[0] = P.y[0] = P.dx[0] = 0;
    for (int l = 1; l < MS_LEVELS; l++) {
        const size_t pl = NC * P.H[l] * P.W[l];
        P.x[l] = o; o += pl; P.y[l] = o; o += pl; P.dx[l] = o; o += pl;
    }
    P.G = o; o += 3 * NC * (size_t)(H - 10) * (W - 10);
    const size_t tiles0 = (size_t)((W - 10 + MS_T - 1) / MS_T) * ((H - 10 + MS_T - 1) / MS_T);
    P.partial = o; o += NC * tiles0 * 2;
    P.total = o;
    return true;
}
MsView planar(float *p, int C, int H, int W) {
    MsView v;
    v.p = p; v.sN = C * H * W; v.sC = H * W; v.sY = W; v.sV = 0; v.sX = 1; v.Wv = W;
    return v;
}
// (element offsets are 32-bit in the kernels: the largest offset a view can produce must stay below 2^31)
bool view_ok(const lara_image_view *v, int N, int C, int H, int W) {
    if (!v || !v->p || v->Wv <= 0 || v->sN < 0 || v->sC < 0 || v->sY < 0 || v->sV < 0 || v->sX < 0) return false;
    const long long top = (long long)(N - 1) * v->sN + (long long)(C - 1) * v->sC + (long long)(H - 1) * v->sY +
                          (long long)((W - 1) / v->Wv) * v->sV + (long long)(v->Wv - 1) * v->sX;
    return top < (1ll << 31);
}
MsView from_c(const lara_image_view *v) {
    MsView m;
    m.p = v->p; m.sN = (int)v->sN; m.sC = (int)v->sC; m.sY = (int)v->sY; m.sV = (int)v->sV; m.sX = (int)v->sX; m.Wv = v->Wv;
    return m;
}
constexpr float MS_C1 = 0.01f * 0.01f, MS_C2 = 0.03f * 0.03f;

}  // namespace

extern "C" {

int64_t lara_ms_ssim_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W) {
    MsPlan P;
    if (!ms_plan(N, C, H, W, P)) return LARA2DGS_E_INVALID;
    return (int64_t)P.total;
}

int lara_ms_ssim_forward(int32_t N, int32_t C, int32_t H, int32_t W, const lara_image_view *X, const lara_image_view *Y,
                         const float *window11, float *means, float *workspace, void *stream) {
    MsPlan P;
    if (!ms_plan(N, C, H, W, P) || !view_ok(X, N, C, H, W) || !view_ok(Y, N, C, H, W) || !window11 || !means || !workspace)
        return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    MsWin win;
    for (int t = 0; t < MS_WIN; t++) win.w[t] = window11[t];
    const unsigned NC = (unsigned)(N * C);
    MsView xv = from_c(X), yv = from_c(Y);
    L2D_PROF("ms_ssim_fwd", s);
    for (int l = 0; l < MS_LEVELS; l++) {
        const int Hl = P.H[l], Wl = P.W[l], Hb = Hl - 10, Wb = Wl - 10;
        const dim3 grid((Wb + MS_T - 1) / MS_T, (Hb + MS_T - 1) / MS_T, (unsigned)N);
        hipLaunchKernelGGL(msssim_maps_kernel<0>, grid, dim3(256), 0, s, xv, yv, C, Hl, Wl, win, MS_C1, MS_C2, workspace + P.partial,
                           (const float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(msssim_means_kernel, dim3(NC), dim3(64), 0, s, workspace + P.partial, (int)(grid.x * grid.y),
                           1.0f / ((float)Hb * (float)Wb), means + (size_t)l * NC * 2);
        if (l + 1 < MS_LEVELS) {
            const int Ho = P.H[l + 1], Wo = P.W[l + 1];
            hipLaunchKernelGGL(msssim_pool_kernel, dim3((Wo + 63) / 64, (Ho + 3) / 4, (unsigned)N), dim3(256), 0, s, xv, yv, C, Hl, Wl, Ho, Wo,
                               workspace + P.x[l + 1], workspace + P.y[l + 1]);
            xv = planar(workspace + P.x[l + 1], C, Ho, Wo);
            yv = planar(workspace + P.y[l + 1], C, Ho, Wo);
        }
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_ms_ssim_backward(int32_t N, int32_t C, int32_t H, int32_t W, const lara_image_view *X, const lara_image_view *Y,
                          const float *window11, const float *d_means, const lara_image_view *dX, float *workspace, void *stream) {
    MsPlan P;
    if (!ms_plan(N, C, H, W, P) || !view_ok(X, N, C, H, W) || !view_ok(Y, N, C, H, W) || !view_ok(dX, N, C, H, W) || !window11 || !d_means ||
        !workspace)
        return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    MsWin win;
    for (int t = 0; t < MS_WIN; t++) win.w[t] = window11[t];
    const unsigned NC = (unsigned)(N * C);
    L2D_PROF("ms_ssim_bwd", s);
    for (int l = MS_LEVELS - 1; l >= 0; l--) {     // coarsest first: a level's gradient needs the pooled level's
        const int Hl = P.H[l], Wl = P.W[l], Hb = Hl - 10, Wb = Wl - 10;
        const MsView xv = l ? planar(workspace + P.x[l], C, Hl, Wl) : from_c(X);
        const MsView yv = l ? planar(workspace + P.y[l], C, Hl, Wl) : from_c(Y);
        const MsView dv = l ? planar(workspace + P.dx[l], C, Hl, Wl) : from_c(dX);
        hipLaunchKernelGGL(msssim_maps_kernel<1>, dim3((Wb + MS_T - 1) / MS_T, (Hb + MS_T - 1) / MS_T, (unsigned)N), dim3(256), 0, s, xv, yv, C,
                           Hl, Wl, win, MS_C1, MS_C2, (float *)nullptr, d_means + (size_t)l * NC * 2, workspace + P.G);
        const bool has_next = l + 1 < MS_LEVELS;
        hipLaunchKernelGGL(msssim_back_kernel, dim3((Wl + MS_T - 1) / MS_T, (Hl + MS_T - 1) / MS_T, (unsigned)N), dim3(256), 0, s, xv, yv, dv, C,
                           Hl, Wl, win, (const float *)(workspace + P.G), has_next ? (const float *)(workspace + P.dx[l + 1]) : (const float *)nullptr,
                           has_next ? P.H[l + 1] : 0, has_next ? P.W[l + 1] : 0);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
