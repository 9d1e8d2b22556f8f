// finedec.hip -- Decoder.forward_fine (lightning/network.py:280-284) as one kernel per direction.  See
// include/lara_finedec.h for the algebra.  One wave = 32 points; the three folded products run on the matrix cores in fp32
// with the folded weights (45 KB forward, 83 KB backward) in LDS; workgroups are persistent and walk tiles of 128 points.
#include "common.h"
#include "../../include/lara_finedec.h"

namespace {

constexpr int FD = 80, NH = 8, CD = 8, NV = 4, HID = 64, SH = 12, TQ = NH * CD;  // TQ = 64 folded query rows

// The three folded products run on the matrix cores in FP32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
// accumulation -- the fixture parity of the VALU version carries over unchanged; the part's fp32 matrix rate is ~5x what
// the 10.6 kFMA-per-point VALU kernel reached).  One wave = 32 points = the 32 columns of every product:
//     t^T   [64 x 32] = Wqk  [64 x 80] . xn^T [80 x 32]
//     hid^T [64 x 32] = W1ov [64 x 64] . u^T  [64 x 32]
//     sh^T  [12 x 32] = W2   [12 x 64] . h^T  [64 x 32]          (and their transposes in the backward)
// A 32x32x2 MFMA takes A[i = lane % 32][k = lane / 32], B[k = lane / 32][j = lane % 32] and leaves
// D[i = 8 (r / 4) + 4 (lane / 32) + r % 4][j = lane % 32] in register r of 16.  Two consequences shape the kernel:
//  * a lane owns ONE point (column j) and, of every 8 consecutive rows, 4: of head h's 8 channels lanes 0-31 hold c = 0..3
//    and lanes 32-63 c = 4..7.  The attention (4-way softmax per head) runs on the lane's own 4 channels plus one
//    exchange with lane ^ 32 per (head, view) dot product;
//  * an accumulator tile IS a B operand: step (tile, r) of the next product takes k = row(tile, r, 0) from the lower
//    half-wave and k = row(tile, r, 1) from the upper one -- register r of every lane, as it stands.  The reduction
//    index is just walked in that order, and the A operand (the weights, staged in LDS with k as the slow index) is
//    read at the matching rows.  Nothing is transposed, nothing leaves the registers between the products.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int crow(const int tile, const int r, const int hf) { return tile * 32 + (r >> 2) * 8 + hf * 4 + (r & 3); }
__device__ __forceinline__ f32x16 mfma2(const float a, const float b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// The product loops are fully unrolled (accumulator registers are addressed statically); left alone, the scheduler hoists ALL
// of a loop's LDS weight reads in front of it (350-480 VGPRs, one wave per SIMD).  A compiler-level memory fence every few
// steps bounds how far the reads run ahead, which keeps the kernels at 256 registers = two waves per SIMD.
#define FD_FENCE(i, every) do { if (((i) % (every)) == (every) - 1) asm volatile("" ::: "memory"); } while (0)

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.f;
    return z;
}

// LDS images of the folded weights (floats).  "T" = reduction index slow, output index fast (conflict-free A reads).
constexpr int L_WQKT = 0;                     // [80][64]   WqkT[f][q]  = Wqk[q][f]
constexpr int L_W1T = L_WQKT + FD * TQ;       // [64][64]   W1T[k][m]   = W1ov[m][k]
constexpr int L_W2T = L_W1T + TQ * HID;       // [64][32]   W2T[m][o]   = W2[o][m], o < 12, else 0
constexpr int L_B1 = L_W2T + HID * 32;        // [64]
constexpr int L_B2 = L_B1 + HID;              // [32]  (12 used)
constexpr int L_FWD_END = L_B2 + 32;
// backward only
constexpr int L_W2P = L_FWD_END;              // [16][64]   W2P[o][m]   = W2[o][m], o < 12, else 0
constexpr int L_W1 = L_W2P + 16 * HID;        // [64][64]   W1[m][k]    = W1ov[m][k]
constexpr int L_WQK = L_W1 + HID * TQ;        // [64][96]   WqkP[q][f]  = Wqk[q][f], f < 80, else 0
constexpr int L_BWD_END = L_WQK + TQ * 96;

__device__ __forceinline__ void stage_weights(float *sw, const float *Wqk, const float *W1ov, const float *b1, const float *W2,
                                              const float *b2, const bool bwd) {
    for (int i = threadIdx.x; i < TQ * FD; i += blockDim.x) sw[L_WQKT + (i % FD) * TQ + i / FD] = Wqk[i];
    for (int i = threadIdx.x; i < HID * TQ; i += blockDim.x) sw[L_W1T + (i % TQ) * HID + i / TQ] = W1ov[i];
    for (int i = threadIdx.x; i < HID * 32; i += blockDim.x) sw[L_W2T + i] = (i % 32) < SH ? W2[(i % 32) * HID + i / 32] : 0.f;
    for (int i = threadIdx.x; i < HID; i += blockDim.x) sw[L_B1 + i] = b1[i];
    for (int i = threadIdx.x; i < 32; i += blockDim.x) sw[L_B2 + i] = i < SH ? b2[i] : 0.f;
    if (bwd) {
        for (int i = threadIdx.x; i < 16 * HID; i += blockDim.x) sw[L_W2P + i] = i < SH * HID ? W2[i] : 0.f;
        for (int i = threadIdx.x; i < HID * TQ; i += blockDim.x) sw[L_W1 + i] = W1ov[i];
        for (int i = threadIdx.x; i < TQ * 96; i += blockDim.x) sw[L_WQK + i] = (i % 96) < FD ? Wqk[(i / 96) * FD + i % 96] : 0.f;
    }
    __syncthreads();
}

// the lane's half of its point's normalised features: xh[s] = xn[pt][40 hf + s]  (K order of the first product: step s
// takes feature s from the lower half-wave and feature 40 + s from the upper one)
__device__ __forceinline__ void load_xn_half(const float *xn_g, const int64_t pt, const int hf, float xh[40]) {
    const float4 *x4 = (const float4 *)(xn_g + pt * FD + hf * 40);
#pragma unroll
    for (int k = 0; k < 10; k++) { const float4 v = x4[k]; xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w; }
}

// t = Wqk xn: two 32-row tiles
__device__ __forceinline__ void folded_queries(const float *sw, const float xh[40], const int col, const int hf, f32x16 t[2]) {
    t[0] = zero16(); t[1] = zero16();
#pragma unroll
    for (int s = 0; s < 40; s++) {
        const float *w = sw + L_WQKT + (hf * 40 + s) * TQ + col;
        t[0] = mfma2(w[0], xh[s], t[0]);
        t[1] = mfma2(w[32], xh[s], t[1]);
        FD_FENCE(s, 4);
    }
}

// the lane's 4 channels of the 4 views' sampled features: pf[j][cl] = point_feats[j][4 hf + cl][pt]
__device__ __forceinline__ void load_pf_half(const float *pf_g, const int64_t n, const int64_t pt, const int hf, float pf[NV][4]) {
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int c = 0; c < 4; c++) pf[j][c] = pf_g[(int64_t)(j * CD + hf * 4 + c) * n + pt];
}

__device__ __forceinline__ float other_half(const float x) { return __shfl_xor(x, 32, 64); }

// softmax weights of all 8 heads from the folded queries (register r of tile mt: head 4 mt + r / 4, channel 4 hf + r % 4)
__device__ __forceinline__ void attention_weights(const f32x16 t[2], const float pf[NV][4], float p[NH][NV]) {
#pragma unroll
    for (int h = 0; h < NH; h++) {
        float sc[NV], m = -1e30f;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) a += t[h >> 2][(h & 3) * 4 + c] * pf[j][c];
            a += other_half(a);
            sc[j] = a;
            m = fmaxf(m, a);
        }
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < NV; j++) { p[h][j] = __expf(sc[j] - m); z += p[h][j]; }
        const float rz = 1.0f / z;
#pragma unroll
        for (int j = 0; j < NV; j++) p[h][j] *= rz;
    }
}

__device__ __forceinline__ void attend(const float p[NH][NV], const float pf[NV][4], f32x16 u[2]) {
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            u[h >> 2][(h & 3) * 4 + c] = p[h][0] * pf[0][c] + p[h][1] * pf[1][c] + p[h][2] * pf[2][c] + p[h][3] * pf[3][c];
}

// pre-activation hidden units: W1ov u + b1 (two tiles)
__device__ __forceinline__ void hidden_pre(const float *sw, const f32x16 u[2], const int col, const int hf, f32x16 hid[2]) {
    hid[0] = zero16(); hid[1] = zero16();
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float *w = sw + L_W1T + crow(mt, r, hf) * HID + col;
            hid[0] = mfma2(w[0], u[mt][r], hid[0]);
            hid[1] = mfma2(w[32], u[mt][r], hid[1]);
            FD_FENCE(r, 4);
        }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) hid[mt][r] += sw[L_B1 + crow(mt, r, hf)];
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
fine_decoder_fwd_kernel(const int n, const float *__restrict__ xn_g, const float *__restrict__ pf_g,
                        const float *__restrict__ Wqk, const float *__restrict__ W1ov, const float *__restrict__ b1,
                        const float *__restrict__ W2, const float *__restrict__ b2, float *__restrict__ sh_g) {
    __shared__ __attribute__((aligned(16))) float sw[L_FWD_END];
    stage_weights(sw, Wqk, W1ov, b1, W2, b2, false);
    const int lane = threadIdx.x & 63, col = lane & 31, hf = lane >> 5, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 128; base < n; base += (int64_t)gridDim.x * 128) {
        const int64_t pt = base + wave * 32 + col;
        const bool live = pt < n;
        const int64_t ptc = live ? pt : (int64_t)n - 1;        // (rows beyond n recompute the last point; nothing is stored)
        float xh[40];
        load_xn_half(xn_g, ptc, hf, xh);
        f32x16 tu[2];
        folded_queries(sw, xh, col, hf, tu);
        {
            float pf[NV][4], p[NH][NV];
            load_pf_half(pf_g, n, ptc, hf, pf);
            attention_weights(tu, pf, p);
            attend(p, pf, tu);
        }
        f32x16 hid[2];
        hidden_pre(sw, tu, col, hf, hid);
        f32x16 out = zero16();
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                out = mfma2(sw[L_W2T + crow(mt, r, hf) * 32 + col], fmaxf(hid[mt][r], 0.f), out);
                FD_FENCE(r, 8);
            }
        if (live) {     // rows 4 hf + (0..3) in registers 0..3, rows 8 + 4 hf + (0..3) in registers 4..7 (hf 0: 8..11)
            float4 *o4 = (float4 *)(sh_g + pt * SH);
            const float *bb = sw + L_B2;
            o4[hf] = make_float4(out[0] + bb[4 * hf], out[1] + bb[4 * hf + 1], out[2] + bb[4 * hf + 2], out[3] + bb[4 * hf + 3]);
            if (hf == 0) o4[2] = make_float4(out[4] + bb[8], out[5] + bb[9], out[6] + bb[10], out[7] + bb[11]);
        }
    }
}

// Backward: the forward recomputed up to the hidden units, then the chain backwards through the same register layouts.
// U, H, DH, DT ([n, 64] each) are written for the weight gradients, which are GEMMs over the point axis (the caller's).
__global__ void __launch_bounds__(256)
fine_decoder_bwd_kernel(const int n, const float *__restrict__ xn_g, const float *__restrict__ pf_g,
                        const float *__restrict__ Wqk, const float *__restrict__ W1ov, const float *__restrict__ b1,
                        const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ dsh_g,
                        float *__restrict__ dxn_g, float *__restrict__ dpf_g, float *__restrict__ U_g,
                        float *__restrict__ H_g, float *__restrict__ DH_g, float *__restrict__ DT_g) {
    extern __shared__ __attribute__((aligned(16))) float sw[];
    stage_weights(sw, Wqk, W1ov, b1, W2, b2, true);
    const int lane = threadIdx.x & 63, col = lane & 31, hf = lane >> 5, wave = threadIdx.x >> 6;
    // a [n, 64] row as the accumulator layout holds it: registers 4 q .. 4 q + 3 of tile mt = columns 32 mt + 8 q + 4 hf + (0..3)
    auto store64 = [&](float *g, const int64_t pt, const f32x16 v[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *(float4 *)(g + pt * 64 + mt * 32 + q * 8 + hf * 4) = make_float4(v[mt][4 * q], v[mt][4 * q + 1], v[mt][4 * q + 2], v[mt][4 * q + 3]);
    };
    for (int64_t base = (int64_t)blockIdx.x * 128; base < n; base += (int64_t)gridDim.x * 128) {
        const int64_t pt = base + wave * 32 + col;
        const bool live = pt < n;
        const int64_t ptc = live ? pt : (int64_t)n - 1;
        f32x16 t[2], u[2];
        {
            float xh[40];
            load_xn_half(xn_g, ptc, hf, xh);
            folded_queries(sw, xh, col, hf, t);
        }
        float pf[NV][4], p[NH][NV];
        load_pf_half(pf_g, n, ptc, hf, pf);
        attention_weights(t, pf, p);
        attend(p, pf, u);
        if (live) store64(U_g, pt, u);
        asm volatile("" ::: "memory");
        f32x16 hid[2], dh[2];
        hidden_pre(sw, u, col, hf, hid);
        // dL/dhid = W2^T dL/dsh: K = the 12 outputs (padded to 16): step s takes output s from the lower half-wave, 8 + s from the upper
        {
            float ds[8];
            const float4 *d4 = (const float4 *)(dsh_g + ptc * SH);
            if (hf == 0) {
                const float4 a = d4[0], b = d4[1];
                ds[0] = a.x; ds[1] = a.y; ds[2] = a.z; ds[3] = a.w; ds[4] = b.x; ds[5] = b.y; ds[6] = b.z; ds[7] = b.w;
            } else {
                const float4 c = d4[2];
                ds[0] = c.x; ds[1] = c.y; ds[2] = c.z; ds[3] = c.w; ds[4] = ds[5] = ds[6] = ds[7] = 0.f;
            }
            dh[0] = zero16(); dh[1] = zero16();
#pragma unroll
            for (int s_ = 0; s_ < 8; s_++) {
                const float *w = sw + L_W2P + (hf * 8 + s_) * HID + col;
                dh[0] = mfma2(w[0], ds[s_], dh[0]);
                dh[1] = mfma2(w[32], ds[s_], dh[1]);
                FD_FENCE(s_, 4);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const bool on = hid[mt][r] > 0.f;
                hid[mt][r] = on ? hid[mt][r] : 0.f;
                dh[mt][r] = on ? dh[mt][r] : 0.f;
            }
        if (live) { store64(H_g, pt, hid); store64(DH_g, pt, dh); }
        asm volatile("" ::: "memory");
        // dL/du = W1ov^T dL/dhid  (A = W1ov[m][k] with m walked in accumulator order)
        f32x16 du[2];
        du[0] = zero16(); du[1] = zero16();
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float *w = sw + L_W1 + crow(mt, r, hf) * TQ + col;
                du[0] = mfma2(w[0], dh[mt][r], du[0]);
                du[1] = mfma2(w[32], dh[mt][r], du[1]);
                FD_FENCE(r, 4);
            }
        asm volatile("" ::: "memory");
        // the attention's backward on the lane's own 4 channels: du -> dt in place, dpf accumulated over the heads
        float dpf[NV][4];
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int c = 0; c < 4; c++) dpf[j][c] = 0.f;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int mt = h >> 2, r0 = (h & 3) * 4;
            float dp[NV], dot = 0.f;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) a += du[mt][r0 + c] * pf[j][c];
                a += other_half(a);
                dp[j] = a;
                dot += p[h][j] * a;
            }
            float dt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const float dsc = p[h][j] * (dp[j] - dot);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    dt[c] += dsc * pf[j][c];
                    dpf[j][c] += p[h][j] * du[mt][r0 + c] + dsc * t[mt][r0 + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) du[mt][r0 + c] = dt[c];
            asm volatile("" ::: "memory");
        }
        if (live) {
            store64(DT_g, pt, du);
#pragma unroll
            for (int j = 0; j < NV; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) dpf_g[(int64_t)(j * CD + hf * 4 + c) * n + pt] = dpf[j][c];
        }
        // dL/dxn = Wqk^T dL/dt: 80 rows = three tiles (the last one half empty)
#pragma unroll
        for (int ft = 0; ft < 3; ft++) {
            f32x16 dx = zero16();
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    dx = mfma2(sw[L_WQK + crow(mt, r, hf) * 96 + ft * 32 + col], du[mt][r], dx);
                    FD_FENCE(r, 8);
                }
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int f0 = ft * 32 + q * 8 + hf * 4;
                    if (f0 < FD) *(float4 *)(dxn_g + pt * FD + f0) = make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
                }
            }
        }
    }
}


// ---- LayerNorm over rows of 80 features, one thread per row ------------------------------------------------------
__global__ void __launch_bounds__(256)
fine_ln_fwd_kernel(const int n, const float *__restrict__ x_g, const float *__restrict__ gamma, const float *__restrict__ beta,
                   const float eps, float *__restrict__ xn_g, float2 *__restrict__ stats) {
    __shared__ float sg[FD], sb[FD];
    if (threadIdx.x < FD) { sg[threadIdx.x] = gamma[threadIdx.x]; sb[threadIdx.x] = beta[threadIdx.x]; }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 *x4 = (const float4 *)(x_g + i * FD);
    float4 v[FD / 4];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < FD / 4; k++) { v[k] = x4[k]; sum += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
    const float mean = sum * (1.0f / FD);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < FD / 4; k++) {
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
        var += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(var * (1.0f / FD) + eps);
    float4 *o4 = (float4 *)(xn_g + i * FD);
#pragma unroll
    for (int k = 0; k < FD / 4; k++) {
        float4 o;
        o.x = (v[k].x - mean) * rstd * sg[4 * k] + sb[4 * k];
        o.y = (v[k].y - mean) * rstd * sg[4 * k + 1] + sb[4 * k + 1];
        o.z = (v[k].z - mean) * rstd * sg[4 * k + 2] + sb[4 * k + 2];
        o.w = (v[k].w - mean) * rstd * sg[4 * k + 3] + sb[4 * k + 3];
        o4[k] = o;
    }
    stats[i] = make_float2(mean, rstd);
}

// wave-wide sum in six DPP adds; the total lands in lane 63 (the shuffle-based butterfly is six LDS permutes per value:
// with 160 values per wave the kernel was bound by them)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float x) {
    x += dpp_term<0xB1, 0xf>(x);    // quad_perm [1,0,3,2]
    x += dpp_term<0x4E, 0xf>(x);    // quad_perm [2,3,0,1]
    x += dpp_term<0x141, 0xf>(x);   // row_half_mirror
    x += dpp_term<0x140, 0xf>(x);   // row_mirror: every lane of a 16-lane row holds the row's sum
    x += dpp_term<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
    x += dpp_term<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}

__global__ void __launch_bounds__(256)
fine_ln_bwd_kernel(const int n, const float *__restrict__ x_g, const float *__restrict__ gamma, const float2 *__restrict__ stats,
                   const float *__restrict__ dxn_g, float *__restrict__ dx_g, float *__restrict__ partials) {
    __shared__ float sg[FD];
    __shared__ float part[4][2 * FD];
    if (threadIdx.x < FD) sg[threadIdx.x] = gamma[threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 xh[FD / 4], g[FD / 4];
    float2 st = make_float2(0.f, 0.f);
    if (on) st = stats[i];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < FD / 4; k++) {
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv;
        if (on) { xv = ((const float4 *)(x_g + i * FD))[k]; dv = ((const float4 *)(dxn_g + i * FD))[k]; }
        xh[k] = make_float4((xv.x - st.x) * st.y, (xv.y - st.x) * st.y, (xv.z - st.x) * st.y, (xv.w - st.x) * st.y);
        // the workgroup's sums of d_xn * xhat and d_xn for d_gamma / d_beta (rows beyond n contribute zeros)
        const float pg[4] = {dv.x * xh[k].x, dv.y * xh[k].y, dv.z * xh[k].z, dv.w * xh[k].w};
        const float pb[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float a = wave_sum_to_lane63(pg[q]), b = wave_sum_to_lane63(pb[q]);
            if (lane == 63) { part[wave][4 * k + q] = a; part[wave][FD + 4 * k + q] = b; }
        }
        g[k] = make_float4(dv.x * sg[4 * k], dv.y * sg[4 * k + 1], dv.z * sg[4 * k + 2], dv.w * sg[4 * k + 3]);
        s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
        s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
    }
    if (on) {
        const float m1 = s1 * (1.0f / FD), m2 = s2 * (1.0f / FD);
        float4 *o4 = (float4 *)(dx_g + i * FD);
#pragma unroll
        for (int k = 0; k < FD / 4; k++)
            o4[k] = make_float4(st.y * (g[k].x - m1 - xh[k].x * m2), st.y * (g[k].y - m1 - xh[k].y * m2),
                                st.y * (g[k].z - m1 - xh[k].z * m2), st.y * (g[k].w - m1 - xh[k].w * m2));
    }
    __syncthreads();
    if (threadIdx.x < 2 * FD)
        partials[(size_t)blockIdx.x * 2 * FD + threadIdx.x] =
            ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}

// ---- the five parameter gradients of the folded decoder: products of the factor arrays over the point axis ------------------
//   dWqk [64,80] = DT^T xn,  dW1ov [64,64] = DH^T U,  db1 [64] = colsum(DH),  dW2 [12,64] = d_sh^T HID,  db2 [12] = colsum(d_sh)
// (rounds 2-4 ran them through the BLAS library as batched GEMMs over 1024-row slabs + torch reductions: 79 launches and 1.1 ms
// per step of someone else's kernels on a path that claims its own).  Reductions over n = 10^5..10^6 rows with 12..80 columns on
// either side are streams, not GEMMs: a workgroup of three waves takes a slab of FW_SLAB rows -- waves 0 / 1 the two column halves
// of dWqk, wave 2 the rest -- on
// v_mfma_f32_32x32x2_f32 (exact fp32 products and sums) with the operands straight from global memory: the instruction wants
// A[i = lane % 32][k = lane / 32] = X[row + lane / 32][i], so a half-wave reads ONE row, and since the order of the rows / columns
// of an outer product is free, lane c fetches 2 or 4 CONSECUTIVE columns with one 8- / 16-byte load and feeds component e to the
// e-th MFMA block (block e then holds columns {2c + e} / {4c + e}): whole rows per instruction instead of 128-byte pieces
// (the first version, one dword per lane and block, streamed at 2.1 TB/s).  A second launch adds the slabs in a fixed order (no
// atomics: reproducible).
constexpr int FW_SLAB = 512;
constexpr int FW_OUT = 64 * 80 + 64 * 64 + 64 + 12 * 64 + 12;     // floats per slab: dWqk | dW1ov | db1 | dW2 | db2
constexpr int FW_O_W1 = 64 * 80, FW_O_B1 = FW_O_W1 + 64 * 64, FW_O_W2 = FW_O_B1 + 64, FW_O_B2 = FW_O_W2 + 12 * 64;

// acc[ea][eb] holds out[AV * rA + ea][BV * cB + eb] with rA = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), cB = lane & 31
template <int AV, int BV>
__device__ __forceinline__ void fw_store(float *__restrict__ out, const int ldo, const int i_rows, const int j_cols, const int lane,
                                         const f32x16 (&acc)[AV][BV]) {
    const int cB = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ea = 0; ea < AV; ea++)
#pragma unroll
        for (int eb = 0; eb < BV; eb++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = AV * ((e & 3) + 8 * (e >> 2) + 4 * kh) + ea, j = BV * cB + eb;
                if (i < i_rows && j < j_cols) out[(size_t)i * ldo + j] = acc[ea][eb][e];
            }
}

__global__ void __launch_bounds__(192)
fine_wgrad_kernel(const int n, const float *__restrict__ xn, const float *__restrict__ U, const float *__restrict__ HID_,
                  const float *__restrict__ DH, const float *__restrict__ DT, const float *__restrict__ d_sh,
                  float *__restrict__ part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * FW_SLAB, rows = min(FW_SLAB, n - row0);
    float *out = part + (size_t)blockIdx.x * FW_OUT;
    const int kk = lane >> 5, c = lane & 31;
    constexpr int UN = 8;      // K steps (of two rows) whose loads are in flight together
    if (wave < 2) {            // dWqk: A = DT (64 columns: float2 per lane), B = xn columns {4c + 2 wave, + 1} (80 columns: lanes c < 20)
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[a][b][e] = 0.f;
        const bool b_ok = c < FD / 4;
        for (int r = 0; r < rows; r += 2 * UN) {
            float2 av[UN], bv[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int rr = r + 2 * u + kk;
                const bool in = rr < rows;
                av[u] = in ? *(const float2 *)(DT + (size_t)(row0 + rr) * 64 + 2 * c) : make_float2(0.f, 0.f);
                bv[u] = (in && b_ok) ? *(const float2 *)(xn + (size_t)(row0 + rr) * FD + 4 * c + 2 * wave) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                acc[0][0] = mfma2(av[u].x, bv[u].x, acc[0][0]);
                acc[0][1] = mfma2(av[u].x, bv[u].y, acc[0][1]);
                acc[1][0] = mfma2(av[u].y, bv[u].x, acc[1][0]);
                acc[1][1] = mfma2(av[u].y, bv[u].y, acc[1][1]);
            }
        }
        // acc[ea][eb] = dWqk[2 rA + ea][4 cB + 2 wave + eb]
        const int kh = lane >> 5;
#pragma unroll
        for (int ea = 0; ea < 2; ea++)
#pragma unroll
            for (int eb = 0; eb < 2; eb++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int i = 2 * ((e & 3) + 8 * (e >> 2) + 4 * kh) + ea, j = 4 * c + 2 * wave + eb;
                    if (j < FD) out[(size_t)i * FD + j] = acc[ea][eb][e];
                }
    } else {                   // dW1ov + db1 (A = DH, B = U: float2 each) and dW2 + db2 (A = d_sh: one float, lanes c < 12; B = HID: float2)
        f32x16 acc1[2][2], acc2[1][2];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) { acc1[0][b][e] = 0.f; acc1[1][b][e] = 0.f; acc2[0][b][e] = 0.f; }
        float cs1x = 0.f, cs1y = 0.f, cs2 = 0.f;
        const bool s_ok = c < SH;
        for (int r = 0; r < rows; r += 2 * UN) {
            float2 dh[UN], uu[UN], hh[UN];
            float ds[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int rr = r + 2 * u + kk;
                const bool in = rr < rows;
                const size_t row = (size_t)(row0 + rr);
                dh[u] = in ? *(const float2 *)(DH + row * HID + 2 * c) : make_float2(0.f, 0.f);
                uu[u] = in ? *(const float2 *)(U + row * 64 + 2 * c) : make_float2(0.f, 0.f);
                hh[u] = in ? *(const float2 *)(HID_ + row * HID + 2 * c) : make_float2(0.f, 0.f);
                ds[u] = (in && s_ok) ? d_sh[row * SH + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                cs1x += dh[u].x; cs1y += dh[u].y; cs2 += ds[u];
                acc1[0][0] = mfma2(dh[u].x, uu[u].x, acc1[0][0]);
                acc1[0][1] = mfma2(dh[u].x, uu[u].y, acc1[0][1]);
                acc1[1][0] = mfma2(dh[u].y, uu[u].x, acc1[1][0]);
                acc1[1][1] = mfma2(dh[u].y, uu[u].y, acc1[1][1]);
                acc2[0][0] = mfma2(ds[u], hh[u].x, acc2[0][0]);
                acc2[0][1] = mfma2(ds[u], hh[u].y, acc2[0][1]);
            }
        }
        fw_store<2, 2>(out + FW_O_W1, 64, 64, 64, lane, acc1);
        fw_store<1, 2>(out + FW_O_W2, HID, SH, HID, lane, acc2);
        cs1x += __shfl_xor(cs1x, 32, 64); cs1y += __shfl_xor(cs1y, 32, 64); cs2 += __shfl_xor(cs2, 32, 64);
        if (lane < 32) { out[FW_O_B1 + 2 * c] = cs1x; out[FW_O_B1 + 2 * c + 1] = cs1y; }
        if (lane < SH) out[FW_O_B2 + lane] = cs2;
    }
}

// out[k] = sum over the slabs, in slab order, four independent chains per thread
// (64 outputs x 4 slab phases per workgroup: a thread walks every fourth slab -- a quarter of the dependent chain -- and the four
// phases of an output are added in phase order through LDS)
__global__ void __launch_bounds__(256)
fine_wgrad_reduce_kernel(const float *__restrict__ part, const int slabs, float *__restrict__ out) {
    __shared__ float ph[4][64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f;
    if (k < FW_OUT) {
        int s = q;
        for (; s + 4 < slabs; s += 8) {
            a0 += part[(size_t)s * FW_OUT + k];
            a1 += part[(size_t)(s + 4) * FW_OUT + k];
        }
        if (s < slabs) a0 += part[(size_t)s * FW_OUT + k];
    }
    ph[q][threadIdx.x & 63] = a0 + a1;
    __syncthreads();
    if (q == 0 && k < FW_OUT) out[k] = (ph[0][threadIdx.x] + ph[1][threadIdx.x]) + (ph[2][threadIdx.x] + ph[3][threadIdx.x]);
}

unsigned fd_grid(int n) {
    const unsigned tiles = (unsigned)((n + 127) / 128);
    return tiles < 1024u ? tiles : 1024u;
}

}  // namespace

extern "C" {

int lara_fine_decoder_forward(int32_t n, const float *xn, const float *pf, const float *Wqk, const float *W1ov,
                              const float *b1, const float *W2, const float *b2, float *sh, void *stream) {
    if (n < 0) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!xn || !pf || !Wqk || !W1ov || !b1 || !W2 || !b2 || !sh) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("fine_decoder_fwd", s);
        hipLaunchKernelGGL(fine_decoder_fwd_kernel, dim3(fd_grid(n)), dim3(256), 0, s, n, xn, pf, Wqk, W1ov, b1, W2, b2, sh);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_fine_decoder_backward(int32_t n, const float *xn, const float *pf, const float *Wqk, const float *W1ov,
                               const float *b1, const float *W2, const float *b2, const float *d_sh, float *d_xn,
                               float *d_pf, float *U, float *HID_, float *DH, float *DT, void *stream) {
    if (n < 0) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!xn || !pf || !Wqk || !W1ov || !b1 || !W2 || !b2 || !d_sh || !d_xn || !d_pf || !U || !HID_ || !DH || !DT)
        return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("fine_decoder_bwd", s);
        static bool lds_set = false;
        if (!lds_set) {     // 83 KB of folded weights (both orientations): beyond the 64 KB a kernel gets without asking
            if (hipFuncSetAttribute((const void *)fine_decoder_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    L_BWD_END * 4) != hipSuccess) return LARA2DGS_E_LAUNCH;
            lds_set = true;
        }
        hipLaunchKernelGGL(fine_decoder_bwd_kernel, dim3(fd_grid(n) < 512u ? fd_grid(n) : 512u), dim3(256), L_BWD_END * 4, s, n, xn, pf,
                           Wqk, W1ov, b1, W2, b2, d_sh, d_xn, d_pf, U, HID_, DH, DT);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int32_t lara_fine_wgrad_floats(void) { return FW_OUT; }
int64_t lara_fine_wgrad_workspace_bytes(int32_t n) { return n <= 0 ? 0 : (int64_t)((n + FW_SLAB - 1) / FW_SLAB) * FW_OUT * 4; }

int lara_fine_decoder_wgrad(int32_t n, const float *xn, const float *U, const float *HID_, const float *DH, const float *DT,
                            const float *d_sh, float *out, void *workspace, void *stream) {
    if (n < 0 || !out) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)FW_OUT * 4, s);
        if (e != hipSuccess) { l2d_set_hip_error(e); return LARA2DGS_E_LAUNCH; }
        return LARA2DGS_OK;
    }
    if (!xn || !U || !HID_ || !DH || !DT || !d_sh || !workspace) return LARA2DGS_E_INVALID;
    const int slabs = (n + FW_SLAB - 1) / FW_SLAB;
    {
        L2D_PROF("fine_decoder_wgrad", s);
        hipLaunchKernelGGL(fine_wgrad_kernel, dim3((unsigned)slabs), dim3(192), 0, s, n, xn, U, HID_, DH, DT, d_sh, (float *)workspace);
        hipLaunchKernelGGL(fine_wgrad_reduce_kernel, dim3((FW_OUT + 63) / 64), dim3(256), 0, s, (const float *)workspace, slabs, out);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int32_t lara_fine_ln_blocks(int32_t n) { return n <= 0 ? 0 : (n + 255) / 256; }

int lara_fine_ln_forward(int32_t n, const float *x, const float *gamma, const float *beta, float eps, float *xn,
                         float *stats, void *stream) {
    if (n < 0) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!x || !gamma || !beta || !xn || !stats) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("fine_ln_fwd", s);
        hipLaunchKernelGGL(fine_ln_fwd_kernel, dim3((unsigned)lara_fine_ln_blocks(n)), dim3(256), 0, s, n, x, gamma, beta, eps, xn,
                           (float2 *)stats);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_fine_ln_backward(int32_t n, const float *x, const float *gamma, const float *stats, const float *d_xn,
                          float *d_x, float *partials, void *stream) {
    if (n < 0) return LARA2DGS_E_INVALID;
    if (n == 0) return LARA2DGS_OK;
    if (!x || !gamma || !stats || !d_xn || !d_x || !partials) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("fine_ln_bwd", s);
        hipLaunchKernelGGL(fine_ln_bwd_kernel, dim3((unsigned)lara_fine_ln_blocks(n)), dim3(256), 0, s, n, x, gamma,
                           (const float2 *)stats, d_xn, d_x, partials);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
