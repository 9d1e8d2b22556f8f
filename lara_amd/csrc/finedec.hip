// finedec.hip -- Decoder.forward_fine (lightning/network.py:280-284) as one kernel per direction.  See
// include/lara_finedec.h for the algebra.  thread = point; the folded weights (40 KB) live in LDS and are read by
// broadcast (every lane the same address); workgroups are persistent and walk tiles of 256 points.
#include "common.h"
#include "../../include/lara_finedec.h"

namespace {

constexpr int FD = 80, NH = 8, CD = 8, NV = 4, HID = 64, SH = 12, TQ = NH * CD;  // TQ = 64 folded query rows
constexpr int W_QK = 0, W_1 = W_QK + TQ * FD, W_B1 = W_1 + HID * TQ, W_2 = W_B1 + HID, W_B2 = W_2 + SH * HID,
              W_END = W_B2 + 16;  // floats of LDS

__device__ __forceinline__ void load_weights(float *sw, const float *Wqk, const float *W1ov, const float *b1,
                                             const float *W2, const float *b2) {
    for (int i = threadIdx.x; i < TQ * FD; i += blockDim.x) sw[W_QK + i] = Wqk[i];
    for (int i = threadIdx.x; i < HID * TQ; i += blockDim.x) sw[W_1 + i] = W1ov[i];
    for (int i = threadIdx.x; i < HID; i += blockDim.x) sw[W_B1 + i] = b1[i];
    for (int i = threadIdx.x; i < SH * HID; i += blockDim.x) sw[W_2 + (i % HID) * SH + i / HID] = W2[i];   // transposed: [HID][12]
    for (int i = threadIdx.x; i < 16; i += blockDim.x) sw[W_B2 + i] = i < SH ? b2[i] : 0.f;
    __syncthreads();
}

// phase 1: t = Wqk xn, four input features at a time (only 4 of the point's 80 features are live at once)
__device__ __forceinline__ void folded_queries(const float *sw, const float *xn_g, const int64_t i, float t[TQ]) {
#pragma unroll
    for (int r = 0; r < TQ; r++) t[r] = 0.f;
    const float4 *x4 = (const float4 *)(xn_g + i * FD);
#pragma unroll 2
    for (int k = 0; k < FD / 4; k++) {
        const float4 x = x4[k];
        const float4 *w = (const float4 *)(sw + W_QK) + k;
#pragma unroll
        for (int r = 0; r < TQ; r++) {
            const float4 q = w[r * (FD / 4)];
            t[r] += (q.x * x.x + q.y * x.y) + (q.z * x.z + q.w * x.w);
        }
    }
}

__device__ __forceinline__ void load_pf(const float *pf_g, const int64_t n, const int64_t i, float pf[NV][CD]) {
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int c = 0; c < CD; c++) pf[j][c] = pf_g[(int64_t)(j * CD + c) * n + i];
}

// one head's 4-way softmax from its folded query t (8 values)
__device__ __forceinline__ void head_softmax(const float *t, const float pf[NV][CD], float p[NV]) {
    float s[NV], m = -1e30f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CD; c++) a += t[c] * pf[j][c];
        s[j] = a;
        m = fmaxf(m, a);
    }
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) { p[j] = __expf(s[j] - m); z += p[j]; }
    const float rz = 1.0f / z;
#pragma unroll
    for (int j = 0; j < NV; j++) p[j] *= rz;
}

__global__ void __launch_bounds__(256)
fine_decoder_fwd_kernel(const int n, const float *__restrict__ xn_g, const float *__restrict__ pf_g,
                        const float *__restrict__ Wqk, const float *__restrict__ W1ov, const float *__restrict__ b1,
                        const float *__restrict__ W2, const float *__restrict__ b2, float *__restrict__ sh_g) {
    __shared__ __attribute__((aligned(16))) float sw[W_END];
    load_weights(sw, Wqk, W1ov, b1, W2, b2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float tu[TQ];   // the folded queries t, then (head by head, in place) the attended features u
        folded_queries(sw, xn_g, i, tu);
        {
            float pf[NV][CD];
            load_pf(pf_g, n, i, pf);
#pragma unroll
            for (int h = 0; h < NH; h++) {
                float p[NV];
                head_softmax(tu + h * CD, pf, p);
#pragma unroll
                for (int c = 0; c < CD; c++) tu[h * CD + c] = p[0] * pf[0][c] + p[1] * pf[1][c] + p[2] * pf[2][c] + p[3] * pf[3][c];
            }
        }
        // phase 2: one hidden unit at a time, consumed at once by the output layer (hid is never an array)
        float out[SH];
#pragma unroll
        for (int m = 0; m < SH; m++) out[m] = sw[W_B2 + m];
#pragma unroll 2
        for (int o = 0; o < HID; o++) {
            const float4 *w1 = (const float4 *)(sw + W_1 + o * TQ);
            float a0 = sw[W_B1 + o], a1 = 0.f;
#pragma unroll
            for (int k = 0; k < TQ / 4; k++) {
                const float4 w = w1[k];
                a0 += w.x * tu[4 * k] + w.z * tu[4 * k + 2];
                a1 += w.y * tu[4 * k + 1] + w.w * tu[4 * k + 3];
            }
            const float hv = fmaxf(a0 + a1, 0.f);
            const float4 *w2 = (const float4 *)(sw + W_2 + o * 12);   // W2 is staged transposed: [HID][12]
            const float4 wa = w2[0], wb = w2[1], wc = w2[2];
            out[0] += wa.x * hv; out[1] += wa.y * hv; out[2] += wa.z * hv; out[3] += wa.w * hv;
            out[4] += wb.x * hv; out[5] += wb.y * hv; out[6] += wb.z * hv; out[7] += wb.w * hv;
            out[8] += wc.x * hv; out[9] += wc.y * hv; out[10] += wc.z * hv; out[11] += wc.w * hv;
        }
        float4 *o4 = (float4 *)(sh_g + i * SH);
        o4[0] = make_float4(out[0], out[1], out[2], out[3]);
        o4[1] = make_float4(out[4], out[5], out[6], out[7]);
        o4[2] = make_float4(out[8], out[9], out[10], out[11]);
    }
}

// Backward.  Pass A recomputes the forward (t parked in the DT array, u and relu(hid) written out: the weight
// gradients are GEMMs of them) and, hidden unit by hidden unit, forms dL/du; pass B is the attention's backward
// head by head (dL/dt replaces t in DT); pass C maps dL/dt back through Wqk, four features at a time.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
fine_decoder_bwd_kernel(const int n, const float *__restrict__ xn_g, const float *__restrict__ pf_g,
                        const float *__restrict__ Wqk, const float *__restrict__ W1ov, const float *__restrict__ b1,
                        const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ dsh_g,
                        float *__restrict__ dxn_g, float *__restrict__ dpf_g, float *__restrict__ U_g,
                        float *__restrict__ H_g, float *__restrict__ DH_g, float *__restrict__ DT_g) {
    __shared__ __attribute__((aligned(16))) float sw[W_END];
    load_weights(sw, Wqk, W1ov, b1, W2, b2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float u[TQ];
        folded_queries(sw, xn_g, i, u);
        {
            float4 *tq = (float4 *)(DT_g + i * TQ);
#pragma unroll
            for (int k = 0; k < TQ / 4; k++) tq[k] = make_float4(u[4 * k], u[4 * k + 1], u[4 * k + 2], u[4 * k + 3]);
        }
        {
            float pf[NV][CD];   // (loaded again for pass B: 32 registers less across the hidden-unit loop)
            load_pf(pf_g, n, i, pf);
#pragma unroll
            for (int h = 0; h < NH; h++) {
                float p[NV];
                head_softmax(u + h * CD, pf, p);
#pragma unroll
                for (int c = 0; c < CD; c++) u[h * CD + c] = p[0] * pf[0][c] + p[1] * pf[1][c] + p[2] * pf[2][c] + p[3] * pf[3][c];
            }
        }
        {
            float4 *uq = (float4 *)(U_g + i * TQ);
#pragma unroll
            for (int k = 0; k < TQ / 4; k++) uq[k] = make_float4(u[4 * k], u[4 * k + 1], u[4 * k + 2], u[4 * k + 3]);
        }
        float dsh[SH];
        {
            const float4 *d4 = (const float4 *)(dsh_g + i * SH);
            const float4 a = d4[0], b = d4[1], c = d4[2];
            dsh[0] = a.x; dsh[1] = a.y; dsh[2] = a.z; dsh[3] = a.w; dsh[4] = b.x; dsh[5] = b.y; dsh[6] = b.z; dsh[7] = b.w;
            dsh[8] = c.x; dsh[9] = c.y; dsh[10] = c.z; dsh[11] = c.w;
        }
#pragma unroll 2
        for (int o = 0; o < HID; o++) {
            const float4 *w1 = (const float4 *)(sw + W_1 + o * TQ);
            float a0 = sw[W_B1 + o], a1 = 0.f;
#pragma unroll
            for (int k = 0; k < TQ / 4; k++) {
                const float4 w = w1[k];
                a0 += w.x * u[4 * k] + w.z * u[4 * k + 2];
                a1 += w.y * u[4 * k + 1] + w.w * u[4 * k + 3];
            }
            const float pre = a0 + a1;
            const float4 *w2 = (const float4 *)(sw + W_2 + o * 12);
            const float4 wa = w2[0], wb = w2[1], wc = w2[2];
            const float g = (wa.x * dsh[0] + wa.y * dsh[1] + wa.z * dsh[2] + wa.w * dsh[3]) +
                            (wb.x * dsh[4] + wb.y * dsh[5] + wb.z * dsh[6] + wb.w * dsh[7]) +
                            (wc.x * dsh[8] + wc.y * dsh[9] + wc.z * dsh[10] + wc.w * dsh[11]);
            const bool on = pre > 0.f;
            const float hv = on ? pre : 0.f, dv = on ? g : 0.f;
            H_g[i * HID + o] = hv;
            DH_g[i * HID + o] = dv;
        }
        // dL/du = W1ov^T dL/dhid, in a loop of its own (u is dead by now: 64 registers less than doing it above)
        float du[TQ];
#pragma unroll
        for (int k = 0; k < TQ; k++) du[k] = 0.f;
#pragma unroll 2
        for (int o = 0; o < HID; o++) {
            const float dv = DH_g[i * HID + o];
            const float4 *w1 = (const float4 *)(sw + W_1 + o * TQ);
#pragma unroll
            for (int k = 0; k < TQ / 4; k++) {
                const float4 w = w1[k];
                du[4 * k] += w.x * dv; du[4 * k + 1] += w.y * dv; du[4 * k + 2] += w.z * dv; du[4 * k + 3] += w.w * dv;
            }
        }
        // pass B: du -> dt in place, dpf accumulated over the heads
        float pf[NV][CD], dpf[NV][CD];
        load_pf(pf_g, n, i, pf);
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int c = 0; c < CD; c++) dpf[j][c] = 0.f;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            float t[CD], p[NV];
            {
                const float4 *tq = (const float4 *)(DT_g + i * TQ + h * CD);
                const float4 a = tq[0], b = tq[1];
                t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
            }
            head_softmax(t, pf, p);
            float dp[NV], dot = 0.f;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < CD; c++) a += du[h * CD + c] * pf[j][c];
                dp[j] = a;
                dot += p[j] * a;
            }
            float dt[CD];
#pragma unroll
            for (int c = 0; c < CD; c++) dt[c] = 0.f;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const float ds = p[j] * (dp[j] - dot);
#pragma unroll
                for (int c = 0; c < CD; c++) {
                    dt[c] += ds * pf[j][c];
                    dpf[j][c] += p[j] * du[h * CD + c] + ds * t[c];
                }
            }
#pragma unroll
            for (int c = 0; c < CD; c++) du[h * CD + c] = dt[c];
        }
        {
            float4 *tq = (float4 *)(DT_g + i * TQ);
#pragma unroll
            for (int k = 0; k < TQ / 4; k++) tq[k] = make_float4(du[4 * k], du[4 * k + 1], du[4 * k + 2], du[4 * k + 3]);
        }
#pragma unroll
        for (int j = 0; j < NV; j++)
#pragma unroll
            for (int c = 0; c < CD; c++) dpf_g[(int64_t)(j * CD + c) * n + i] = dpf[j][c];
        // pass C: dxn = Wqk^T dt
        float4 *x4 = (float4 *)(dxn_g + i * FD);
#pragma unroll 2
        for (int k = 0; k < FD / 4; k++) {
            const float4 *w = (const float4 *)(sw + W_QK) + k;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < TQ; r++) {
                const float4 q = w[r * (FD / 4)];
                a.x += q.x * du[r]; a.y += q.y * du[r]; a.z += q.z * du[r]; a.w += q.w * du[r];
            }
            x4[k] = a;
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

unsigned fd_grid(int n) {
    const unsigned tiles = (unsigned)((n + 255) / 256);
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
        hipLaunchKernelGGL(fine_decoder_bwd_kernel, dim3(fd_grid(n)), dim3(256), 0, s, n, xn, pf, Wqk, W1ov, b1, W2, b2,
                           d_sh, d_xn, d_pf, U, HID_, DH, DT);
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
