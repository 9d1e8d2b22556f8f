// attention.hip -- LaRa's group cross-attention (lightning/network.py:57-102, attention part
// :88-93) on gfx950 matrix cores.
//
// Reference op, per group g of 8 query voxels and 4 image-feature tokens (one per input view):
//     y = x + W_o . softmax( (LN(x) W_q^T) (c W_k^T)^T / 4 ) (c W_v^T)      16 heads x head_dim 16
// with x [G,8,256], c [G,4,800], nn.MultiheadAttention(256, 16, kdim=vdim=800, bias=False).
//
// Kernels (bf16 operands, fp32 accumulate -- what the reference runs under bf16-mixed autocast):
//   ln_cast          LayerNorm(256) + cast to bf16, one wave per token             HBM-bound
//   gemm_bf16_nt     C[M,N] = A[M,K] W[N,K]^T on v_mfma_f32_32x32x16_bf16; used for the Q, K|V and
//                    output projections (epilogue: bf16 store, or fp32 store + residual add)
//   group_attn       QK^T, softmax over the 4 keys, and AV on v_mfma_f32_16x16x16_bf16.  The tiny
//                    per-(group, head) problems (8x16 . 16x4 and 8x4 . 4x16) are packed four groups
//                    to a tile: S^T = K Q^T puts a query's four scores into the four accumulator
//                    registers of one lane (softmax needs no cross-lane traffic), and those registers
//                    ARE the A-operand fragment of the following P.V product.
#include "common.h"
#include "../../include/lara_groupattn.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// ---- LayerNorm(C = 256) + bf16 cast: one wave per token, 4 channels per lane ---------------------
__global__ void __launch_bounds__(256)
ln_cast_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
               const float *__restrict__ beta, const float eps, unsigned short *__restrict__ out,
               const int tokens) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tok >= tokens) return;
    const float4 v = ((const float4 *)(x + (size_t)tok * 256))[lane];
    float s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s * (1.0f / 256.0f);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d4 = v.w - mean;
    float q = a * a + b * b + c * c + d4 * d4;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + eps);
    const float4 g = ((const float4 *)gamma)[lane], be = ((const float4 *)beta)[lane];
    ushort4 o;
    o.x = f2bf(a * rstd * g.x + be.x); o.y = f2bf(b * rstd * g.y + be.y);
    o.z = f2bf(c * rstd * g.z + be.z); o.w = f2bf(d4 * rstd * g.w + be.w);
    ((ushort4 *)(out + (size_t)tok * 256))[lane] = o;
}

// ---- C[M,N] = A[M,K] . W[N,K]^T, bf16 in, fp32 accumulate ------------------------------------------
// Workgroup tile 128x128, K step 32, four waves as 2x2, wave tile 64x64 = 2x2 MFMA tiles of 32x32.
// Operand panels are staged through LDS (double buffered, register staging: the next K tile is
// loaded into VGPRs while the current one feeds the matrix cores).  Both operands have K contiguous,
// so a fragment is one ds_read_b128; LDS rows are padded from 64 to 80 bytes, which spreads the 16
// rows a ds_read_b128 lane group touches over all 16 sixteen-byte slots of the 256-byte bank row
// (conflict free).  MFMA 32x32x16 operand map: A[i = lane&31][k = 8*(lane>>5) + e],
// B[k = 8*(lane>>5) + e][j = lane&31]; C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
constexpr int GM = 128, GN = 128, GK = 32;
constexpr int LROW = GK * 2 + 16;  // padded LDS row, bytes

template <int EPI>  // 0: bf16 store   1: fp32 store of acc + resid
__global__ void __launch_bounds__(256)
gemm_bf16_nt_kernel(const unsigned short *__restrict__ A, const unsigned short *__restrict__ W,
                    void *__restrict__ Cout, const float *__restrict__ resid, const int M, const int N,
                    const int K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][(GM + GN) * LROW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bm0 = blockIdx.x * GM, bn0 = blockIdx.y * GN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r = lane & 31, kh = lane >> 5;

    // staging map: 256 rows (128 of A, 128 of W) x 4 sixteen-byte chunks per K tile; thread t moves
    // chunk (t & 3) of rows (t >> 2) + 64 i, i = 0..3
    const int srow = tid >> 2, schunk = tid & 3;
    const unsigned short *gsrc[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = srow + 64 * i;  // 0..255: A rows then W rows
        const unsigned short *base = row < GM ? A + (size_t)min(bm0 + row, M - 1) * K
                                              : W + (size_t)min(bn0 + row - GM, N - 1) * K;
        gsrc[i] = base + schunk * 8;
        loff[i] = row * LROW + schunk * 16;
    }
    const int ktiles = (K + GK - 1) / GK;
    uint4 stage[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = kt * GK + schunk * 8;
            stage[i] = k < K ? *(const uint4 *)(gsrc[i] + (size_t)kt * GK) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) *(uint4 *)(&lds[buf][loff[i]]) = stage[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < ktiles) gload(kt + 1);  // in flight while this tile is multiplied
        const unsigned char *la = &lds[buf][(wm + r) * LROW + kh * 16];
        const unsigned char *lb = &lds[buf][(GM + wn + r) * LROW + kh * 16];
#pragma unroll
        for (int s = 0; s < GK / 16; s++) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a[i] = *(const bf16x8 *)(la + i * 32 * LROW + s * 32);
                b[i] = *(const bf16x8 *)(lb + i * 32 * LROW + s * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ktiles) lstore(buf ^ 1);  // the other buffer was last read one iteration ago
        __syncthreads();
    }
    // Epilogue through LDS: the accumulator layout gives every lane one element of 16 different
    // rows (4-byte stores, issue bound); bouncing a 32x64 block per wave through LDS turns that into
    // 16-byte row-contiguous loads of the residual and 16-byte stores.
    float *ep = (float *)&lds[0][0] + wave * (32 * 68);  // 32 rows x (64 + 4 pad) floats per wave
#pragma unroll
    for (int i = 0; i < 2; i++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ep[((e & 3) + 8 * (e >> 2) + 4 * kh) * 68 + j * 32 + r] = acc[i][j][e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = q * 64 + lane, lr = idx >> 4, c4 = (idx & 15) * 4;
            const int row = bm0 + wm + i * 32 + lr, col = bn0 + wn + c4;
            if (row < M && col < N) {
                const float4 v = *(const float4 *)(ep + lr * 68 + c4);
                const size_t o = (size_t)row * N + col;
                if (EPI == 0) {
                    ushort4 h;
                    h.x = f2bf(v.x); h.y = f2bf(v.y); h.z = f2bf(v.z); h.w = f2bf(v.w);
                    *(ushort4 *)((unsigned short *)Cout + o) = h;
                } else {
                    const float4 rs = *(const float4 *)(resid + o);
                    *(float4 *)((float *)Cout + o) = make_float4(v.x + rs.x, v.y + rs.y, v.z + rs.z, v.w + rs.w);
                }
            }
        }
    }
}

// ---- per-group softmax attention on 16x16x16 bf16 MFMA ---------------------------------------------
// One wave per unit of 4 groups (32 query tokens, 16 key/value tokens); Q [G*8,256], KV [G*4,512]
// (K in columns 0..255, V in 256..511), O [G*8,256]; all bf16.
__global__ void __launch_bounds__(256)
group_attn_kernel(const unsigned short *__restrict__ Q, const unsigned short *__restrict__ KV,
                  unsigned short *__restrict__ O, const int G) {
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g0 = unit * 4;
    if (g0 >= G) return;
    const int c16 = lane & 15, q4 = lane >> 4;
    // a ragged last unit re-reads the last valid group; its stores are masked
    const int kv_row = min(g0 * 4 + c16, G * 4 - 1);        // K as A operand: row = key token
    const int v_row0 = min(g0 * 4 + 4 * q4, G * 4 - 4);     // V as B operand: k = key token 4*q4 + e
    for (int h = 0; h < 16; h++) {
        const s16x4 kf = *(const s16x4 *)(KV + (size_t)kv_row * 512 + h * 16 + 4 * q4);
        s16x4 vf;
#pragma unroll
        for (int e = 0; e < 4; e++) vf[e] = (short)KV[(size_t)(v_row0 + e) * 512 + 256 + h * 16 + c16];
#pragma unroll
        for (int t = 0; t < 2; t++) {  // query groups g0+2t, g0+2t+1
            const int q_row = min((g0 + 2 * t) * 8 + c16, G * 8 - 1);
            const s16x4 qf = *(const s16x4 *)(Q + (size_t)q_row * 256 + h * 16 + 4 * q4);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            // S^T[key = 4*q4 + r][query = c16]
            s = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf, s, 0, 0, 0);
            const bool valid = q4 == 2 * t + (c16 >> 3);  // key group == query group
            const float s0 = s[0] * 0.25f, s1 = s[1] * 0.25f, s2 = s[2] * 0.25f, s3 = s[3] * 0.25f;
            const float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
            const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx), e2 = __expf(s2 - mx), e3 = __expf(s3 - mx);
            const float inv = valid ? 1.0f / (e0 + e1 + e2 + e3) : 0.f;
            s16x4 pf;  // P[query = c16][key = 4*q4 + e]: exactly the A-operand fragment of P.V
            pf[0] = (short)f2bf(e0 * inv); pf[1] = (short)f2bf(e1 * inv);
            pf[2] = (short)f2bf(e2 * inv); pf[3] = (short)f2bf(e3 * inv);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, vf, o, 0, 0, 0);  // O[query = 4*q4 + r][d = c16]
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int tok = (g0 + 2 * t) * 8 + 4 * q4 + rr;
                if (tok < G * 8) O[(size_t)tok * 256 + h * 16 + c16] = f2bf(o[rr]);
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t lara_groupattn_workspace_bytes(int32_t G) {
    if (G < 0) return LARA2DGS_E_INVALID;
    // xn [G*8,256] + q [G*8,256] + kv [G*4,512] + o [G*8,256], bf16
    return (int64_t)G * (8 * 256 * 3 + 4 * 512) * 2 + 1024;
}

int lara_groupattn_forward(int32_t G, int32_t cond_dim, const float *x, const uint16_t *cond_bf16,
                           const float *ln_weight, const float *ln_bias, float eps,
                           const uint16_t *wq, const uint16_t *wkv, const uint16_t *wo, float *y,
                           void *workspace, void *stream) {
    if (G < 0 || cond_dim <= 0 || (cond_dim % 16) != 0) return LARA2DGS_E_INVALID;
    if (G == 0) return LARA2DGS_OK;
    if (!x || !cond_bf16 || !ln_weight || !ln_bias || !wq || !wkv || !wo || !y || !workspace)
        return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    unsigned short *xn = (unsigned short *)workspace;
    unsigned short *q = xn + (size_t)G * 8 * 256;
    unsigned short *kv = q + (size_t)G * 8 * 256;
    unsigned short *o = kv + (size_t)G * 4 * 512;
    const int Mq = G * 8, Mkv = G * 4;
    {
        L2D_PROF("ga_ln_cast", s);
        hipLaunchKernelGGL(ln_cast_kernel, dim3((Mq + 3) / 4), dim3(256), 0, s, x, ln_weight, ln_bias, eps, xn, Mq);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_gemm_q", s);
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0>), dim3((Mq + 127) / 128, 2), dim3(256), 0, s, xn, wq,
                           (void *)q, (const float *)nullptr, Mq, 256, 256);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_gemm_kv", s);
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0>), dim3((Mkv + 127) / 128, 4), dim3(256), 0, s, cond_bf16,
                           wkv, (void *)kv, (const float *)nullptr, Mkv, 512, cond_dim);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_attn", s);
        hipLaunchKernelGGL(group_attn_kernel, dim3((G + 15) / 16), dim3(256), 0, s, q, kv, o, G);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_gemm_o", s);
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<1>), dim3((Mq + 127) / 128, 2), dim3(256), 0, s, o, wo,
                           (void *)y, x, Mq, 256, 256);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
