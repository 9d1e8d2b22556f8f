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
#include "mfma_gemm.h"
#include "../../include/lara_groupattn.h"

namespace {

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
    if (G < 0 || cond_dim <= 0 || (cond_dim % 32) != 0) return LARA2DGS_E_INVALID;
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
        hipLaunchKernelGGL(ln_cast_kernel, dim3((Mq + 3) / 4), dim3(256), 0, s, x, ln_weight, ln_bias, eps, xn,
                           (float2 *)nullptr, Mq);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_gemm_q", s);
        GemmP p{};
        p.A = xn; p.W = wq; p.C = q; p.M = Mq; p.N = 256; p.K = 256;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 0>), dim3((Mq + 127) / 128, 2), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_gemm_kv", s);
        GemmP p{};
        p.A = cond_bf16; p.W = wkv; p.C = kv; p.M = Mkv; p.N = 512; p.K = cond_dim;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 0>), dim3((Mkv + 127) / 128, 4), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_attn", s);
        hipLaunchKernelGGL(group_attn_kernel, dim3((G + 15) / 16), dim3(256), 0, s, q, kv, o, G);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_gemm_o", s);
        GemmP p{};
        p.A = o; p.W = wo; p.C = y; p.resid = x; p.M = Mq; p.N = 256; p.K = 256;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 1>), dim3((Mq + 127) / 128, 2), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
