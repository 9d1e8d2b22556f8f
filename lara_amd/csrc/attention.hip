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
#include <cstdlib>

#include "mfma_gemm.h"
#include "group_attn.h"
#include "../../include/lara_groupattn.h"

#include <atomic>
namespace {
// 0: five launches (LayerNorm, Q, K|V, attention, output projection); 1 / 2: the K|V projection + ONE wave-private kernel for
// the rest (group_attn_fused_kernel / group_attn_fused2_kernel).  LARA_GA_FUSED at load (default 2: 216 us per layer at 4 scenes
// against 286 for the five launches and 302 for the first cut), lara_groupattn_set_fused at run time.
std::atomic<int> g_ga_fused{[] { const char *e = getenv("LARA_GA_FUSED"); const int m = e ? atoi(e) : 2; return m < 0 ? 0 : (m > 2 ? 2 : m); }()};
}

extern "C" {

int lara_groupattn_set_fused(int32_t mode) { return g_ga_fused.exchange(mode < 0 ? 0 : (mode > 2 ? 2 : mode)); }

int64_t lara_groupattn_workspace_bytes(int32_t G) {
    if (G < 0) return LARA2DGS_E_INVALID;
    // xn [G*8,256] + q [G*8,256] + kv [G*4,512] + o [G*8,256], bf16; + 256 KB in front for the fused path's packed weights
    return (int64_t)G * (8 * 256 * 3 + 4 * 512) * 2 + 1024 + 262144;
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
    {
        // LayerNorm + Q projection + attention + output projection + residual as ONE wave-private kernel behind the K|V
        // projection: x read once (and once more for the residual), y written once; xn, Q and O never leave the CU.  The
        // first cut (mode 1, round 2) lost to the four launches it replaces (237 vs 190 us); the second (mode 2, round 4: packed
        // weight fragments, K|V's second half through registers at 24 KB of LDS per wave, epilogue bounced through LDS, hardware
        // bf16 conversion) takes 141 us.  DESIGN.md section 3.3.
        const int fused = g_ga_fused.load();
        if (fused) {
            unsigned short *kvf = (unsigned short *)((char *)workspace + 262144) + (size_t)G * 8 * 256 * 2;
            {
                L2D_PROF("ga_gemm_kv", s);
                GemmP p{};
                p.A = cond_bf16; p.W = wkv; p.C = kvf; p.M = G * 4; p.N = 512; p.K = cond_dim;
                if (launch_gemm_ring<0, 0>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
            }
            L2D_CHECK_LAUNCH();
            {
                L2D_PROF("ga_fused", s);
                const int units = (G + 3) / 4;
                if (fused == 2) {
                    // the two 256 x 256 weights in fragment order (2 x 128 KB at the start of the workspace): two launches of 32
                    // workgroups in front of the step
                    unsigned short *wqp = (unsigned short *)workspace, *wop = wqp + 65536;
                    hipLaunchKernelGGL(pack_weight_frag_kernel, dim3(32), dim3(256), 0, s, wq, wqp);
                    hipLaunchKernelGGL(pack_weight_frag_kernel, dim3(32), dim3(256), 0, s, wo, wop);
                    hipLaunchKernelGGL(group_attn_fused2_kernel, dim3(units), dim3(64), 0, s, x, ln_weight, ln_bias, eps, wqp, kvf, wop, y, G);
                } else
                    hipLaunchKernelGGL(group_attn_fused_kernel, dim3((units + 1) / 2), dim3(128), 0, s, x, ln_weight, ln_bias, eps,
                                       wq, kvf, wo, y, G);
            }
            L2D_CHECK_LAUNCH();
            return LARA2DGS_OK;
        }
    }
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
        // K = cond_dim = 800: deep enough for the LDS-DMA ring (102 -> 77 us); the K = 256 / 512 GEMMs of
        // the block are bound by their fp32 residual streams and ran no faster on it
        if (launch_gemm_ring<0, 0>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
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
