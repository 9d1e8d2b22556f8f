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
#include "group_attn.h"
#include "../../include/lara_groupattn.h"

extern "C" {

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
    // LayerNorm + Q projection + attention + output projection + residual as ONE wave-private kernel behind the K|V
    // projection: x read once (and once more for the residual), y written once; xn, Q and O never leave the CU (packed weight
    // fragments, K|V's second half through registers at 24 KB of LDS per wave, epilogue bounced through LDS, hardware bf16
    // conversion: 141 us against 190 us for the four launches it replaces; DESIGN.md section 3.3.  The training forward, which
    // keeps xn / Q / O for the backward, is lara_groupblock_forward_train in encoder_bwd.hip.)
    unsigned short *kvf = (unsigned short *)((char *)workspace + 262144) + (size_t)G * 8 * 256 * 2;
    {
        L2D_PROF("ga_gemm_kv", s);
        GemmP p{};
        p.A = cond_bf16; p.W = wkv; p.C = kvf; p.M = G * 4; p.N = 512; p.K = cond_dim;
        // K = cond_dim = 800: deep enough for the LDS-DMA ring (102 -> 77 us)
        if (launch_gemm_ring<0, 0>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("ga_fused", s);
        const int units = (G + 3) / 4;
        // the two 256 x 256 weights in fragment order (2 x 128 KB at the start of the workspace): one launch of 64
        // workgroups in front of the step
        unsigned short *wqp = (unsigned short *)workspace, *wop = wqp + 65536;
        hipLaunchKernelGGL(pack_weight_frag_kernel, dim3(64), dim3(256), 0, s, wq, wqp, wo, wop);
        hipLaunchKernelGGL(group_attn_fused2_kernel<false>, dim3(units), dim3(64), 0, s, x, ln_weight, ln_bias, eps, wqp, kvf, wop, y, G,
                           (unsigned short *)nullptr, (unsigned short *)nullptr, (unsigned short *)nullptr);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
