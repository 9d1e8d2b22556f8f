/*
 * lara_coarsedec.h -- LaRa's coarse Gaussian decoder `Decoder.forward_coarse` as one kernel per direction (part of
 * liblara2dgs.so).  The caller directly behind the volume transformer and in front of the rasteriser's inputs
 * ("the callers either side of the path"); opt-in, like the other rows of 8f.
 *
 * Replaces lightning/network.py:259-278 (modules declared at :229-233), called at :458 on every voxel of the batch
 * (B x 64^3 = 1 048 576 rows of 80 features at configs/base.yaml):
 *
 *     parameters = self.mlp_coarse(feats).float()            Linear(80,80) ReLU Linear(80,80) ReLU Linear(80, 22 K)
 *     parameters = parameters.view(*parameters.shape[:-1], K, -1)
 *     offset, sh, opacity, scaling, rotation = split(parameters, [3, sh_dim, 1, 2, 4], dim=-1)
 *     opacity = opacity + opacity_shift;  scaling = scaling + scaling_shift;  offset = sigmoid(offset) * 2 - 1
 *
 * Arithmetic: the reference trains under bf16-mixed (train_lightning.py:74), so each Linear has bf16 operands, fp32
 * accumulation and a bf16 result, and `.float()` widens the last one.  The kernels do exactly that on the matrix cores
 * (v_mfma_f32_16x16x16_bf16; weights and biases are rounded to bf16 when they are staged, activations when they become
 * the next layer's operand), computing H^T = W X^T so that a layer's accumulator registers ARE the next layer's operand
 * (csrc/coarsedec.hip).  Rows of 80 features in, 22 K values out: the kernels are bound by their HBM streams.
 *
 * forward:  x [M,80] fp32 -> offset [M,K*3] (in (-1,1)), sh [M,K*sh_dim], scaling [M,K*2], rotation [M,K*4],
 *           opacity [M,K] -- i.e. the reference's [B, voxels*K, c] tensors, which are the same memory.
 * backward: gradients of the five outputs (any may be NULL = zero) + x + the offset OUTPUT -> dx [M,80] fp32, and the
 *           bf16 factor matrices the parameter gradients are products of, M_pad = lara_coarse_decoder_padded_rows(M) rows each
 *           (rows >= M of the dz matrices are written as zeros):
 *               xb, h1, h2 [M_pad,88]: 80 features, then a column of ones and 7 of zeros
 *               dz1, dz2 [M_pad,80];  dz3 [M_pad,48]  (columns >= 22 K zero)
 *               G1 = dz1^T xb,  G2 = dz2^T h1,  G3 = dz3^T h2   [80|80|48, 88]     (lara_gemm_tn_bf16, lara_groupattn.h)
 *               dW_l = G_l[:, :80] (G3: rows < 22 K),  db_l = G_l[:, 80] -- the column of ones makes the bias gradient (the
 *               column sums of dz) part of the same product: no separate reduction, no atomics, reproducible.
 * Only the reference's sizes are built: 80 features, hidden 80, K * (10 + sh_dim) <= 48.
 * Returns 0 or a negative LARA2DGS_E_* code; work is enqueued on `stream`, no host synchronisation.
 */
#ifndef LARA_COARSEDEC_H
#define LARA_COARSEDEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int64_t lara_coarse_decoder_padded_rows(int64_t M);

int lara_coarse_decoder_forward(int32_t M, int32_t K, int32_t sh_dim, const float *x, const float *w1, const float *b1,
                                const float *w2, const float *b2, const float *w3, const float *b3, float opacity_shift,
                                float scaling_shift, float *offset, float *sh, float *scaling, float *rotation,
                                float *opacity, void *stream);

int lara_coarse_decoder_backward(int32_t M, int32_t K, int32_t sh_dim, const float *x, const float *w1, const float *b1,
                                 const float *w2, const float *b2, const float *w3, const float *offset_out,
                                 const float *d_offset, const float *d_sh, const float *d_scaling, const float *d_rotation,
                                 const float *d_opacity, float *dx, uint16_t *xb, uint16_t *h1, uint16_t *h2,
                                 uint16_t *dz1, uint16_t *dz2, uint16_t *dz3, void *stream);

#ifdef __cplusplus
}
#endif
#endif
