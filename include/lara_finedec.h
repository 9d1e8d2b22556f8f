/*
 * lara_finedec.h -- LaRa's fine-stage decoder `Decoder.forward_fine` as one kernel per direction (part of
 * liblara2dgs.so).  SURVEY.md section 8f row 4, second half.
 *
 * Replaces lightning/network.py:280-284 (modules declared at :234-240), called at :518 on the <= 524 288 surviving
 * Gaussians of a scene, directly behind the point sampler (lara_pointfeat.h):
 *
 *     volume_feat = self.norm(volume_feat.unsqueeze(1))                       LayerNorm(80)
 *     x  = self.cross_att(volume_feat, point_feats, point_feats)[0]           MultiheadAttention(80, 8 heads, kdim=vdim=8,
 *                                                                             bias=False): ONE query against the 4 views
 *     sh = self.mlp_fine(x).float()                                           Linear(80,64) + ReLU + Linear(64,12)
 *
 * With one query per point the attention is linear algebra around a 4-way softmax, and the projections fold
 * (in exact arithmetic; the Python wrapper forms the folded matrices with autograd, so gradients reach the
 * reference's own parameters):
 *     t[h]   = Wqk[h] xn,            Wqk[h]  = Wk[h]^T Wq[h] / sqrt(10)     [8 x 80] per head  -> Wqk  [64,80]
 *     p[h,j] = softmax_j( t[h] . pf[j] ),      u[h] = sum_j p[h,j] pf[j]    (pf[j] = the 8 sampled channels of view j)
 *     hid    = relu(W1ov u + b1),    W1ov    = W1 Wo blockdiag(Wv[h])       [64,64]   (out-projection and first MLP
 *                                                                            layer have no non-linearity between them)
 *     sh     = W2 hid + b2
 * Per point: 80 + 32 floats in, 12 out, ~10.6 k FMAs (the unfolded sequence: ~25 k and eight [n,80]-sized
 * intermediates).  fp32 arithmetic (the reference trains these layers under bf16 autocast; fp32 is at least as
 * precise).  Layouts: xn [n,80] = the LayerNorm output; pf [4][8][n] = the sampler's own output layout (the
 * reference permutes it to [n,4,8] with an einsum, network.py:515); sh [n,12].
 *
 * Backward: d_sh [n,12] -> d_xn [n,80], d_pf [4][8][n], plus the four per-point factor arrays the weight gradients are
 * products of over the point axis (caller allocates [n,64] each):
 *     U, HID (post-ReLU), DH = dL/d(pre-ReLU), DT = dL/dt:
 *     dWqk = DT^T xn,  dW1ov = DH^T U,  db1 = colsum(DH),  dW2 = d_sh^T HID,  db2 = colsum(d_sh)
 * -- lara_fine_decoder_wgrad below (fp32 matrix-core products over 512-row slabs + an ordered sum: reproducible).
 * Returns 0 or a negative LARA2DGS_E_* code; work is enqueued on `stream`, no host synchronisation.
 * Only the reference's sizes are built (80 / 8 heads x 10 / 4 views x 8 channels / 64 / 12).
 */
#ifndef LARA_FINEDEC_H
#define LARA_FINEDEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int lara_fine_decoder_forward(int32_t n, const float *xn, const float *pf, const float *Wqk, const float *W1ov,
                              const float *b1, const float *W2, const float *b2, float *sh, void *stream);

int lara_fine_decoder_backward(int32_t n, const float *xn, const float *pf, const float *Wqk, const float *W1ov,
                               const float *b1, const float *W2, const float *b2, const float *d_sh, float *d_xn,
                               float *d_pf, float *U, float *HID, float *DH, float *DT, void *stream);

/* The five parameter gradients from the factor arrays the backward left: out = [dWqk 64x80 | dW1ov 64x64 | db1 64 | dW2 12x64 |
 * db2 12] (lara_fine_wgrad_floats() = 10 060 floats, fully overwritten); workspace: lara_fine_decoder_wgrad_workspace_bytes(n). */
int32_t lara_fine_wgrad_floats(void);
int64_t lara_fine_wgrad_workspace_bytes(int32_t n);
int lara_fine_decoder_wgrad(int32_t n, const float *xn, const float *U, const float *HID, const float *DH, const float *DT,
                            const float *d_sh, float *out, void *workspace, void *stream);

/* The LayerNorm in front (network.py:281, `self.norm`): rows of 80 features are too short for torch's row-per-
 * workgroup kernels (0.55 ms forward / 0.9 ms backward for 524 288 rows); here one thread owns a row.
 *   forward : x [n,80] -> xn [n,80] = (x - mean) * rstd * gamma + beta, stats [n,2] = (mean, rstd)   (eps as given)
 *   backward: d_xn [n,80] -> d_x [n,80]; partials [blocks][160]: per workgroup, the sums over its rows of
 *             d_xn * xhat (80) and d_xn (80) -- d_gamma / d_beta are their column sums (fixed order: reproducible).
 *             `blocks` = lara_fine_ln_blocks(n). */
int32_t lara_fine_ln_blocks(int32_t n);
int lara_fine_ln_forward(int32_t n, const float *x, const float *gamma, const float *beta, float eps, float *xn,
                         float *stats, void *stream);
int lara_fine_ln_backward(int32_t n, const float *x, const float *gamma, const float *stats, const float *d_xn,
                          float *d_x, float *partials, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_FINEDEC_H */
