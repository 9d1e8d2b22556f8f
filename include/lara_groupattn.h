/*
 * lara_groupattn.h -- C ABI of the group cross-attention of LaRa's volume transformer on MI355X
 * matrix cores (part of liblara2dgs.so).
 *
 * Replaces, for the forward pass, the attention step of `GroupAttBlock.forward`
 * (lightning/network.py:88-93):
 *     patches = patches + self.cross_attn(self.norm1(patches), cond, cond, need_weights=False)[0]
 * where `cross_attn = nn.MultiheadAttention(256, 16 heads, kdim=vdim=800, bias=False,
 * batch_first=True)` (network.py:65-67), patches = [G, 8, 256] (2^3 voxels per group) and
 * cond = [G, 4, 800] (one image-feature token per input view; network.py:145-150).
 * Operands are bf16 with fp32 accumulation -- the precision the reference runs this step in under
 * `precision="bf16-mixed"` (train_lightning.py:74); LayerNorm and softmax are fp32.
 *
 * All pointers are device pointers; weights are bf16 in nn.MultiheadAttention's own layouts:
 *   wq  [256, 256]       = q_proj_weight
 *   wkv [512, cond_dim]  = k_proj_weight stacked over v_proj_weight
 *   wo  [256, 256]       = out_proj.weight
 * Work is enqueued on `stream` (hipStream_t as void*); no host synchronisation.
 */
#ifndef LARA_GROUPATTN_H
#define LARA_GROUPATTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of the caller-owned workspace for G groups (size it with THIS function: round 4 added 256 KB in front for the fused
 * kernel's packed weight fragments) */
int64_t lara_groupattn_workspace_bytes(int32_t G);

/* y[G,8,256] (fp32) = x + out_proj(attention(LN(x), cond)).  x fp32 [G,8,256];
 * cond_bf16 [G,4,cond_dim] bf16; ln_weight / ln_bias fp32 [256].  Returns 0 or LARA2DGS_E_*. */
int lara_groupattn_forward(int32_t G, int32_t cond_dim, const float *x, const uint16_t *cond_bf16,
                           const float *ln_weight, const float *ln_bias, float eps,
                           const uint16_t *wq, const uint16_t *wkv, const uint16_t *wo, float *y,
                           void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The whole GroupAttBlock and the VolTransformer tail (rows A2 / A3 of SURVEY.md section 8).
 *
 * Activations live as fp32 *token rows* x[M, 256], M = scenes * R^3, in GROUP-MAJOR order: row
 *   m = ((((b * R/2 + gd) * R/2 + gh) * R/2 + gw) * 8 + (z*4 + y*2 + x)   <->  voxel (2gd+z, 2gh+y, 2gw+x)
 * which is the order `GroupAttBlock.forward` unfolds the volume into for attention
 * (network.py:82-86, block_size 2).  The reference converts volume <-> patches twice per layer
 * (network.py:82-86, 96-98); here the 3x3x3 convolution reads its neighbours through the token
 * order instead, so the layout never changes between layers.
 * ---------------------------------------------------------------------------------------------- */

/* Weights of one GroupAttBlock (network.py:57-79), device pointers.  bf16 matrices keep torch's
 * [out, in] layout; wconv is cnn.weight [256, 256, 3, 3, 3] re-laid as [out][kd][kh][kw][in]. */
typedef struct lara_groupblock_weights {
    const float *ln1_w, *ln1_b;        /* norm1                          [256]            */
    const uint16_t *wq, *wkv, *wo;     /* cross_attn (see above)                          */
    const float *ln2_w, *ln2_b;        /* norm2                                           */
    const uint16_t *w1;                /* mlp[0].weight                  [512, 256] bf16  */
    const float *b1;                   /* mlp[0].bias                    [512]            */
    const uint16_t *w2;                /* mlp[3].weight                  [256, 512] bf16  */
    const float *b2;                   /* mlp[3].bias                    [256]            */
    const float *ln3_w, *ln3_b;        /* norm3                                           */
    const uint16_t *wconv;             /* cnn.weight re-laid             [256, 27, 256]   */
    float eps;                         /* LayerNorm eps (all three norms)                 */
} lara_groupblock_weights;

int64_t lara_groupblock_workspace_bytes(int32_t scenes, int32_t R);

/* In place: x <- GroupAttBlock(x, cond) for block_size 2, i.e. (network.py:88-100)
 *     x = x + cross_attn(norm1(x), cond, cond);  x = x + mlp(norm2(x));
 *     x = norm3(x);                               x = x + cnn(x)        (3x3x3, padding 1, no bias)
 * x fp32 [scenes * R^3, 256] group-major; cond_bf16 [scenes * (R/2)^3, 4, cond_dim]. */
int lara_groupblock_forward(int32_t scenes, int32_t R, int32_t cond_dim, float *x,
                            const uint16_t *cond_bf16, const lara_groupblock_weights *w,
                            void *workspace, void *stream);

/* VolTransformer tail (network.py:156-163): out = deconv(norm(x)) as channels-last
 * [scenes, 2R, 2R, 2R, Cout] fp32.  wdeconv = deconv.weight [256, Cout, 2, 2, 2] re-laid as
 * [(i*2+j)*2+k][Cout][256] bf16; bias fp32 [Cout]; Cout % 4 == 0.  Workspace: scenes*R^3*512 bytes. */
int lara_voltrans_head_forward(int32_t scenes, int32_t R, const float *x, const float *ln_w,
                               const float *ln_b, float eps, const uint16_t *wdeconv,
                               const float *bias, int32_t Cout, float *out, void *workspace,
                               void *stream);

/* Layout converters between the reference's volume tensor [scenes, C, R, R, R] (fp32) and token
 * rows [scenes * R^3, C] in group-major order (used once for pos_embed and by the tests). */
int lara_tokens_from_volume(int32_t scenes, int32_t R, int32_t C, const float *volume, float *tokens,
                            void *stream);
int lara_volume_from_tokens(int32_t scenes, int32_t R, int32_t C, const float *tokens, float *volume,
                            void *stream);

/* C[M, N] = A[M, K] . W[N, K]^T, bf16 operands, fp32 accumulate, on the 256 x 256 LDS-DMA ring kernel; C is bf16, or
 * fp32 when c_fp32 != 0.  K % 32 == 0; operands below 4 GB.  (The all-layers dcond product of the training path.) */
int lara_gemm_nt_bf16(int32_t M, int32_t N, int32_t K, const uint16_t *A, const uint16_t *W, void *C, int32_t c_fp32,
                      void *stream);

/* dst[b][c][r] = src[b][r][c] for `batch` row-major fp32 [rows, cols] matrices; dst is fp32, or bf16 (round to
 * nearest even) when dst_bf16 != 0.  VolTransformer.forward's `b v c d h w -> (b d h w) v c` rearrangement of the
 * image features fused with their bf16 cast (network.py:145-150: rows = v * c, cols = d * h * w) and, with rows and
 * cols swapped, its backward -- one pass each instead of a strided torch copy plus a cast. */
int lara_batched_transpose(int32_t batch, int32_t rows, int32_t cols, const float *src, void *dst, int32_t dst_bf16,
                           void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training.  In the reference the backward is torch autograd through the modules above under bf16-mixed
 * autocast.  Here
 *   lara_groupblock_forward_train   x_out <- GroupAttBlock(x_in, cond), out of place, and every intermediate
 *                                   the backward needs (LayerNorm outputs, q, k|v, attention output, the
 *                                   MLP's pre-activation and hidden rows, ...) kept in `saved`
 *                                   (lara_groupblock_save_bytes: 7.2 KB per token row)
 *   lara_groupblock_backward        g   in: dL/d(block output) fp32 [M, 256]   out: dL/d(block input), in place
 *                                   dcond  += dL/d(cond) fp32 [M/2, cond_dim]   (the same cond feeds every layer)
 *                                   dw     += the parameter gradients, fp32, layouts of lara_groupblock_weights
 *                                   `saved`: what forward_train left, or NULL to re-run the block's forward
 *                                   inside the backward (trades 0.6 ms per layer for the memory)
 * The caller zero-fills dcond and dw once per step.  Matrix products: bf16 operands, fp32 accumulate;
 * all reductions have a fixed order (bit-reproducible).  Requires R >= 4.
 * ---------------------------------------------------------------------------------------------- */

/* the bf16 matrices of lara_groupblock_weights, transposed (row-major [in, out]); wconv_t is
 * [in][mirrored tap][out], i.e. wconv_t[ci][t][co] = wconv[co][26 - t][ci] */
typedef struct lara_groupblock_weights_t {
    const uint16_t *wq_t;    /* [256, 256]        */
    const uint16_t *wkv_t;   /* [cond_dim, 512]   */
    const uint16_t *wo_t;    /* [256, 256]        */
    const uint16_t *w1_t;    /* [256, 512]        */
    const uint16_t *w2_t;    /* [512, 256]        */
    const uint16_t *wconv_t; /* [256, 27, 256]    */
} lara_groupblock_weights_t;

/* fp32 gradient accumulators, same shapes as the fields of lara_groupblock_weights */
typedef struct lara_groupblock_grads {
    float *ln1_w, *ln1_b, *wq, *wkv, *wo, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2, *ln3_w, *ln3_b, *wconv;
} lara_groupblock_grads;

int64_t lara_groupblock_backward_workspace_bytes(int32_t scenes, int32_t R);
int64_t lara_groupblock_save_bytes(int32_t scenes, int32_t R);

int lara_groupblock_forward_train(int32_t scenes, int32_t R, int32_t cond_dim, const float *x_in, float *x_out,
                                  const uint16_t *cond_bf16, const lara_groupblock_weights *w, void *saved,
                                  void *stream);

int lara_groupblock_backward(int32_t scenes, int32_t R, int32_t cond_dim, const float *x_in,
                             const uint16_t *cond_bf16, const lara_groupblock_weights *w,
                             const lara_groupblock_weights_t *wt, const void *saved, float *g, float *dcond,
                             const lara_groupblock_grads *dw, int32_t chained, uint16_t *dkv, int32_t lddkv,
                             void *workspace, void *stream);
/* dkv != NULL: leave dL/d(K|V) of this block there (rows of 512 bf16, `lddkv` elements apart; lddkv >= 512,
 * lddkv % 8 == 0) and do NOT touch dcond (which may then be NULL).  The same cond feeds every layer
 * (network.py:152-155): the caller forms dcond = [dkv of all layers] . [wkv of all layers] in ONE product after the
 * sweep (lara_gemm_nt_bf16, K = layers * 512) instead of one read-modify-write of the fp32 dcond tensor per layer. */
/* chained != 0: this call continues a backward sweep -- `workspace` was last used by a lara_groupblock_backward call
 * with the same (scenes, R) whose output g is this call's input g, and nothing touched either since.  The call then
 * reuses what that call left in the workspace (the bf16 copy of g its last LayerNorm backward wrote, the convolution's
 * neighbour table) instead of rebuilding them.  chained == 0 (the first block of a sweep) builds both. */

/* Backward of lara_voltrans_head_forward.  x: the rows that entered the head; dout: fp32
 * [scenes, 2R, 2R, 2R, Cout]; wdeconv_t: wdeconv transposed, [256, 8 * Cout] bf16.  Writes g = dL/dx
 * (fp32 [M, 256]) and ACCUMULATES d_ln_w, d_ln_b [256], d_wdeconv [8 * Cout, 256] and d_bias8
 * [8 * Cout] (the bias gradient per kernel tap: sum the 8 taps).  Cout % 16 == 0. */
int64_t lara_voltrans_head_backward_workspace_bytes(int32_t scenes, int32_t R, int32_t Cout);
int lara_voltrans_head_backward(int32_t scenes, int32_t R, const float *x, const float *ln_w, const float *ln_b,
                                float eps, const uint16_t *wdeconv_t, int32_t Cout, const float *dout, float *g,
                                float *d_ln_w, float *d_ln_b, float *d_wdeconv, float *d_bias8, void *workspace,
                                void *stream);

/* Building blocks of the above, exported for the parity tests:
 *   dst[N, Kc] += A[M, N]^T . B[M, Kc]   (bf16 operands, M % 16 == 0, N and Kc even)
 *   LayerNorm(256) backward: dx = dLN(dy; x, gamma) (+ skip); dgamma, dbeta accumulated
 *   backward of the per-group attention core: (q, k|v, dO) -> dq [G*8, 256], dk|dv [G*4, 512], bf16 */
int64_t lara_gemm_tn_workspace_bytes(void);
int lara_gemm_tn_bf16(int32_t M, int32_t N, int32_t Kc, const uint16_t *A, const uint16_t *B, float *dst,
                      void *workspace, void *stream);
int lara_layernorm256_backward(int32_t rows, const float *dy, const float *x, const float *gamma, float eps,
                               const float *skip, float *dx, float *dgamma, float *dbeta, void *workspace,
                               void *stream);
int lara_groupattn_core_backward(int32_t G, const uint16_t *q, const uint16_t *kv, const uint16_t *d_o, uint16_t *dq,
                                 uint16_t *dkv, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_GROUPATTN_H */
