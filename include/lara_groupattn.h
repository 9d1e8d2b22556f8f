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

/* bytes of the caller-owned workspace for G groups */
int64_t lara_groupattn_workspace_bytes(int32_t G);

/* y[G,8,256] (fp32) = x + out_proj(attention(LN(x), cond)).  x fp32 [G,8,256];
 * cond_bf16 [G,4,cond_dim] bf16; ln_weight / ln_bias fp32 [256].  Returns 0 or LARA2DGS_E_*. */
int lara_groupattn_forward(int32_t G, int32_t cond_dim, const float *x, const uint16_t *cond_bf16,
                           const float *ln_weight, const float *ln_bias, float eps,
                           const uint16_t *wq, const uint16_t *wkv, const uint16_t *wo, float *y,
                           void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_GROUPATTN_H */
