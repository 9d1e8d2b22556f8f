/*
 * lara_loss.h -- the pixel terms of LaRa's training loss as one kernel per direction (part of liblara2dgs.so).
 * The caller directly behind the renderer's outputs ("the callers either side of the path"); opt-in.
 *
 * Replaces the elementwise chains of lightning/loss.py:28-58 (without the MS-SSIM term, :42-45) over the stacked outputs
 * of `Network.forward` ([B, H, V*W, C] maps, network.py:527-529) and the targets `batch['tar_rgb']` [B, V, H, W, 3]
 * (loss.py:24 permutes them into the side-by-side layout: here that is index arithmetic):
 *
 *     terms[0] = mean((image      - tar)^2)                                          loss.py:33-34
 *     terms[1] = mean((image_fine - tar)^2)                                          (the '_fine' round of the same loop)
 *     terms[2] = mean(rend_dist)                                                     loss.py:48
 *     terms[3] = mean((1 - sum_c rend_normal_c * depth_normal_c) * acc_map)          loss.py:52-56 (acc_map detached)
 *
 * (the caller forms loss = t0 + t1 + 1000 t2 + 0.2 t3 and the statistics from the four scalars).  Any of image_fine,
 * rend_dist, and the normal triple (rend_normal, depth_normal, acc_map together) may be NULL: its term is 0.
 * torch runs this as ~25 elementwise / reduction kernels forward and ~30 backward over 8.4 M pixels; here each direction
 * is one pass: forward reads 68 bytes per pixel and leaves per-workgroup partial sums (`partials`,
 * lara_loss_partial_floats(pixels) floats) that a second small kernel adds in a fixed order (reproducible, no atomics);
 * backward reads the same maps and the four upstream gradients g[4] (a device array: no host read) and writes
 * d_image, d_image_fine, d_rend_dist, d_rend_normal, d_depth_normal (any may be NULL = not wanted), fully overwritten.
 * Returns 0 or a negative LARA2DGS_E_* code; work is enqueued on `stream`, no host synchronisation.
 */
#ifndef LARA_LOSS_H
#define LARA_LOSS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int64_t lara_loss_partial_floats(int64_t pixels);

int lara_loss_terms_forward(int32_t B, int32_t V, int32_t H, int32_t W, const float *tar_rgb, const float *image,
                            const float *image_fine, const float *rend_dist, const float *rend_normal,
                            const float *depth_normal, const float *acc_map, float *terms, float *partials, void *stream);

int lara_loss_terms_backward(int32_t B, int32_t V, int32_t H, int32_t W, const float *tar_rgb, const float *image,
                             const float *image_fine, const float *rend_normal, const float *depth_normal,
                             const float *acc_map, const float *g_terms, float *d_image, float *d_image_fine,
                             float *d_rend_dist, float *d_rend_normal, float *d_depth_normal, void *stream);

#ifdef __cplusplus
}
#endif
#endif
