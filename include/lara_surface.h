/*
 * lara_surface.h -- the per-view post-processing of LaRa's renderer as one kernel per direction (part of
 * liblara2dgs.so).  SURVEY.md section 8f row 2 (opt-in fusion).
 *
 * Replaces the torch code between the rasteriser call and the return of `Renderer.render_img`
 * (lightning/renderer_2dgs.py:220-268, with `depth_to_normal` / `depths_to_points` :73-89): ~15 small torch
 * kernels per view forward and ~25 backward, all launch-bound on 512 x 512 maps.
 *
 *   image        [H,W,3] = clamp(color, 0, 1)                                                   (:220, :257)
 *   acc_map      [H,W]   = allmap[1]                                                             (:227, :259)
 *   rend_normal  [H,W,3] = allmap[2:5] (view space) . world_view_transform[:3,:3]^T              (:230-231)
 *   depth        [H,W,1] = (1 - ratio) * nan_to_num(allmap[0] / allmap[1], 0, 0)
 *                          + ratio * nan_to_num(allmap[5], 0, 0)                                 (:234-247)
 *   depth_normal [H,W,3] = normalize(cross(P(y+1,x) - P(y-1,x), P(y,x+1) - P(y,x-1))) * acc_map (no gradient
 *                          through this acc_map), zero on the one-pixel border, with
 *                          P = rays[...,:3] + depth * rays[...,3:]                              (:73-89, :251-255)
 *   rend_dist    [H,W]   = allmap[6]                                                             (:244, :262)
 *
 * color [3,H,W], allmap [7,H,W] are the rasteriser's outputs (lara2dgs_forward); rays [H,W,6]; rot [9] is the
 * row-major 3x3 matrix M with rend_normal = n . M, i.e. M = world_view_transform[:3,:3].T as the caller's
 * camera stores it.  All fp32 device pointers.  The backward takes the gradients of the six outputs (any may
 * be NULL = zero) and writes dL/dcolor [3,H,W] and dL/dallmap [7,H,W] (both fully overwritten).  Where
 * allmap[1] == 0 the reference's `x / 0` backward yields NaN/inf in channels 0 and 1, which the rasteriser's
 * backward never reads (such a pixel has no contributor); this backward drops the non-finite term there
 * (channel 0: 0, channel 1: the acc_map gradient alone).
 * Returns 0 or a negative LARA2DGS_E_* code; work is enqueued on `stream`, no host synchronisation.
 */
#ifndef LARA_SURFACE_H
#define LARA_SURFACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int lara_surface_maps_forward(int32_t H, int32_t W, const float *color, const float *allmap, const float *rays,
                              const float *rot, float depth_ratio, float *image, float *depth, float *acc_map,
                              float *rend_normal, float *depth_normal, float *rend_dist, void *stream);

int lara_surface_maps_backward(int32_t H, int32_t W, const float *color, const float *allmap, const float *rays,
                               const float *rot, float depth_ratio, const float *g_image, const float *g_depth,
                               const float *g_acc_map, const float *g_rend_normal, const float *g_depth_normal,
                               const float *g_rend_dist, float *d_color, float *d_allmap, void *stream);

/* The three activations `render_img` applies in front of the rasteriser call (renderer_2dgs.py:181-189) as one launch per
 * direction: opacity [P,1] -> sigmoid, scales [P,2] -> exp, rotations [P,4] -> x / max(|x|_2, 1e-12) (F.normalize).
 * scales / rotations (and their outputs) may be NULL (the cov3D_precomp path).  The backward takes the ACTIVATED opacity and
 * scales and the RAW rotations; gradients may be NULL (= zero), outputs NULL (= not wanted). */
int lara_activate_gaussians_forward(int64_t P, const float *opacity, const float *scales, const float *rotations,
                                    float *opacity_out, float *scales_out, float *rotations_out, void *stream);

int lara_activate_gaussians_backward(int64_t P, const float *opacity_act, const float *scales_act, const float *rotations,
                                     const float *g_opacity, const float *g_scales, const float *g_rotations,
                                     float *d_opacity, float *d_scales, float *d_rotations, void *stream);

/* All views of a scene in one launch per direction, written side by side the way `Network.forward` concatenates them
 * (lightning/network.py:527: `torch.cat([view[k] ...], dim=1)`): color [n,3,H,W], allmap [n,7,H,W], rays [n,H,W,6],
 * rots [n,9]; every output (and, in the backward, every output gradient) is ONE [H, n*W, C] map in which view v owns the
 * columns [v*W, (v+1)*W).  d_color [n,3,H,W] and d_allmap [n,7,H,W] are fully overwritten.  With n = 1 these are the
 * single-view entry points above.  d_allmap may be NULL when the five map gradients (everything but g_image) are NULL: the
 * seven planes would be zeros, and lara2dgs_backward* takes NULL for exactly that. */
int lara_surface_maps_forward_views(int32_t n_views, int32_t H, int32_t W, const float *color, const float *allmap,
                                    const float *rays, const float *rots, float depth_ratio, float *image, float *depth,
                                    float *acc_map, float *rend_normal, float *depth_normal, float *rend_dist, void *stream);

int lara_surface_maps_backward_views(int32_t n_views, int32_t H, int32_t W, const float *color, const float *allmap,
                                     const float *rays, const float *rots, float depth_ratio, const float *g_image,
                                     const float *g_depth, const float *g_acc_map, const float *g_rend_normal,
                                     const float *g_depth_normal, const float *g_rend_dist, float *d_color, float *d_allmap,
                                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_SURFACE_H */
