/*
 * lara_rays.h -- device-side camera-ray generation for LaRa's batch dictionary (part of
 * liblara2dgs.so).  SURVEY.md section 8f row 3.
 *
 * Replaces `build_rays(c2ws, ixts, H, W, scale)` of the reference's data loader
 * (dataLoader/utils.py:21-34, called twice per scene at dataLoader/gobjverse.py:92-95 for
 * `tar_rays` (scale 1) and `tar_rays_down` (scale 1/16)), which runs on the CPU workers and ships
 * 6 floats per pixel per view through the DataLoader; here the cameras are device tensors and the
 * rays are written where they are consumed (lightning/network.py:364,493,522;
 * renderer_2dgs.py:78-89).
 *
 *   Hs = int(H * scale), Ws = int(W * scale) -- computed ONCE, by the caller, exactly as the reference
 *   does in Python (double precision), and passed in: the library never re-derives the output size;
 *   `scale` only rescales the intrinsics:  K_s = diag(scale, scale, 1) * K
 *   rays[v, y, x, 0:3] = c2w[v][0:3, 3]
 *   rays[v, y, x, 3:6] = R_v * K_s^-1 * (x + 0.5, y + 0.5, 1)^T            (R_v = c2w[v][0:3, 0:3])
 *
 * c2ws [n_views, 4, 4], ixts [n_views, 3, 3] (neither is modified: the reference scales `ixts` in
 * place and its callers pass copies), rays [n_views, Hs, Ws, 6]; fp32, row-major, device pointers.
 * Returns 0 or a negative LARA2DGS_E_* code; work is enqueued on `stream`, no host synchronisation.
 *
 * ABI note: until ABI version 4 this entry point was `lara_build_rays(n_views, H, W, scale, ...)` taking the INPUT
 * size.  The output-size form has the same C signature, so it carries a new name: a caller built against the old
 * header fails at link / symbol lookup instead of silently producing full-resolution maps with scaled intrinsics.
 */
#ifndef LARA_RAYS_H
#define LARA_RAYS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int lara_build_rays_out(int32_t n_views, int32_t Hs, int32_t Ws, float scale, const float *c2ws,
                    const float *ixts, float *rays, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_RAYS_H */
