/*
 * lara_pointfeat.h -- LaRa's fine-stage point sampler as one gather kernel per direction (part of
 * liblara2dgs.so).  SURVEY.md section 8f row 4.
 *
 * Replaces the body of `Network.get_point_feats` (lightning/network.py:390-411) after `points = points[mask]`:
 * `projection` (:182-187) of the n Gaussian centres into the V input views, `F.grid_sample` (bilinear, zeros
 * padding, align_corners=False) of the 8-channel stack [input image (3) | coarse render (3) | acc_map | depth]
 * at the projected positions, and the depth residual -- and, for training, its autograd backward (the
 * reference pays a [V,8,h,w] concatenation + permute, a [V,1,n,2] grid, the sampled [V,8,1,n] tensor and
 * their backward counterparts).
 *
 *   p_c  = w2c[v][:3,:3] p + w2c[v][:3,3];   q = K[v] p_c;   (x, y) = q.xy / q.z;   z = q.z
 *   s_c  = bilinear sample of channel c at pixel position (x, y)   [pixel i is centred at i: with
 *          grid = (xy + 0.5) / (w, h) * 2 - 1 and align_corners=False the sample position is exactly (x, y)]
 *   out[v][c][i] = s_c  (c = 0..6),   out[v][7][i] = | s_7 - z |
 *
 * points [n,3]; w2cs [V,4,4]; ixts [V,3,3]; img_ref [V,3,h,w] (planar, as `_inps[i]`); image [V,h,w,3],
 * acc_map [V,h,w], depth [V,h,w,1] (channel-last, as `render_img` returns them); out [V,8,n].  fp32, device.
 * Backward: g_out [V,8,n] -> d_points [n,3] (overwritten) and ACCUMULATES into d_image, d_acc_map, d_depth
 * (caller zero-fills; float atomics, like torch's grid_sample backward; any of the three may be NULL).
 * `workspace`: lara_point_feats_workspace_bytes(V, h, w) bytes (a packed channel-last copy of the four maps, 32
 * bytes per pixel, and in the backward its gradient); nothing is kept between calls.
 * Returns 0 or a negative LARA2DGS_E_* code; work is enqueued on `stream`, no host synchronisation.
 */
#ifndef LARA_POINTFEAT_H
#define LARA_POINTFEAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int64_t lara_point_feats_workspace_bytes(int32_t V, int32_t h, int32_t w);

int lara_point_feats_forward(int32_t n, int32_t V, int32_t h, int32_t w, const float *points, const float *w2cs,
                             const float *ixts, const float *img_ref, const float *image, const float *acc_map,
                             const float *depth, float *out, void *workspace, void *stream);

int lara_point_feats_backward(int32_t n, int32_t V, int32_t h, int32_t w, const float *points, const float *w2cs,
                              const float *ixts, const float *img_ref, const float *image, const float *acc_map,
                              const float *depth, const float *g_out, float *d_points, float *d_image,
                              float *d_acc_map, float *d_depth, void *workspace, void *stream);

/* The same with the three render-derived maps in the SIDE-BY-SIDE layout of the multi-view renderer
 * (`lara_surface_maps_forward_views`: image [h, row_views*w, 3], acc_map [h, row_views*w], depth [h, row_views*w, 1] --
 * what network.py:527 builds with torch.cat(dim=1)); the sampler reads views 0 .. V-1 of the `row_views` in a row
 * (network.py:499: the first n_views_sel).  No slicing / stacking copy in front, and the backward accumulates straight
 * into gradient maps of the full side-by-side shape (views >= V receive nothing).  row_views = 0: the [V,h,w,c] layout. */
int lara_point_feats_forward_concat(int32_t n, int32_t V, int32_t row_views, int32_t h, int32_t w, const float *points,
                                    const float *w2cs, const float *ixts, const float *img_ref, const float *image,
                                    const float *acc_map, const float *depth, float *out, void *workspace, void *stream);

int lara_point_feats_backward_concat(int32_t n, int32_t V, int32_t row_views, int32_t h, int32_t w, const float *points,
                                     const float *w2cs, const float *ixts, const float *img_ref, const float *image,
                                     const float *acc_map, const float *depth, const float *g_out, float *d_points,
                                     float *d_image, float *d_acc_map, float *d_depth, void *workspace, void *stream);

/* The fine stage's `x[mask]` rows (network.py:514-524: centres, SH, opacity, scaling, rotation of the surfels the mask keeps) as
 * ONE launch per direction instead of one index_select / index_copy_ per tensor.  `idx`: n int64 row indices on the device.
 *   scatter = 0 (forward):   items[k].dst[r][0..width) = items[k].src[idx[r]][0..width)         for r < n
 *   scatter = 1 (backward):  items[k].dst[idx[r]][0..width) = items[k].src[r][0..width)         (dst zero-filled by the caller;
 *                            the indices of a mask are unique, so this is a copy, not an accumulation)
 * All tensors fp32, row-major, rows of `width` floats; count <= LARA_ROWS_MAX; n * (sum of widths) < 2^31. */
#define LARA_ROWS_MAX 8
typedef struct {
    const float *src;
    float *dst;
    int32_t width;
} lara_rows_item;

int lara_take_rows(int32_t n, const int64_t *idx, int32_t count, const lara_rows_item *items, int32_t scatter, void *stream);

/* The fine stage's volume-feature rows (network.py:509: every kept Gaussian reads the feature row of its voxel,
 * `x.unsqueeze(1).expand(-1, K, -1)[mask.view(-1, K)]`).  `vox`: n ASCENDING int64 voxel indices on the device (mask indices / K:
 * the rows of one voxel are consecutive).  width % 4 == 0; rows fp32, 16-byte aligned.
 *   backward = 0:  dst[r][0..width) = src[vox[r]][0..width)                                       for r < n
 *   backward = 1:  dst[v][0..width) = sum over the rows r with vox[r] == v of src[r][0..width)   (in row order, no atomics;
 *                  voxels without a row are not written: the caller zero-fills dst) */
int lara_voxel_rows(int32_t n, int32_t width, const int64_t *vox, const float *src, float *dst, int32_t backward, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_POINTFEAT_H */
