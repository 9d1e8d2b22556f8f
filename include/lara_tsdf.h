/*
 * lara_tsdf.h -- TSDF fusion of rendered depth / colour maps on the device (part of liblara2dgs.so).
 * SURVEY.md section 8f row 4 ("then GPU TSDF fusion to replace Open3D").
 *
 * Replaces the `volume.integrate(rgbd, intrinsic, extrinsic)` loop of `MeshExtractor.extract`
 * (tools/meshExtractor.py:67-110), which hands every one of the 48 rendered views to Open3D's CPU
 * `ScalableTSDFVolume` through two device->host copies and three numpy conversions per view.  Open3D is a third-party
 * dependency that is ABSENT from /root/reference and from this image (pip `open3d`, version not pinned by the
 * reference: requirements list it without one); the arithmetic below restates its published per-voxel update
 * (Open3D `UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier`, which `ScalableTSDFVolume` applies to its
 * 16^3 blocks) -- parity for this row is against our CPU restatement (oracle/tsdf_ref.py), UNPINNED by the reference.
 *
 * Volume: a dense res^3 grid with voxel centres at origin + voxel_length * (idx + 0.5) (Open3D's convention), holding
 * tsdf (fp32, initial 0), weight (fp32, initial 0) and colour (3 x fp32); laid out [x][y][z] like Open3D's
 * UniformTSDFVolume (index = (x * res + y) * res + z).  One launch integrates n_views views in order (the running
 * averages are order dependent; the order is the caller's, as with successive `integrate` calls):
 *   p_c = E[v] p;  skip if p_c.z <= 0;  (u_f, v_f) = (p_c.x fx / p_c.z + cx + 0.5, p_c.y fy / p_c.z + cy + 0.5);
 *   skip unless 0.0001 <= u_f < W - 0.0001 (same for v_f);  d = depth[(int)v_f][(int)u_f];
 *   skip if d <= 0 or d > depth_trunc[v]  (RGBDImage.create_from_color_and_depth zeroes depths beyond depth_trunc);
 *   sdf = (d - p_c.z) * sqrt(((u - cx)/fx)^2 + ((v - cy)/fy)^2 + 1);  skip if sdf <= -sdf_trunc;
 *   t = min(1, sdf / sdf_trunc);  tsdf = (tsdf w + t) / (w + 1);  colour likewise;  w += 1.
 * depth [n_views][H][W] fp32 (the caller zeroes alpha < alpha_thres pixels, meshExtractor.py:92); colour
 * [n_views][H][W][3] fp32 in 0..255 (the reference quantises to uint8 first, :99); intrinsics [n_views][4] = fx, fy,
 * cx, cy; extrinsics [n_views][16] row-major world->camera (= world_view_transform^T, :107); depth_trunc [n_views].
 * All pointers are device pointers; work is enqueued on `stream`.  Returns 0 or a negative LARA2DGS_E_* code.
 */
#ifndef LARA_TSDF_H
#define LARA_TSDF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int lara_tsdf_integrate(int32_t res, const float *origin /* host, [3] */, float voxel_length, float sdf_trunc,
                        int32_t n_views, int32_t H, int32_t W, const float *depth, const float *color,
                        const float *intrinsics, const float *extrinsics, const float *depth_trunc, float *tsdf,
                        float *weight, float *rgb, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_TSDF_H */
