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

/* Block-sparse integration: the semantics of Open3D's `ScalableTSDFVolume` (the class tools/meshExtractor.py:67 instantiates;
 * [RECALLED], Open3D absent).  The volume is cut into 16^3-voxel blocks (Open3D: volume units, volume_unit_resolution 16).  Per
 * view, every `depth_sampling_stride`-th pixel (Open3D default 4) with a valid depth is back-projected through `cam_to_world`
 * ([n_views][16], the inverse of `extrinsics`) and the blocks within +- sdf_trunc of the point (per axis) are TOUCHED; a view is
 * integrated only into the blocks it touched -- free space in front of the surface is never allocated, unlike the dense
 * `lara_tsdf_integrate`, which folds a view into every voxel whose projection has a depth.  The per-voxel update is the same code
 * in both: where both integrate a view they produce the same bits.  res % 16 == 0; for Open3D's unit grid origin / (16 *
 * voxel_length) must be integral.  Storage stays the dense [x][y][z] arrays (HBM is cheap here; the WORK is sparse: one workgroup
 * per 256 voxels of a touched block); `touched` [n_views][(res/16)^3] bytes is scratch (overwritten), `allocated` [(res/16)^3]
 * bytes accumulates the blocks ever touched (caller zero-fills once).  n_views <= 64 per call. */
int lara_tsdf_integrate_blocks(int32_t res, const float *origin /* host, [3] */, float voxel_length, float sdf_trunc,
                               int32_t n_views, int32_t H, int32_t W, int32_t depth_sampling_stride, const float *depth,
                               const float *color, const float *intrinsics, const float *extrinsics, const float *cam_to_world,
                               const float *depth_trunc, float *tsdf, float *weight, float *rgb, uint8_t *touched,
                               uint8_t *allocated, void *stream);

/* Mesh extraction (`volume.extract_triangle_mesh()`, meshExtractor.py:110): marching cubes over the cells whose 8 corner voxels
 * were all observed (weight > 0), corners at the voxel centres, a vertex at the linear zero crossing of tsdf along a cell edge,
 * vertex colour interpolated the same way (0..1).  The case table (csrc/mc_tables.h) is derived by tools/gen_mc_tables.py and is
 * watertight in the ambiguous cases.  Two launches around a prefix sum the caller owns:
 *   lara_tsdf_mesh_count -> counts [res^3] int32: triangles of cell (x,y,z) at index (x*res + y)*res + z (caller zero-fills:
 *                           cells of blocks with allocated[block] == 0 are not visited; allocated may be NULL = visit all);
 *   ends = inclusive prefix sum of counts (int64), T = ends[last];
 *   lara_tsdf_mesh_emit  -> vertices [T][3][3], colors [T][3][3] fp32 and edge_keys [T][3] int64: the id of the grid edge a
 *                           vertex lies on (equal keys = the same vertex: the caller welds with them). */
int lara_tsdf_mesh_count(int32_t res, const float *origin, float voxel_length, const float *tsdf, const float *weight,
                         const float *rgb, const uint8_t *allocated, int32_t *counts, void *stream);
int lara_tsdf_mesh_emit(int32_t res, const float *origin, float voxel_length, const float *tsdf, const float *weight,
                        const float *rgb, const uint8_t *allocated, const int32_t *counts, const int64_t *ends, float *vertices,
                        float *colors, int64_t *edge_keys, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LARA_TSDF_H */
