// rays.hip -- camera rays of LaRa's batch dictionary, generated on the device.
// Reference: build_rays (dataLoader/utils.py:21-34); see include/lara_rays.h for the contract.
// One thread per pixel; a workgroup covers 256 consecutive pixels of a row-major view and writes its
// 6 KB through LDS as 16-byte stores.  Pure streaming store: HBM-write-bound.
#include "common.h"
#include "../../include/lara_rays.h"

namespace {

__global__ void __launch_bounds__(256)
build_rays_kernel(const int Hs, const int Ws, const float scale, const float *__restrict__ c2ws,
                  const float *__restrict__ ixts, float *__restrict__ rays) {
    __shared__ __attribute__((aligned(16))) float tile[256 * 6];
    const int view = blockIdx.y;
    const int pix0 = blockIdx.x * 256, pix = pix0 + threadIdx.x, npix = Hs * Ws;
    const float *c = c2ws + view * 16, *k = ixts + view * 9;
    // K_s = diag(scale, scale, 1) K, inverted through the adjugate (scalar loads: uniform per view)
    const float a00 = k[0] * scale, a01 = k[1] * scale, a02 = k[2] * scale;
    const float a10 = k[3] * scale, a11 = k[4] * scale, a12 = k[5] * scale;
    const float a20 = k[6], a21 = k[7], a22 = k[8];
    const float c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const float c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const float inv_det = 1.0f / (a00 * c00 + a01 * c10 + a02 * c20);
    const int y = pix / Ws, x = pix - y * Ws;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const float dx = (c00 * px + c01 * py + c02) * inv_det;
    const float dy = (c10 * px + c11 * py + c12) * inv_det;
    const float dz = (c20 * px + c21 * py + c22) * inv_det;
    // 24 bytes per pixel: bounce the workgroup's 256 pixels through LDS so that the global stores are
    // 16 bytes per lane and contiguous across the wave
    float2 *t2 = (float2 *)(tile + threadIdx.x * 6);
    t2[0] = make_float2(c[3], c[7]);
    t2[1] = make_float2(c[11], c[0] * dx + c[1] * dy + c[2] * dz);
    t2[2] = make_float2(c[4] * dx + c[5] * dy + c[6] * dz, c[8] * dx + c[9] * dy + c[10] * dz);
    __syncthreads();
    const int nvalid = min(256, npix - pix0);          // pixels of this workgroup inside the view
    float *dst = rays + ((size_t)view * npix + pix0) * 6;  // 16-byte aligned: pix0 * 24 and npix * 24 are multiples of 16 when npix is even
    if (((size_t)view * npix * 6) % 4 == 0) {
        for (int i = threadIdx.x; i < nvalid * 6 / 4; i += 256) ((float4 *)dst)[i] = ((const float4 *)tile)[i];
        for (int i = (nvalid * 6 / 4) * 4 + threadIdx.x; i < nvalid * 6; i += 256) dst[i] = tile[i];
    } else {
        for (int i = threadIdx.x; i < nvalid * 6; i += 256) dst[i] = tile[i];
    }
}

}  // namespace

extern "C" int lara_build_rays_out(int32_t n_views, int32_t Hs, int32_t Ws, float scale, const float *c2ws,
                               const float *ixts, float *rays, void *stream) {
    // Hs, Ws are the OUTPUT size, computed once by the caller (the reference's int(H*scale) is a
    // double-precision product; recomputing it here in fp32 could disagree by one row)
    if (n_views < 0 || Hs < 0 || Ws < 0 || !(scale > 0.f)) return LARA2DGS_E_INVALID;
    if (n_views == 0 || Hs == 0 || Ws == 0) return LARA2DGS_OK;
    if ((int64_t)Hs * Ws > 0x7fffffff) return LARA2DGS_E_INVALID;
    if (!c2ws || !ixts || !rays || n_views > 65535) return LARA2DGS_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    {
        L2D_PROF("build_rays", s);
        hipLaunchKernelGGL(build_rays_kernel, dim3((Hs * Ws + 255) / 256, n_views), dim3(256), 0, s, Hs, Ws, scale,
                           c2ws, ixts, rays);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}
