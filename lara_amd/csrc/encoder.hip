// encoder.hip -- the rest of LaRa's GroupAttBlock (lightning/network.py:57-102) and the
// VolTransformer tail (network.py:156-163) on gfx950 matrix cores: rows A2 / A3 of SURVEY.md s.8.
//
//   x = x + cross_attn(norm1(x), cond, cond)        attention.hip (lara_groupattn_forward)
//   x = x + mlp(norm2(x))                            LN+cast, GEMM(+bias, GELU), GEMM(+bias, +residual)
//   x = norm3(x);  x = x + cnn(x)                    LN+cast (+row stats), implicit-GEMM 3x3x3 conv whose
//                                                    epilogue redoes the LayerNorm of its own row in fp32
//   out = deconv(norm(x))                            LN+cast, GEMM with a stride-2 scatter epilogue
//
// The activations stay in the attention's group-major token order for all twelve layers: the
// convolution gathers its 27 neighbours through that order (mfma_gemm.h, AMODE 1), so the
// volume <-> patches permutes the reference pays twice per layer (network.py:82-86, 96-98) vanish.
#include <cstdlib>

#include "mfma_gemm.h"
#include "mlp_fused.h"
#include "../../include/lara_groupattn.h"

namespace {

__global__ void __launch_bounds__(256)
tokens_volume_kernel(const int R, const int C, const float *__restrict__ src, float *__restrict__ dst,
                     const int M, const int to_tokens) {
    // one thread per (token, channel); token rows are the contiguous side
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * C) return;
    const int m = (int)(i / C), c = (int)(i - (size_t)m * C);
    int b, d, h, w;
    token_to_voxel(m, R, b, d, h, w);
    const size_t v = ((((size_t)b * C + c) * R + d) * R + h) * R + w;
    if (to_tokens) dst[i] = src[v];
    else dst[v] = src[i];
}

// dst[b][c][r] = src[b][r][c]: 64 x 64 tiles through LDS, both sides in 256-byte (fp32) / 128-byte (bf16) runs
template <bool BF16>
__global__ void __launch_bounds__(256)
batched_transpose_kernel(const float *__restrict__ src, void *__restrict__ dst, const int rows, const int cols) {
    __shared__ float t[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const size_t base = (size_t)blockIdx.z * rows * cols;
#pragma unroll 4
    for (int k = 0; k < 16; k++) {
        const int r = r0 + ty * 16 + k, c = c0 + tx;
        t[ty * 16 + k][tx] = (r < rows && c < cols) ? src[base + (size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; k++) {
        const int c = c0 + ty * 16 + k, r = r0 + tx;
        if (c < cols && r < rows) {
            const float v = t[tx][ty * 16 + k];
            if (BF16) ((unsigned short *)dst)[base + (size_t)c * rows + r] = f2bf(v);
            else ((float *)dst)[base + (size_t)c * rows + r] = v;
        }
    }
}

}  // namespace

extern "C" {

int lara_batched_transpose(int32_t batch, int32_t rows, int32_t cols, const float *src, void *dst, int32_t dst_bf16,
                           void *stream) {
    if (batch < 0 || rows < 0 || cols < 0 || batch > 65535 || (rows + 63) / 64 > 65535) return LARA2DGS_E_INVALID;
    if (batch == 0 || rows == 0 || cols == 0) return LARA2DGS_OK;
    if (!src || !dst) return LARA2DGS_E_INVALID;
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
    if (dst_bf16) hipLaunchKernelGGL(batched_transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
    else hipLaunchKernelGGL(batched_transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int64_t lara_groupblock_workspace_bytes(int32_t scenes, int32_t R) {
    if (scenes < 0 || R <= 0 || (R & 1)) return LARA2DGS_E_INVALID;
    const int64_t M = (int64_t)scenes * R * R * R;
    // xn | q | o | kv (bf16 [M,256] each; q|o doubles as the MLP hidden [M,512]) + (mean, rstd) per row
    // + one zeroed row (the convolution's padding voxels); + 256 KB: the attention step's fused path keeps its packed weights
    // in front of its regions (lara_groupattn_workspace_bytes)
    return M * 256 * 2 * 4 + M * 8 + 512 + 1024 + 262144;
}

int lara_groupblock_forward(int32_t scenes, int32_t R, int32_t cond_dim, float *x,
                            const uint16_t *cond_bf16, const lara_groupblock_weights *w,
                            void *workspace, void *stream) {
    if (scenes < 0 || R <= 0 || (R & 1) || cond_dim <= 0 || (cond_dim % 32) != 0 || !w) return LARA2DGS_E_INVALID;
    if (scenes == 0) return LARA2DGS_OK;
    if (!x || !cond_bf16 || !workspace || !w->ln1_w || !w->ln1_b || !w->wq || !w->wkv || !w->wo || !w->ln2_w ||
        !w->ln2_b || !w->w1 || !w->b1 || !w->w2 || !w->b2 || !w->ln3_w || !w->ln3_b || !w->wconv)
        return LARA2DGS_E_INVALID;
    const int64_t M64 = (int64_t)scenes * R * R * R;
    if (M64 * 2056 + 512 >= (1ll << 32)) return LARA2DGS_E_INVALID;  // kernels address the workspace with 32-bit offsets
    const int M = (int)M64, G = M / 8;
    hipStream_t s = (hipStream_t)stream;
    // 1. attention step, in place (its workspace is the first 4 * M * 512 bytes of ours)
    int rc = lara_groupattn_forward(G, cond_dim, x, cond_bf16, w->ln1_w, w->ln1_b, w->eps, w->wq, w->wkv, w->wo,
                                    x, workspace, stream);
    if (rc) return rc;
    unsigned short *xn = (unsigned short *)workspace;
    unsigned short *hid = xn + (size_t)M * 256;  // [M, 512]
    float2 *stats = (float2 *)((char *)workspace + (size_t)M * 256 * 2 * 4);
#ifndef LARA_MLP_UNFUSED
    char *zero_row = (char *)xn + (size_t)M * 512;      // row M of xn: written by the fused MLP kernel (the hidden tensor's old place)
#else
    char *zero_row = (char *)workspace + (size_t)M * 256 * 2 * 4 + (size_t)M * 8;
#endif
    // 2. MLP + norm3: ONE kernel per 128-row tile (mlp_fused.h); the hidden tensor never leaves the CU
#ifndef LARA_MLP_UNFUSED
    (void)hid;
    {
        L2D_PROF("gb_mlp_fused", s);
        MlpP p{};
        p.x1 = x; p.x2 = x; p.ln2_w = w->ln2_w; p.ln2_b = w->ln2_b; p.b1 = w->b1; p.b2 = w->b2; p.ln3_w = w->ln3_w; p.ln3_b = w->ln3_b;
        p.w1 = w->w1; p.w2 = w->w2; p.xn3 = xn; p.stats = stats; p.eps = w->eps; p.M = M;
        if (launch_mlp_fused<0>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
#else       // (rounds 2-5: four launches; tools/build_variant.sh -DLARA_MLP_UNFUSED for A/B runs)
    {
        L2D_PROF("gb_ln2", s);
        hipLaunchKernelGGL(ln_cast_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, w->ln2_w, w->ln2_b, w->eps, xn,
                           (float2 *)nullptr, M);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("gb_mlp1", s);
        GemmP p{};
        p.A = xn; p.W = w->w1; p.C = hid; p.bias = w->b1; p.M = M; p.N = 512; p.K = 256;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 2>), dim3((M + 127) / 128, 4), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("gb_mlp2", s);
        GemmP p{};
        p.A = hid; p.W = w->w2; p.C = x; p.resid = x; p.bias = w->b2; p.M = M; p.N = 256; p.K = 512;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 3>), dim3((M + 127) / 128, 2), dim3(256), 0, s, p);
    }
    L2D_CHECK_LAUNCH();
    // 3. norm3 + convolution + residual on the normalised activations
    {
        L2D_PROF("gb_ln3", s);
        hipLaunchKernelGGL(ln_cast_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, w->ln3_w, w->ln3_b, w->eps, xn,
                           stats, M);
    }
#endif
    L2D_CHECK_LAUNCH();
#ifdef LARA_MLP_UNFUSED
    if (hipMemsetAsync(zero_row, 0, 512, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
#endif      // (the fused kernel zero-fills the row behind xn's last: mlp_fused.h)
    {
        L2D_PROF("gb_conv3d", s);
        GemmP p{};
        p.A = xn; p.W = w->wconv; p.C = x; p.resid = x; p.M = M; p.N = 256; p.K = 27 * 256;
        p.R = R; p.Cin = 256; p.stats = stats; p.gamma = w->ln3_w; p.beta = w->ln3_b;
        p.zero_off = (uint32_t)(zero_row - (char *)xn);
        if (launch_gemm_ring<1, 4>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_voltrans_head_forward(int32_t scenes, int32_t R, const float *x, const float *ln_w,
                               const float *ln_b, float eps, const uint16_t *wdeconv,
                               const float *bias, int32_t Cout, float *out, void *workspace,
                               void *stream) {
    if (scenes < 0 || R <= 0 || (R & 1) || Cout <= 0 || (Cout & 3)) return LARA2DGS_E_INVALID;
    if (scenes == 0) return LARA2DGS_OK;
    if (!x || !ln_w || !ln_b || !wdeconv || !bias || !out || !workspace) return LARA2DGS_E_INVALID;
    const int M = scenes * R * R * R;
    hipStream_t s = (hipStream_t)stream;
    unsigned short *xn = (unsigned short *)workspace;
    {
        L2D_PROF("vt_ln", s);
        hipLaunchKernelGGL(ln_cast_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, ln_w, ln_b, eps, xn,
                           (float2 *)nullptr, M);
    }
    L2D_CHECK_LAUNCH();
    {
        L2D_PROF("vt_deconv", s);
        GemmP p{};
        p.A = xn; p.W = wdeconv; p.C = out; p.bias = bias; p.M = M; p.N = 8 * Cout; p.K = 256;
        p.R = R; p.Cout = Cout;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 5>), dim3((M + 127) / 128, (8 * Cout + 127) / 128), dim3(256), 0,
                           s, p);
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

static int tokens_volume(int32_t scenes, int32_t R, int32_t C, const float *src, float *dst, void *stream,
                         int to_tokens) {
    if (scenes < 0 || R <= 0 || (R & 1) || C <= 0) return LARA2DGS_E_INVALID;
    if (scenes == 0) return LARA2DGS_OK;
    if (!src || !dst) return LARA2DGS_E_INVALID;
    const int M = scenes * R * R * R;
    const size_t n = (size_t)M * C;
    hipLaunchKernelGGL(tokens_volume_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, C,
                       src, dst, M, to_tokens);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_tokens_from_volume(int32_t scenes, int32_t R, int32_t C, const float *volume, float *tokens,
                            void *stream) {
    return tokens_volume(scenes, R, C, volume, tokens, stream, 1);
}

int lara_volume_from_tokens(int32_t scenes, int32_t R, int32_t C, const float *tokens, float *volume,
                            void *stream) {
    return tokens_volume(scenes, R, C, tokens, volume, stream, 0);
}

}  // extern "C"
