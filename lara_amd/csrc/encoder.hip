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

// ---- 3x3x3 convolution, 256 x 256 tile, LDS-DMA ring -------------------------------------------------
// C[M,256] = LN3(x) + sum over 27 taps of A_tap[M,256] . W_tap[256,256]^T, as in mfma_gemm.h (AMODE 1,
// EPI 4), restructured around what limited that kernel: register staging put as
// many LDS cycles into ds_write_b128 (13 cycles per wave instruction) as into fragment reads, and kept
// the operand re-reads through L2 at two passes over A.  Here
//   * the whole N = 256 sits in one workgroup (8 waves as 2 x 4, wave tile 128 x 64), so the gathered
//     operand is fetched once per tap;
//   * K tiles (32 channels of one tap: A 256 rows x 64 B, W 256 rows x 64 B = 32 KB) travel global ->
//     LDS with global_load_lds_dwordx4: no staging registers, no ds_write; every wave issues 4 of the 32
//     one-KB pieces of a tile;
//   * a ring of four tile buffers keeps three tiles in flight: iteration kt waits for its own four
//     pieces of tile kt with a COUNTED vmcnt (8 younger pieces stay in flight), one barrier makes
//     everybody's pieces visible and retires the reads of tile kt-1, whose buffer is then refilled
//     with tile kt+3;
//   * LDS rows are 64 B, unpadded (the DMA writes lane-linear); the 16-byte chunk index is XORed with
//     (row >> 2) & 3 on the SOURCE address and on the fragment reads, which spreads the rows a
//     ds_read_b128 lane group touches over all four chunk slots.
constexpr int RT = 256;             // tile rows (M) and columns (N)
constexpr int RING = 4;             // tile buffers
constexpr int RTILE = 2 * RT * 64;  // bytes per K tile: A panel + W panel

__global__ void __launch_bounds__(512)
conv3d_ring_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // RING * RTILE bytes
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int M = p.M, K = p.K, Cin = p.Cin;
    const int bm0 = blockIdx.x * RT;
    const int wr = wave >> 2, wc = wave & 3;  // wave tile: rows wr*128.., columns wc*64..
    const int r = lane & 31, kh = lane >> 5;
    const char *Ab = (const char *)p.A, *Wb = (const char *)p.W;

    // staging: wave w moves pieces {2w, 2w+1} of the A panel and of the W panel; piece i = rows 16 i ..
    // 16 i + 15, lane L lands in row 16 i + L/4, physical chunk L & 3, and must therefore FETCH the
    // logical chunk (L & 3) ^ ((row >> 2) & 3)
    int srow[2];
    uint32_t woff[2], noff[2];
    int vb[2], vd[2], vh[2], vw[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        srow[q] = 32 * wave + 16 * q + (lane >> 2);
        const int chunk = (lane & 3) ^ ((srow[q] >> 2) & 3);
        woff[q] = (uint32_t)srow[q] * (uint32_t)(K * 2) + chunk * 16;  // N = 256 rows of W exactly
        token_to_voxel(min(bm0 + srow[q], M - 1), p.R, vb[q], vd[q], vh[q], vw[q]);
        noff[q] = chunk * 16;
    }
    const int ktiles = K / 32, kpt = Cin / 32;
    int cur_tap = -1;
    auto issue = [&](const int kt) {  // this wave's four pieces of K tile kt
        const int tap = kt / kpt, kc = kt - tap * kpt;
        if (tap != cur_tap) {
            cur_tap = tap;
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int nd = vd[q] + dz, nh = vh[q] + dy, nw = vw[q] + dx;
                const bool in = (unsigned)nd < (unsigned)p.R && (unsigned)nh < (unsigned)p.R && (unsigned)nw < (unsigned)p.R;
                const int chunk = (lane & 3) ^ ((srow[q] >> 2) & 3);
                noff[q] = (in ? (uint32_t)voxel_to_token(vb[q], nd, nh, nw, p.R) * (uint32_t)(Cin * 2) : p.zero_off) + chunk * 16;
            }
        }
        unsigned char *buf = ring + (kt % RING) * RTILE;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ab + (noff[q] + (uint32_t)kc * 64)),
                                             (__attribute__((address_space(3))) void *)(buf + (2 * wave + q) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Wb + (woff[q] + (uint32_t)kt * 64)),
                                             (__attribute__((address_space(3))) void *)(buf + RT * 64 + (2 * wave + q) * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // fragment addresses inside a tile buffer (row * 64 + swizzled chunk * 16), for K steps 0 and 1
    int aoff[4][2], boff[2][2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = wr * 128 + 32 * i + r;
            aoff[i][s] = row * 64 + (((2 * s + kh) ^ ((row >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int row = wc * 64 + 32 * j + r;
            boff[j][s] = RT * 64 + row * 64 + (((2 * s + kh) ^ ((row >> 2) & 3)) << 4);
        }
    }

    issue(0);
    if (1 < ktiles) issue(1);
    if (2 < ktiles) issue(2);
    for (int kt = 0; kt < ktiles; kt++) {
        // my four pieces of tile kt have landed once at most the pieces of the tiles issued after it
        // (two tiles = 8 pieces, fewer at the tail) are still in flight
        const int younger = min(ktiles - 1 - kt, 2);
        if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // all pieces of tile kt visible; all reads of tile kt-1 retired
        if (kt + 3 < ktiles) issue(kt + 3);
        const unsigned char *buf = ring + (kt % RING) * RTILE;
        bf16x8 a[2][4], b[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
#pragma unroll
            for (int i = 0; i < 4; i++) a[s2][i] = *(const bf16x8 *)(buf + aoff[i][s2]);
#pragma unroll
            for (int j = 0; j < 2; j++) b[s2][j] = *(const bf16x8 *)(buf + boff[j][s2]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s2][i], b[s2][j], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment reads done before the next barrier
    }
    __syncthreads();

    // epilogue: out = LN3(x)[row] (redone in fp32 from the row statistics) + acc, through LDS so that
    // the loads of x and the stores are 16 bytes per lane
    float *ep = (float *)ring + wave * (32 * 68);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ep[((e & 3) + 8 * (e >> 2) + 4 * kh) * 68 + j * 32 + r] = acc[i][j][e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = q * 64 + lane, lr = idx >> 4, c4 = (idx & 15) * 4;
            const int row = bm0 + wr * 128 + i * 32 + lr, col = wc * 64 + c4;
            if (row < M) {
                const float4 v = *(const float4 *)(ep + lr * 68 + c4);
                const size_t o = (size_t)row * 256 + col;
                const float4 rs = *(const float4 *)(p.resid + o);
                const float2 st = p.stats[row];
                const float4 ga = *(const float4 *)(p.gamma + col), be = *(const float4 *)(p.beta + col);
                *(float4 *)((float *)p.C + o) =
                    make_float4(v.x + (rs.x - st.x) * st.y * ga.x + be.x, v.y + (rs.y - st.x) * st.y * ga.y + be.y,
                                v.z + (rs.z - st.x) * st.y * ga.z + be.z, v.w + (rs.w - st.x) * st.y * ga.w + be.w);
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t lara_groupblock_workspace_bytes(int32_t scenes, int32_t R) {
    if (scenes < 0 || R <= 0 || (R & 1)) return LARA2DGS_E_INVALID;
    const int64_t M = (int64_t)scenes * R * R * R;
    // xn | q | o | kv (bf16 [M,256] each; q|o doubles as the MLP hidden [M,512]) + (mean, rstd) per row
    // + one zeroed row (the convolution's padding voxels)
    return M * 256 * 2 * 4 + M * 8 + 512 + 1024;
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
    char *zero_row = (char *)workspace + (size_t)M * 256 * 2 * 4 + (size_t)M * 8;
    // 2. MLP
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
    L2D_CHECK_LAUNCH();
    if (hipMemsetAsync(zero_row, 0, 512, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    {
        L2D_PROF("gb_conv3d", s);
        GemmP p{};
        p.A = xn; p.W = w->wconv; p.C = x; p.resid = x; p.M = M; p.N = 256; p.K = 27 * 256;
        p.R = R; p.Cin = 256; p.stats = stats; p.gamma = w->ln3_w; p.beta = w->ln3_b;
        p.zero_off = (uint32_t)(zero_row - (char *)xn);
        static const int ring_conv = getenv("LARA_CONV_RING") ? atoi(getenv("LARA_CONV_RING")) : 1;
        if (ring_conv) {
            static const hipError_t attr = hipFuncSetAttribute((const void *)conv3d_ring_kernel,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, RING * RTILE);
            if (attr != hipSuccess) return LARA2DGS_E_LAUNCH;
            hipLaunchKernelGGL(conv3d_ring_kernel, dim3((M + 255) / 256), dim3(512), RING * RTILE, s, p);
        } else {
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, 4, 256, 128>), dim3((M + 255) / 256, 2), dim3(256), 0, s, p);
        }
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
