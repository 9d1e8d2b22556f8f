// mfma_gemm.h -- bf16 MFMA building blocks shared by the attention and encoder-block kernels:
// LayerNorm+cast, and an LDS-tiled NT GEMM with fused epilogues / implicit 3x3x3 convolution.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// fp32 -> bf16, round to nearest even, on gfx950's v_cvt_pk_bf16_f32 (two values per instruction; the integer form
// u += 0x7fff + ((u >> 16) & 1) >> 16 is three half-rate instructions per value, and a kernel that rounds hundreds of values per lane
// -- the fused attention step: 512 -- spent 40 % of its vector instructions on it).  Same result for every finite input and inf.
// (through the compiler's own conversion, not inline asm: an asm statement that reads an MFMA accumulator is invisible to the
// hazard recognizer -- the first version did exactly that behind the attention's P.V product and produced NaNs)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f2bf2(float lo, float hi) {   // bf16(lo) | bf16(hi) << 16
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    unsigned u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)f2bf2(f, f); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// ---- LayerNorm(C = 256) + bf16 cast: one wave per token, 4 channels per lane ---------------------
__global__ void __launch_bounds__(256)
ln_cast_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
               const float *__restrict__ beta, const float eps, unsigned short *__restrict__ out,
               float2 *__restrict__ stats, const int tokens) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tok >= tokens) return;
    const float4 v = ((const float4 *)(x + (size_t)tok * 256))[lane];
    float s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s * (1.0f / 256.0f);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d4 = v.w - mean;
    float q = a * a + b * b + c * c + d4 * d4;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + eps);
    const float4 g = ((const float4 *)gamma)[lane], be = ((const float4 *)beta)[lane];
    ushort4 o;
    o.x = f2bf(a * rstd * g.x + be.x); o.y = f2bf(b * rstd * g.y + be.y);
    o.z = f2bf(c * rstd * g.z + be.z); o.w = f2bf(d4 * rstd * g.w + be.w);
    ((ushort4 *)(out + (size_t)tok * 256))[lane] = o;
    if (stats && lane == 0) stats[tok] = make_float2(mean, rstd);  // lets a later epilogue redo the LN in fp32
}

// ---- C[M,N] = A[M,K] . W[N,K]^T, bf16 in, fp32 accumulate ------------------------------------------
// Workgroup tile TM x 128 (TM = 128 or 256), K step 32, four waves as 2x2, wave tile (TM/2) x 64 =
// (TM/64) x 2 MFMA tiles of 32x32.  TM = 256 halves the operand re-reads through L2 (the implicit
// convolution at 128x128 moved 10.8 GB per launch through L2 for 0.46 TFLOP of work) and needs six
// instead of eight ds_read_b128 per eight MFMAs.
// Operand panels are staged through LDS (double buffered, register staging: the next K tile is
// loaded into VGPRs while the current one feeds the matrix cores).  Both operands have K contiguous,
// so a fragment is one ds_read_b128; LDS rows are padded from 64 to 80 bytes, which spreads the 16
// rows a ds_read_b128 lane group touches over all 16 sixteen-byte slots of the 256-byte bank row
// (conflict free).  MFMA 32x32x16 operand map: A[i = lane&31][k = 8*(lane>>5) + e],
// B[k = 8*(lane>>5) + e][j = lane&31]; C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
// AMODE 1 turns the A operand into the im2col view of a 3x3x3, padding-1 convolution without ever
// materialising it: activations are channels-last rows of a volume of edge R whose rows are ordered
// group-major (2x2x2 voxel blocks, the order the attention works in), K = 27 * Cin, and K tile kt
// reads channels [32 kc, 32 kc + 32) of the row of the voxel's (dz, dy, dx) neighbour (zeros outside
// the volume); W is the weight re-laid as [Cout][tap][Cin], i.e. still K-contiguous.
constexpr int GK = 32;
constexpr int LROW = GK * 2 + 16;  // padded LDS row, bytes

struct GemmP {
    const unsigned short *A, *W;
    void *C;
    const float *resid;  // EPI 1, 3, 4: fp32 [M, N]
    const float *bias;   // EPI 2, 3, 5: fp32 [N] (EPI 5: [Cout])
    int M, N, K;
    int R, Cin;          // AMODE 1 / EPI 5: volume edge, input channels
    const float2 *stats; // EPI 4: per-row (mean, rstd) of resid
    const float *gamma, *beta;
    int Cout;            // EPI 5
    uint32_t zero_off;   // AMODE 1: byte offset from A of a zeroed row of Cin bf16 (the padding voxels)
    unsigned short *C2;  // EPI 6: bf16 [M, N] pre-activation out; EPI 7: the same, read
    float *colsum;       // EPI 7 (gemm_bf16_nt_kernel, optional): [row tiles of 128][N] column sums of the stored bf16 values per row tile;
                         // EPI 9: [row tiles of 256][3][256] partial column sums (dgamma, dbeta, the bias gradient behind the LayerNorm)
    const float *lnx;    // EPI 9: fp32 [M, 256], the input of the LayerNorm whose backward the epilogue runs (stats = its mean, rstd)
};

// 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on v_exp / v_rcp
__device__ __forceinline__ float gelu_erf_grad(const float x) {  // Phi(x) + x phi(x)
    const float z = fabsf(x) * 0.70710678f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float ex = __expf(-z * z);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * ex, x));
    return cdf + x * ex * 0.3989422804f;
}
__device__ __forceinline__ float gelu_erf(const float x) {
    const float z = fabsf(x) * 0.70710678f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// row of the group-major token order <-> voxel (b, d, h, w); Gd = R / 2 blocks per edge
__device__ __forceinline__ void token_to_voxel(const int m, const int R, int &b, int &d, int &h, int &w) {
    const int Gd = R >> 1, l = m & 7;
    int g = m >> 3;
    const int gw = g % Gd; g /= Gd;
    const int gh = g % Gd; g /= Gd;
    const int gd = g % Gd; b = g / Gd;
    d = 2 * gd + (l >> 2); h = 2 * gh + ((l >> 1) & 1); w = 2 * gw + (l & 1);
}
__device__ __forceinline__ int voxel_to_token(const int b, const int d, const int h, const int w, const int R) {
    const int Gd = R >> 1;
    return ((((b * Gd + (d >> 1)) * Gd + (h >> 1)) * Gd + (w >> 1)) << 3) | ((d & 1) << 2) | ((h & 1) << 1) | (w & 1);
}

// Workgroup -> (row tile, column tile), XCD-aware.  Workgroup ids are dispatched round-robin over the 8 XCDs
// (id % 8), each with its own L2; the column tiles of one row tile read the same A rows.  With the launch's
// natural order (all row tiles of column 0, then column 1, ...) those re-reads are far apart in time and on other
// XCDs, i.e. they come from HBM again (xn2 four times for the 256 -> 512 MLP layer).  Here ids are dealt in groups
// of 8 row tiles x all column tiles: row tile r lands on XCD r % 8 for every column tile, 8 ids apart, so the
// re-reads hit that XCD's L2.  gridDim = (row tiles, column tiles) as before; a ragged last group keeps the bijection.
__device__ __forceinline__ void xcd_tile(int &rt, int &ct) {
    const int rtiles = gridDim.x, nc = gridDim.y;
    const int L = blockIdx.x + rtiles * blockIdx.y;
    const int full = (rtiles >> 3) << 3, base = full * nc;
    if (L < base) {
        const int q = L / (8 * nc), rem = L - q * (8 * nc);
        ct = rem >> 3;
        rt = q * 8 + (rem & 7);
    } else {
        const int t = rtiles - full, l = L - base;
        ct = l / t;
        rt = full + (l - ct * t);
    }
}

// EPI 0: bf16 store            1: fp32 store of acc + resid        2: bf16 store of gelu(acc + bias)
//     3: fp32 acc + bias + resid   4: fp32 LN(resid row) + acc (LN redone from stats, gamma, beta)
//     5: fp32 acc + bias scattered as a stride-2, kernel-2 transposed convolution (N = 8 * Cout)
// (training, gemm_bf16_nt_kernel only)
//     6: z = acc + bias -> bf16 C2, gelu(z) -> bf16 C       7: bf16 store of acc * gelu'(C2)
//     8: fp32 store
// The fused epilogues, applied to four consecutive columns (row, col .. col + 3) of the accumulator tile; `base_out`
// is EPI 5's output offset of the row's voxel (2d, 2h, 2w).  Shared by both GEMM kernels.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmP &p, float4 v, const int row, const int col, const int N,
                                              const unsigned long long base_out, float4 *stored = nullptr) {
    const size_t o = (size_t)row * N + col;
    if (EPI == 2 || EPI == 3 || EPI == 6) {
        const float4 bs = *(const float4 *)(p.bias + col);
        v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
    }
    if (EPI == 0 || EPI == 2 || EPI == 6 || EPI == 7) {
        if (EPI == 6) {
            ushort4 z;
            z.x = f2bf(v.x); z.y = f2bf(v.y); z.z = f2bf(v.z); z.w = f2bf(v.w);
            *(ushort4 *)(p.C2 + o) = z;
        }
        if (EPI == 7) {
            const ushort4 z = *(const ushort4 *)(p.C2 + o);
            v.x *= gelu_erf_grad(bf2f(z.x)); v.y *= gelu_erf_grad(bf2f(z.y));
            v.z *= gelu_erf_grad(bf2f(z.z)); v.w *= gelu_erf_grad(bf2f(z.w));
        }
        if (EPI == 2 || EPI == 6) {  // erf GELU (nn.GELU's default); the result is rounded to bf16 anyway
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
        }
        ushort4 h;
        h.x = f2bf(v.x); h.y = f2bf(v.y); h.z = f2bf(v.z); h.w = f2bf(v.w);
        *(ushort4 *)((unsigned short *)p.C + o) = h;
        if (stored) *stored = make_float4(bf2f(h.x), bf2f(h.y), bf2f(h.z), bf2f(h.w));     // (what a later column sum of C would add up)
    } else if (EPI == 8) {
        *(float4 *)((float *)p.C + o) = v;
    } else if (EPI == 1 || EPI == 3) {
        const float4 rs = *(const float4 *)(p.resid + o);
        *(float4 *)((float *)p.C + o) = make_float4(v.x + rs.x, v.y + rs.y, v.z + rs.z, v.w + rs.w);
    } else if (EPI == 4) {
        const float4 rs = *(const float4 *)(p.resid + o);
        const float2 st = p.stats[row];
        const float4 ga = *(const float4 *)(p.gamma + col), be = *(const float4 *)(p.beta + col);
        *(float4 *)((float *)p.C + o) =
            make_float4(v.x + (rs.x - st.x) * st.y * ga.x + be.x, v.y + (rs.y - st.x) * st.y * ga.y + be.y,
                        v.z + (rs.z - st.x) * st.y * ga.z + be.z, v.w + (rs.w - st.x) * st.y * ga.w + be.w);
    } else {  // EPI 5: column = tap * Cout + co, tap = (i*2 + j)*2 + k of the 2x2x2 kernel
        const int tap = col / p.Cout, co = col - tap * p.Cout;
        const size_t R2 = 2 * (size_t)p.R;
        const size_t oo = (base_out + ((size_t)(tap >> 2) * R2 + ((tap >> 1) & 1)) * R2 + (tap & 1)) * p.Cout + co;
        const float4 bs = *(const float4 *)(p.bias + co);
        *(float4 *)((float *)p.C + oo) = make_float4(v.x + bs.x, v.y + bs.y, v.z + bs.z, v.w + bs.w);
    }
}

template <int AMODE, int EPI, int GM = 128, int GN = 128>
__global__ void __launch_bounds__(256)
gemm_bf16_nt_kernel(const GemmP p) {
    constexpr int MI = GM / 64;         // 32-row MFMA tiles per wave along M
    constexpr int NJ = GN / 64;         // 32-column MFMA tiles per wave along N
    constexpr int NA = GM / 64;         // staging passes that carry A rows
    constexpr int NS = (GM + GN) / 64;  // staging passes per K tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][(GM + GN) * LROW];
    const unsigned short *__restrict__ A = p.A, *__restrict__ W = p.W;
    const int M = p.M, N = p.N, K = p.K;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int rt_, ct_;
    xcd_tile(rt_, ct_);
    const int bm0 = rt_ * GM, bn0 = ct_ * GN;
    const int wm = (wave >> 1) * (GM / 2), wn = (wave & 1) * (GN / 2);
    const int r = lane & 31, kh = lane >> 5;

    // staging map: GM + 128 rows (A rows, then W rows) x 4 sixteen-byte chunks per K tile; thread t
    // moves chunk (t & 3) of rows (t >> 2) + 64 i, i = 0..NS-1 (i < NA: A rows, else W rows)
    // rows are dealt so that the 16 lanes of a store group hit rows {0, 4, 8, 12} + c: with 80-byte rows
    // those start 0, 64, 128, 192 bytes into the 256-byte bank row (rows 0..3 would wrap onto each other)
    const int sq = tid >> 2, schunk = tid & 3;
    const int srow = (sq & ~15) | ((sq & 3) << 2) | ((sq >> 2) & 3);
    // global addresses are a uniform base + a 32-bit byte offset per thread (operands are < 4 GB): one
    // v_add per load in the main loop instead of 64-bit pointer arithmetic
    const char *Ab = (const char *)A, *Wb = (const char *)W;
    uint32_t goff[NS];
    int loff[NS];
    int vb[NA], vd[NA], vh[NA], vw[NA];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const int row = srow + 64 * i;  // A rows then W rows
        goff[i] = row < GM ? (uint32_t)min(bm0 + row, M - 1) * (uint32_t)((AMODE ? p.Cin : K) * 2) + schunk * 16
                           : (uint32_t)min(bn0 + row - GM, N - 1) * (uint32_t)(K * 2) + schunk * 16;
        loff[i] = row * LROW + schunk * 16;
        if (AMODE == 1 && i < NA) token_to_voxel(min(bm0 + row, M - 1), p.R, vb[i < NA ? i : 0], vd[i < NA ? i : 0], vh[i < NA ? i : 0], vw[i < NA ? i : 0]);
    }
    const int ktiles = K / GK;               // K is a multiple of 32 (checked by the callers)
    const int kpt = AMODE ? p.Cin / GK : 1;  // K tiles per filter tap
    // (staging registers are named scalars, not an array: an indexed private array ends up in scratch)
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 st0 = z4, st1 = z4, st2 = z4, st3 = z4, st4 = z4, st5 = z4, st6 = z4, st7 = z4;
    uint32_t noff[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) noff[i] = 0;
    // pass i of the staging map: from A (dense row or the current tap's neighbour row) or from W
#define L2D_GSRC(i) ((i) < NA ? (AMODE == 1 ? Ab + (noff[(i) < NA ? (i) : 0] + kcb) : Ab + (goff[i] + kb)) : Wb + (goff[i] + kb))
    auto gload = [&](int kt) {
        const uint32_t kb = (uint32_t)kt * (GK * 2);
        uint32_t kcb = 0;
        if (AMODE == 1) {
            const int tap = kt / kpt, kc = kt - tap * kpt;
            if (kc == 0) {  // a new filter tap: resolve this thread's neighbour rows once for its kpt K tiles
                const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
#pragma unroll
                for (int i = 0; i < NA; i++) {
                    const int nd = vd[i] + dz, nh = vh[i] + dy, nw = vw[i] + dx;
                    const bool in = (unsigned)nd < (unsigned)p.R && (unsigned)nh < (unsigned)p.R && (unsigned)nw < (unsigned)p.R;
                    // outside the volume: a row of zeros the caller keeps next to the activations (no branch)
                    noff[i] = (in ? (uint32_t)voxel_to_token(vb[i], nd, nh, nw, p.R) * (uint32_t)(p.Cin * 2) : p.zero_off) + schunk * 16;
                }
            }
            kcb = (uint32_t)kc * (GK * 2);
        }
        st0 = *(const uint4 *)L2D_GSRC(0); st1 = *(const uint4 *)L2D_GSRC(1);
        st2 = *(const uint4 *)L2D_GSRC(2); st3 = *(const uint4 *)L2D_GSRC(3);
        if (NS > 4) { st4 = *(const uint4 *)L2D_GSRC(NS > 4 ? 4 : 0); st5 = *(const uint4 *)L2D_GSRC(NS > 5 ? 5 : 0); }
        if (NS > 6) { st6 = *(const uint4 *)L2D_GSRC(NS > 6 ? 6 : 0); st7 = *(const uint4 *)L2D_GSRC(NS > 7 ? 7 : 0); }
    };
#undef L2D_GSRC
    auto lstore = [&](int buf) {
        unsigned char *l = &lds[buf][0];
        *(uint4 *)(l + loff[0]) = st0; *(uint4 *)(l + loff[1]) = st1;
        *(uint4 *)(l + loff[2]) = st2; *(uint4 *)(l + loff[3]) = st3;
        if (NS > 4) { *(uint4 *)(l + loff[NS > 4 ? 4 : 0]) = st4; *(uint4 *)(l + loff[NS > 5 ? 5 : 0]) = st5; }
        if (NS > 6) { *(uint4 *)(l + loff[NS > 6 ? 6 : 0]) = st6; *(uint4 *)(l + loff[NS > 7 ? 7 : 0]) = st7; }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < ktiles) gload(kt + 1);  // in flight while this tile is multiplied
        const unsigned char *la = &lds[buf][(wm + r) * LROW + kh * 16];
        const unsigned char *lb = &lds[buf][(GM + wn + r) * LROW + kh * 16];
        // fragments of K step s+1 are fetched while the MFMAs of step s run
        bf16x8 a[2][MI], b[2][NJ];
#pragma unroll
        for (int i = 0; i < MI; i++) a[0][i] = *(const bf16x8 *)(la + i * 32 * LROW);
#pragma unroll
        for (int j = 0; j < NJ; j++) b[0][j] = *(const bf16x8 *)(lb + j * 32 * LROW);
#pragma unroll
        for (int s = 0; s < GK / 16; s++) {
            if (s + 1 < GK / 16) {
#pragma unroll
                for (int i = 0; i < MI; i++) a[(s + 1) & 1][i] = *(const bf16x8 *)(la + i * 32 * LROW + (s + 1) * 32);
#pragma unroll
                for (int j = 0; j < NJ; j++) b[(s + 1) & 1][j] = *(const bf16x8 *)(lb + j * 32 * LROW + (s + 1) * 32);
            }
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1][i], b[s & 1][j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ktiles) lstore(buf ^ 1);  // the other buffer was last read one iteration ago
        __syncthreads();
    }
    // Epilogue through LDS: the accumulator layout gives every lane one element of 16 different
    // rows (4-byte stores, issue bound); bouncing a 32x64 block per wave through LDS turns that into
    // 16-byte row-contiguous loads of the residual and 16-byte stores.
    float *ep = (float *)&lds[0][0] + wave * (32 * 68);  // 32 rows x (64 + 4 pad) floats per wave
    unsigned long long *rowbase = (unsigned long long *)((float *)&lds[0][0] + 4 * (32 * 68)) + wave * 32;  // EPI 5
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);          // EPI 7: this thread's share of the tile's column sums
#pragma unroll
    for (int ij = 0; ij < MI * (NJ / 2); ij++) {
        const int i = ij / (NJ / 2), jh = ij % (NJ / 2);  // 32 rows x 64 columns of the wave tile per trip
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ep[((e & 3) + 8 * (e >> 2) + 4 * kh) * 68 + j * 32 + r] = acc[i][2 * jh + j][e];
        if (EPI == 5 && lane < 32) {  // output offset of the voxel (2d, 2h, 2w) of each of the wave's 32 rows
            int b, d, h, w;
            token_to_voxel(min(bm0 + wm + i * 32 + lane, M - 1), p.R, b, d, h, w);
            const size_t R2 = 2 * (size_t)p.R;
            rowbase[lane] = (((size_t)b * R2 + 2 * d) * R2 + 2 * h) * R2 + 2 * w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = q * 64 + lane, lr = idx >> 4, c4 = (idx & 15) * 4;
            const int row = bm0 + wm + i * 32 + lr, col = bn0 + wn + jh * 64 + c4;
            if (row < M && col < N) {
                if (EPI == 7 && NJ == 2) {
                    float4 st;
                    gemm_epilogue<EPI>(p, *(const float4 *)(ep + lr * 68 + c4), row, col, N, 0ull, &st);
                    cs.x += st.x; cs.y += st.y; cs.z += st.z; cs.w += st.w;
                } else {
                    gemm_epilogue<EPI>(p, *(const float4 *)(ep + lr * 68 + c4), row, col, N, EPI == 5 ? rowbase[lr] : 0ull);
                }
            }
        }
    }
    // EPI 7: the bias gradient of the layer in front is the column sum of what was just stored.  A thread holds four columns of
    // 16 of the wave's 64 rows; the four lane groups add up with two shuffles, the two waves of a column half through LDS, and the
    // workgroup leaves one row of per-tile sums (a separate pass over C would read it all again: 134 MB per block of LaRa's encoder)
    if (EPI == 7 && NJ == 2 && p.colsum) {
        cs.x += __shfl_xor(cs.x, 16); cs.y += __shfl_xor(cs.y, 16); cs.z += __shfl_xor(cs.z, 16); cs.w += __shfl_xor(cs.w, 16);
        cs.x += __shfl_xor(cs.x, 32); cs.y += __shfl_xor(cs.y, 32); cs.z += __shfl_xor(cs.z, 32); cs.w += __shfl_xor(cs.w, 32);
        __syncthreads();
        float *cb = (float *)&lds[0][0];      // [2 row halves][GN]
        if (lane < 16) *(float4 *)(cb + (wave >> 1) * GN + wn + lane * 4) = cs;
        __syncthreads();
        if (tid < GN && bn0 + tid < N) p.colsum[(size_t)rt_ * N + bn0 + tid] = cb[tid] + cb[GN + tid];
    }
}

// ---- the same GEMM family on a 256 x 256 tile with an LDS-DMA ring ------------------------------------
// Same operands, AMODE and EPI as gemm_bf16_nt_kernel, restructured around what limited that kernel
// (DESIGN.md section 3.4): register staging put as many LDS cycles into ds_write_b128 (13 cycles per
// wave instruction) as into fragment reads, and the 3x3x3 convolution re-read its gathered operand
// through L2 once per 128 output columns.  Here
//   * a workgroup of 8 waves (2 x 4, wave tile 128 x 64) owns 256 x 256 outputs (for the convolution:
//     all of N, so the gathered operand is fetched once per tap);
//   * K tiles (32 channels of one tap: A 256 rows x 64 B, W 256 rows x 64 B = 32 KB) travel global ->
//     LDS with global_load_lds_dwordx4: no staging registers, no ds_write; every wave issues 4 of the 32
//     one-KB pieces of a tile;
//   * a ring of four tile buffers keeps three tiles in flight behind the one being multiplied: the
//     barrier that publishes tile kt+1 sits between the two K steps of tile kt (two-phase schedule,
//     see the main loop), each wave waits for its own four pieces with a COUNTED vmcnt (8 younger
//     pieces stay in flight), and tile kt's buffer is refilled with tile kt+4 right after it;
//   * LDS rows are 64 B, unpadded (the DMA writes lane-linear); the 16-byte chunk index is XORed with
//     (row >> 2) & 3 on the SOURCE address and on the fragment reads, which spreads the rows a
//     ds_read_b128 lane group touches over all four chunk slots.
constexpr int RT = 256;             // tile rows (M) and columns (N)
constexpr int RING = 4;             // tile buffers
constexpr int RTILE = 2 * RT * 64;  // bytes per K tile: A panel + W panel

// Epilogue of the 256 x 256 ring kernels, through LDS (32 x 64 per wave per trip): the same fused variants as
// gemm_bf16_nt_kernel.  Wave tile: rows wr * 128 .., columns wc * 64 ..; the caller has synchronised the workgroup
// after its last fragment read (the ring is reused as the bounce buffer).
template <int EPI>
__device__ __forceinline__ void ring_epilogue(const GemmP &p, f32x16 (&acc)[4][2], unsigned char *ring, const int wave,
                                              const int lane, const int bm0, const int bn0, const int wr, const int wc) {
    const int M = p.M, N = p.N;
    const int r = lane & 31, kh = lane >> 5;
    float *ep = (float *)ring + wave * (32 * 68);
    unsigned long long *rowbase = (unsigned long long *)((float *)ring + 8 * (32 * 68)) + wave * 32;  // EPI 5
#pragma unroll
    for (int i = 0; i < 4; i++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ep[((e & 3) + 8 * (e >> 2) + 4 * kh) * 68 + j * 32 + r] = acc[i][j][e];
        if (EPI == 5 && lane < 32) {
            int b, d, h, w;
            token_to_voxel(min(bm0 + wr * 128 + i * 32 + lane, M - 1), p.R, b, d, h, w);
            const size_t R2 = 2 * (size_t)p.R;
            rowbase[lane] = (((size_t)b * R2 + 2 * d) * R2 + 2 * h) * R2 + 2 * w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = q * 64 + lane, lr = idx >> 4, c4 = (idx & 15) * 4;
            const int row = bm0 + wr * 128 + i * 32 + lr, col = bn0 + wc * 64 + c4;
            if (row < M && col < N)
                gemm_epilogue<EPI>(p, *(const float4 *)(ep + lr * 68 + c4), row, col, N, EPI == 5 ? rowbase[lr] : 0ull);
        }
    }
}

// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), the same value in each of them
__device__ __forceinline__ float row16_total(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, false));   // row_half_mirror
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, false));   // row_mirror
    return x;
}

// EPI 9 -- the convolution's input gradient with the LayerNorm backward behind it in the epilogue (network.py:94-100 backwards:
// x_out = pn + cnn(pn), pn = norm3(x2)): dy = acc + resid is the gradient of pn; the tile holds whole rows (N = 256), so
//   dx = rstd (a - mean(a) - xhat mean(a xhat)),  a = dy gamma,  xhat = (x2 - mean) rstd   (mean, rstd: the forward's stats)
// is formed here -- fp32 to C (in place over resid), bf16 to C2 -- with the per-tile column sums of dy xhat, dy and dx (dgamma, dbeta,
// the bias gradient of the layer in front) to p.colsum.  The stand-alone pass read and wrote the 134 MB gradient stream once more
// (76 us per block).  Through LDS like ring_epilogue (32 x 64 per wave per trip); a row's 256 columns sit in the four waves of a
// wave row: its two sums are DPP row totals (16 lanes = 64 columns) exchanged through 2 KB of LDS, 16 rows at a time.
__device__ __forceinline__ void ring_epilogue_lnbwd(const GemmP &p, f32x16 (&acc)[4][2], unsigned char *ring, const int wave,
                                                    const int lane, const int bm0, const int wr, const int wc) {
    const int M = p.M, tid = wave * 64 + lane;
    const int r = lane & 31, kh = lane >> 5;
    float *ep = (float *)ring + wave * (32 * 68);
    float2 *rowsum = (float2 *)((float *)ring + 8 * (32 * 68) + 512);   // [2 wave rows][32 rows][4 wave columns]
    const int lg = lane >> 4, c4 = (lane & 15) * 4, col = wc * 64 + c4;
    const float4 ga = *(const float4 *)(p.gamma + col);
    float cg[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f}, co[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ep[((e & 3) + 8 * (e >> 2) + 4 * kh) * 68 + j * 32 + r] = i == 0 ? acc[0][j][e] : i == 1 ? acc[1][j][e] : i == 2 ? acc[2][j][e] : acc[3][j][e];
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            __syncthreads();   // the bounce tile is written (half 0) / the row sums of the half before have been read
            float4 dy[4], xh[4];
            float rstd[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lr = 16 * half + 4 * q + lg, row = bm0 + wr * 128 + i * 32 + lr;
                const bool in = row < M;
                const size_t o = (size_t)(in ? row : 0) * 256 + col;
                const float4 v = *(const float4 *)(ep + lr * 68 + c4), rs = *(const float4 *)(p.resid + o), xx = *(const float4 *)(p.lnx + o);
                const float2 st = p.stats[in ? row : 0];
                const float m = in ? 1.f : 0.f;
                dy[q] = make_float4(m * (v.x + rs.x), m * (v.y + rs.y), m * (v.z + rs.z), m * (v.w + rs.w));
                xh[q] = make_float4((xx.x - st.x) * st.y, (xx.y - st.x) * st.y, (xx.z - st.x) * st.y, (xx.w - st.x) * st.y);
                rstd[q] = st.y;
                const float a0 = dy[q].x * ga.x, a1 = dy[q].y * ga.y, a2 = dy[q].z * ga.z, a3 = dy[q].w * ga.w;
                const float sa = row16_total((a0 + a1) + (a2 + a3));
                const float sh = row16_total((a0 * xh[q].x + a1 * xh[q].y) + (a2 * xh[q].z + a3 * xh[q].w));
                if ((lane & 15) == 0) rowsum[(wr * 32 + lr) * 4 + wc] = make_float2(sa, sh);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lr = 16 * half + 4 * q + lg, row = bm0 + wr * 128 + i * 32 + lr;
                const float4 s01 = *(const float4 *)(rowsum + (wr * 32 + lr) * 4), s23 = *(const float4 *)(rowsum + (wr * 32 + lr) * 4 + 2);
                const float sa = ((s01.x + s01.z) + (s23.x + s23.z)) * (1.0f / 256.0f), sh = ((s01.y + s01.w) + (s23.y + s23.w)) * (1.0f / 256.0f);
                const float a0 = dy[q].x * ga.x, a1 = dy[q].y * ga.y, a2 = dy[q].z * ga.z, a3 = dy[q].w * ga.w;
                const float4 dx = make_float4(rstd[q] * (a0 - sa - xh[q].x * sh), rstd[q] * (a1 - sa - xh[q].y * sh),
                                              rstd[q] * (a2 - sa - xh[q].z * sh), rstd[q] * (a3 - sa - xh[q].w * sh));
                if (row < M) {
                    const size_t o = (size_t)row * 256 + col;
                    *(float4 *)((float *)p.C + o) = dx;
                    ushort4 hb;
                    hb.x = f2bf(dx.x); hb.y = f2bf(dx.y); hb.z = f2bf(dx.z); hb.w = f2bf(dx.w);
                    *(ushort4 *)(p.C2 + o) = hb;
                    cg[0] += dy[q].x * xh[q].x; cg[1] += dy[q].y * xh[q].y; cg[2] += dy[q].z * xh[q].z; cg[3] += dy[q].w * xh[q].w;
                    cb[0] += dy[q].x; cb[1] += dy[q].y; cb[2] += dy[q].z; cb[3] += dy[q].w;
                    co[0] += dx.x; co[1] += dx.y; co[2] += dx.z; co[3] += dx.w;
                }
            }
        }
    }
    // column sums: a lane holds four columns of the rows = lane >> 4 (mod 4) of its wave row; the four lane groups add up with two
    // shuffles, the two wave rows through LDS
#pragma unroll
    for (int c = 0; c < 4; c++) {
        cg[c] += __shfl_xor(cg[c], 16); cg[c] += __shfl_xor(cg[c], 32);
        cb[c] += __shfl_xor(cb[c], 16); cb[c] += __shfl_xor(cb[c], 32);
        co[c] += __shfl_xor(co[c], 16); co[c] += __shfl_xor(co[c], 32);
    }
    __syncthreads();
    float *cs = (float *)ring;   // [2 wave rows][3][256]
    if (lane < 16) {
        *(float4 *)(cs + (wr * 3 + 0) * 256 + col) = make_float4(cg[0], cg[1], cg[2], cg[3]);
        *(float4 *)(cs + (wr * 3 + 1) * 256 + col) = make_float4(cb[0], cb[1], cb[2], cb[3]);
        *(float4 *)(cs + (wr * 3 + 2) * 256 + col) = make_float4(co[0], co[1], co[2], co[3]);
    }
    __syncthreads();
    for (int k = tid; k < 768; k += 512) p.colsum[(size_t)(bm0 / RT) * 768 + k] = cs[k] + cs[768 + k];
}

template <int AMODE, int EPI>
__global__ void __launch_bounds__(512)
gemm_ring_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // RING * RTILE bytes
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int M = p.M, N = p.N, K = p.K, Cin = AMODE ? p.Cin : 0;
    int rt_, ct_;
    xcd_tile(rt_, ct_);
    const int bm0 = rt_ * RT, bn0 = ct_ * RT;
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
        woff[q] = (uint32_t)min(bn0 + srow[q], N - 1) * (uint32_t)(K * 2) + chunk * 16;
        if (AMODE == 1) {
            token_to_voxel(min(bm0 + srow[q], M - 1), p.R, vb[q], vd[q], vh[q], vw[q]);
            noff[q] = chunk * 16;
        } else {
            vb[q] = vd[q] = vh[q] = vw[q] = 0;
            noff[q] = (uint32_t)min(bm0 + srow[q], M - 1) * (uint32_t)(K * 2) + chunk * 16;  // dense A row
        }
    }
    const int ktiles = K / 32, kpt = AMODE ? Cin / 32 : 1;
    int cur_tap = -1;
    auto issue = [&](const int kt) {  // this wave's four pieces of K tile kt
        const int tap = AMODE ? kt / kpt : 0, kc = AMODE ? kt - tap * kpt : kt, ksrc = kt;
        if (AMODE == 1 && tap != cur_tap) {
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
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Wb + (woff[q] + (uint32_t)ksrc * 64)),
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

    // Two-phase schedule.  A K tile is two K steps of eight MFMAs; the fragments of a step are fetched from LDS while
    // the eight MFMAs of the step before run (256 matrix-core cycles of cover), and the barrier that publishes tile
    // kt+1 sits BETWEEN the two steps of tile kt: by then every fragment of tile kt is in registers (lgkmcnt(0) is free,
    // its reads were issued a step ago), so tile kt's buffer is refilled with tile kt+4 right after the barrier --
    // three tiles stay in flight -- and the matrix cores always have a step's worth of issued work behind them.
    // (Round 5 measured the CDNA guides' staggered form of this loop on the same box: TWO barriers per K tile -- "tile kt+1 landed" /
    // "nobody reads tile kt any more" -- with wave row 1 one barrier behind wave row 0 (every SIMD holds one wave of each row, so one
    // is inside a step's MFMAs while the other sits in its waits) and s_setprio(1) around the MFMA runs.  Parity-green on the first
    // build -- and slower: convolution 476-485 us against 435-437, its input gradient 452 against 426.  The lock-step is not what
    // idles the matrix pipe: with two waves per SIMD each already waits for the other's MFMAs ~40 % of its cycles (SQ_WAIT_INST_ANY /
    // SQ_WAVE_CYCLES), the pipe is busy 58 % of the SIMD's cycles at the clock the part sustains under this load, and a second
    // barrier per tile costs more than the stagger returns.  DESIGN.md section 3.4.)
    bf16x8 fa[2][4], fb[2][2];
    auto fetch = [&](const int set, const int kt, const int step) {
        const unsigned char *buf = ring + (kt % RING) * RTILE;
#pragma unroll
        for (int i = 0; i < 4; i++) fa[set][i] = *(const bf16x8 *)(buf + aoff[i][step]);
#pragma unroll
        for (int j = 0; j < 2; j++) fb[set][j] = *(const bf16x8 *)(buf + boff[j][step]);
    };
    auto mac = [&](const int set, const int i0, const int i1) {
#pragma unroll
        for (int i = i0; i < i1; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    auto wait_tile = [&](const int younger) {   // my pieces of a tile have landed once only `younger` newer tiles are in flight
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    issue(0);
    if (1 < ktiles) issue(1);
    if (2 < ktiles) issue(2);
    if (3 < ktiles) issue(3);
    wait_tile(min(ktiles - 1, 3));
    __builtin_amdgcn_s_barrier();
    fetch(0, 0, 0);
    // (The fetches are issued after the first two MFMAs of a step, not before them: the compiler's waitcnt pass puts
    // lgkmcnt(0) in front of the step's first MFMA at this loop's joins, which would otherwise wait for the fetch
    // just issued instead of the one a step old.)
    for (int kt = 0; kt < ktiles; kt++) {
        mac(0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, kt, 1);
        __builtin_amdgcn_sched_barrier(0);
        mac(0, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all of tile kt is in registers
        if (kt + 1 < ktiles) {
            wait_tile(min(ktiles - 2 - kt, 2));
            __builtin_amdgcn_s_barrier();   // tile kt+1 visible to everybody; nobody reads tile kt's buffer any more
        }
        __builtin_amdgcn_sched_barrier(0);
        mac(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ktiles) {
            if (kt + 4 < ktiles) issue(kt + 4);
            fetch(0, kt + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mac(1, 1, 4);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    ring_epilogue<EPI == 9 ? 1 : EPI>(p, acc, ring, wave, lane, bm0, bn0, wr, wc);   // (EPI 9 exists on gemm_ring2_kernel only)
}

// ---- the ring kernel, finely interleaved --------------------------------------------------------------
// Same tiles, ring, swizzle and epilogues as gemm_ring_kernel; what changes is WHEN a wave issues its non-matrix
// instructions.  Two waves share a SIMD's matrix pipe and alternate on it, so each has a ~32-cycle hole behind every MFMA
// it issues; a K tile's 12 fragment reads and 4 LDS-DMA pieces (each tens of cycles of issue) fit in those holes only when
// they are dealt one per MFMA -- gemm_ring_kernel issues them in two clumps (6 reads; 4 pieces + 6 reads) behind which both
// waves of a SIMD, in lock-step after the barrier, leave the pipe idle.  Here
//   * the K loop is unrolled over the ring (4 tiles): ring slots, LDS offsets and the DMA destinations are literals -- no
//     address arithmetic on the vector ALU, M0 is a scalar move (the wave index is read into an SGPR once), a piece's global
//     address is SGPR base + a lane offset that moves once per four tiles + the instruction's offset field;
//   * tile kt+3 is issued DURING tile kt, a piece at a time between MFMAs, into the slot tile kt-1 left at the last barrier
//     (two tiles and a half in flight instead of three); each wave waits for its pieces of tile kt+1 with a counted vmcnt;
//   * fragment reads of the next K step are dealt one per MFMA in the order the next step consumes them, under the
//     compiler's counted lgkmcnt waits.
// Round 5, same box, LaRa's convolution (M = 131072, N = 256, K = 6912): gemm_ring_kernel 432-445 us -> 396-409 with the
// interleave -> 401-402 with the two wave rows issuing in different K steps (its input gradient 425-435 -> 399-405).  Timing-only
// builds of this loop (results invalid) place the rest: MFMAs alone 278 us (1.67 PF: the clock the part sustains), + fragment
// reads 312, + pieces 362 (no barrier), + the barrier 408 -- the barrier costs nothing without the pieces (306): what it costs is
// the waves' different luck at issuing them; frozen source addresses (everything from L1 / L2) change nothing (405).
// Where inside a K step the pieces sit (behind MFMAs 2-5 or 2, 4, 6, 7) and s_setprio(1) over the K step that issues none
// measured within noise of each other (394-400 us, two passes each).
// The epilogue (half a megabyte per tile: residual in, result out) runs with the CU's matrix pipe idle -- one workgroup per CU, and a
// round's workgroups reach it together: 36 of the convolution's 392 us (measured by skipping it).  Built to hide it and measured on
// the same box: 4-wave workgroups on 128 x 256 tiles with a three-stage ring (72 KB), two per CU, each running under the other's
// epilogue and waits -- parity-green, 485 us against 400: the W panel then travels once per 128 rows, 6 pieces per wave and K tile
// instead of 4, and the pieces are what this loop pays for.  The block's MLP products (K = 256 / 512, HBM-bound at 94-104 us) on this
// kernel: 99-102 us, no gain over gemm_bf16_nt_kernel's four workgroups per CU.
// Needs K % 128 == 0 (and Cin % 128 == 0 for the convolution); launch_gemm_ring falls back to gemm_ring_kernel otherwise.
template <int AMODE, int EPI>
__global__ void __launch_bounds__(512)
gemm_ring2_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // RING * RTILE bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K, Cin = AMODE ? p.Cin : 0;
    int rt_, ct_;
    xcd_tile(rt_, ct_);
    const int bm0 = rt_ * RT, bn0 = ct_ * RT;
    const int wr = wave >> 2, wc = wave & 3;  // wave tile: rows wr*128.., columns wc*64..
    const int r = lane & 31, kh = lane >> 5;
    const char *Ab = (const char *)p.A, *Wb = (const char *)p.W;

    // staging (as gemm_ring_kernel): wave w moves pieces {2w, 2w+1} of the A panel and of the W panel
    int srow[2];
    uint32_t woff[2], noff[2];
    int vb[2], vd[2], vh[2], vw[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        srow[q] = 32 * wave + 16 * q + (lane >> 2);
        const int chunk = (lane & 3) ^ ((srow[q] >> 2) & 3);
        woff[q] = (uint32_t)min(bn0 + srow[q], N - 1) * (uint32_t)(K * 2) + chunk * 16;
        if (AMODE == 1) {
            token_to_voxel(min(bm0 + srow[q], M - 1), p.R, vb[q], vd[q], vh[q], vw[q]);
            noff[q] = chunk * 16;
        } else {
            vb[q] = vd[q] = vh[q] = vw[q] = 0;
            noff[q] = (uint32_t)min(bm0 + srow[q], M - 1) * (uint32_t)(K * 2) + chunk * 16;  // dense A row
        }
    }
    auto set_tap = [&](const int tap) {   // the convolution's gathered rows of one tap
        const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int nd = vd[q] + dz, nh = vh[q] + dy, nw = vw[q] + dx;
            const bool in = (unsigned)nd < (unsigned)p.R && (unsigned)nh < (unsigned)p.R && (unsigned)nw < (unsigned)p.R;
            const int chunk = (lane & 3) ^ ((srow[q] >> 2) & 3);
            noff[q] = (in ? (uint32_t)voxel_to_token(vb[q], nd, nh, nw, p.R) * (uint32_t)(Cin * 2) : p.zero_off) + chunk * 16;
        }
    };
    // issue side: the group (four K tiles = 128 channels) whose tiles are being fetched.  A piece's address is the operand's
    // base (SGPR pair) + a 32-bit lane offset that moves once per group + the tile's 64 t bytes in the instruction's offset
    // field; that field is added to the LDS address too, so M0 (a literal + the wave's base) is set 64 t lower.
    const int gpt = AMODE ? Cin / 128 : 1;
    int g_in_tap = 0, tap_i = 0;
    if (AMODE == 1) set_tap(0);
    auto next_group = [&]() {
        woff[0] += 256; woff[1] += 256;
        if (AMODE == 1 && ++g_in_tap == gpt) {
            g_in_tap = 0;
            set_tap(++tap_i);
        } else {
            noff[0] += 256; noff[1] += 256;
        }
    };
    // (The piece is inline assembly on purpose: behind the compiler's own global_load_lds its waitcnt pass stops counting LDS
    // returns and drains lgkmcnt to 0 in front of every MFMA that consumes a fragment -- i.e. each K step would start by waiting for
    // the read issued two MFMAs earlier.  Invisible to that pass, the reads get the counted waits their issue order allows; the
    // pieces themselves are only ever waited for by the explicit vmcnt below.)
    const uint32_t mine = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)ring + wave * 2048;
#define R2_DMA(base, voff, lds, t)                                                                                   \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"                            \
                 :: "s"((uint32_t)((lds) - (t) * 64)), "v"(voff), "s"(base), "n"((t) * 64) : "memory")
    // piece 2 wave + q of tile t of the issue group, into ring slot `slot`
#define R2_DMA_A(slot, t, q) R2_DMA(Ab, noff[q], mine + (slot) * RTILE + (q) * 1024, t)
#define R2_DMA_W(slot, t, q) R2_DMA(Wb, woff[q], mine + (slot) * RTILE + RT * 64 + (q) * 1024, t)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // fragment addresses inside a tile buffer (row * 64 + swizzled chunk * 16), for K steps 0 and 1
    // (rows 32 i further are 2048 i bytes further with the same swizzle; ring slots 2 and 3 are beyond a ds_read's 16-bit offset
    // field and get base registers of their own, pinned so that they are not re-derived with a vector add per read)
    int aoff[2][2], boff[2][2];   // [K step][ring half]
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int ra = wr * 128 + r, rb = wc * 64 + r;
        aoff[s][0] = ra * 64 + (((2 * s + kh) ^ ((ra >> 2) & 3)) << 4);
        boff[s][0] = RT * 64 + rb * 64 + (((2 * s + kh) ^ ((rb >> 2) & 3)) << 4);
        aoff[s][1] = aoff[s][0] + 2 * RTILE;
        boff[s][1] = boff[s][0] + 2 * RTILE;
        asm volatile("" : "+v"(aoff[s][1]), "+v"(boff[s][1]));
    }
    bf16x8 fa[2][4], fb[2][2];
#define R2_SB __builtin_amdgcn_sched_barrier(0)
#define R2_M(set, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0)
#define R2_RA(set, i, slot, step) fa[set][i] = *(const bf16x8 *)(ring + ((slot) & 1) * RTILE + aoff[step][(slot) >> 1] + (i) * 2048)
#define R2_RB(set, j, slot, step) fb[set][j] = *(const bf16x8 *)(ring + ((slot) & 1) * RTILE + boff[step][(slot) >> 1] + (j) * 2048)
    // one K step: the eight MFMAs of fragment set USE, between them the six fragment reads of the step after it (set USE ^ 1,
    // from ring slot `slot`, K step `step`) and pieces of tile `it` of the issue group (into slot `it`): MODE 1 = its A and W
    // pieces number q, MODE 2 = all four, MODE 0 = none
#define R2_HALF(USE, READ, slot, step, MODE, it, q)                      \
    R2_M(USE, 0, 0); R2_SB;                                              \
    if (READ) R2_RA(USE ^ 1, 0, slot, step);                             \
    R2_SB; R2_M(USE, 0, 1); R2_SB;                                       \
    if (READ) R2_RB(USE ^ 1, 0, slot, step);                             \
    if ((MODE) == 2) R2_DMA_A(it, it, 0);                                \
    R2_SB; R2_M(USE, 1, 0); R2_SB;                                       \
    if (READ) R2_RB(USE ^ 1, 1, slot, step);                             \
    if ((MODE) == 1) R2_DMA_A(it, it, q);                                \
    if ((MODE) == 2) R2_DMA_W(it, it, 0);                                \
    R2_SB; R2_M(USE, 1, 1); R2_SB;                                       \
    if (READ) R2_RA(USE ^ 1, 1, slot, step);                             \
    if ((MODE) == 2) R2_DMA_A(it, it, 1);                                \
    R2_SB; R2_M(USE, 2, 0); R2_SB;                                       \
    if (READ) R2_RA(USE ^ 1, 2, slot, step);                             \
    if ((MODE) == 1) R2_DMA_W(it, it, q);                                \
    if ((MODE) == 2) R2_DMA_W(it, it, 1);                                \
    R2_SB; R2_M(USE, 2, 1); R2_SB;                                       \
    if (READ) R2_RA(USE ^ 1, 3, slot, step);                             \
    R2_SB; R2_M(USE, 3, 0); R2_M(USE, 3, 1); R2_SB;
    // tile t of a group sits in ring slot t; during it the tile three further on (slot and tile-in-group (t + 3) & 3) is issued:
    // M1 / M2 = what the first / second K step issues.  VM = this wave's younger pieces allowed in flight when its pieces of the
    // NEXT tile must have landed (-1: there is no next tile)
#define R2_TILE(t, M1, M2, VM)                                                                  \
    R2_HALF(0, true, t, 1, M1, ((t) + 3) & 3, 0)                                                \
    if ((VM) >= 0) {                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
        if ((VM) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                         \
        else if ((VM) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                    \
        else if ((VM) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_s_barrier();                                                           \
    }                                                                                           \
    R2_SB;                                                                                      \
    R2_HALF(1, (VM) >= 0, ((t) + 1) & 3, 0, M2, ((t) + 3) & 3, 1)
    // the K loop; VMS = VM of the steady state (= of the last group's first tile)
#define R2_LOOP(M1, M2, VMS)                                      \
    for (int g = 0, groups = K / 128; g + 1 < groups; g++) {      \
        R2_TILE(0, M1, M2, VMS)                                   \
        next_group();                                             \
        R2_SB;                                                    \
        R2_TILE(1, M1, M2, VMS)                                   \
        R2_TILE(2, M1, M2, VMS)                                   \
        R2_TILE(3, M1, M2, VMS)                                   \
    }                                                             \
    R2_TILE(0, M1, M2, VMS)                                       \
    R2_TILE(1, 0, 0, 4)                                           \
    R2_TILE(2, 0, 0, 0)                                           \
    R2_TILE(3, 0, 0, -1)

    // prologue: tiles 0, 1, 2
    R2_DMA_A(0, 0, 0); R2_DMA_W(0, 0, 0); R2_DMA_A(0, 0, 1); R2_DMA_W(0, 0, 1);
    R2_DMA_A(1, 1, 0); R2_DMA_W(1, 1, 0); R2_DMA_A(1, 1, 1); R2_DMA_W(1, 1, 1);
    R2_DMA_A(2, 2, 0); R2_DMA_W(2, 2, 0); R2_DMA_A(2, 2, 1); R2_DMA_W(2, 2, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    R2_RA(0, 0, 0, 0); R2_RB(0, 0, 0, 0); R2_RB(0, 1, 0, 0); R2_RA(0, 1, 0, 0); R2_RA(0, 2, 0, 0); R2_RA(0, 3, 0, 0);
    R2_SB;
    // a SIMD holds one wave of each wave row (waves w and w + 4): row 0 issues a tile's four pieces during the first K step, row 1
    // during the second, so that the two waves sharing a matrix pipe are never both held up behind a piece at the same time
    // (measured against every wave issuing two pieces per K step, R2_LOOP(1, 1, 6): convolution 407-409 -> 401-402 us)
    if (wr == 0) {
        R2_LOOP(2, 0, 8)
    } else {
        R2_LOOP(0, 2, 4)
    }
#undef R2_LOOP
#undef R2_TILE
#undef R2_DMA_A
#undef R2_DMA_W
#undef R2_DMA
#undef R2_HALF
#undef R2_RA
#undef R2_RB
#undef R2_M
#undef R2_SB
    __syncthreads();

    // ring_epilogue's accumulator order is acc[i][j] too
    if (EPI == 9) ring_epilogue_lnbwd(p, acc, ring, wave, lane, bm0, wr, wc);
    else ring_epilogue<EPI == 9 ? 1 : EPI>(p, acc, ring, wave, lane, bm0, bn0, wr, wc);
}

// host-side launch of the ring kernel (dynamic LDS above the 64 KB default needs the attribute once)
template <int AMODE, int EPI>
static inline hipError_t launch_gemm_ring(const GemmP &p, hipStream_t s) {
    static const hipError_t attr = hipFuncSetAttribute((const void *)gemm_ring_kernel<AMODE, EPI>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, RING * RTILE);
    if (attr != hipSuccess) return attr;
    // EPI 9's row sums need a tile that holds WHOLE rows of the LayerNorm, its saved statistics and its outputs
    if (EPI == 9 && (p.N != RT || !p.lnx || !p.stats || !p.colsum || !p.C2 || !p.gamma)) return hipErrorInvalidValue;
    if (p.K % 128 == 0 && (!AMODE || p.Cin % 128 == 0)) {
        static const hipError_t attr2 = hipFuncSetAttribute((const void *)gemm_ring2_kernel<AMODE, EPI>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, RING * RTILE);
        if (attr2 != hipSuccess) return attr2;
        hipLaunchKernelGGL((gemm_ring2_kernel<AMODE, EPI>), dim3((p.M + RT - 1) / RT, (p.N + RT - 1) / RT), dim3(512),
                           RING * RTILE, s, p);
        return hipSuccess;
    }
    if (EPI == 9) return hipErrorInvalidValue;   // (the fused LayerNorm backward needs gemm_ring2_kernel's shapes: callers check ring2_shape)
    hipLaunchKernelGGL((gemm_ring_kernel<AMODE, EPI>), dim3((p.M + RT - 1) / RT, (p.N + RT - 1) / RT), dim3(512),
                       RING * RTILE, s, p);
    return hipSuccess;
}
static inline bool ring2_shape(const GemmP &p, const int amode) { return p.K % 128 == 0 && (!amode || p.Cin % 128 == 0); }

}  // namespace
