// coarsedec.hip -- `Decoder.forward_coarse` (lightning/network.py:259-278) as one kernel per direction; the interface
// and the arithmetic contract are in include/lara_coarsedec.h.
//
// Mapping.  v_mfma_f32_16x16x16_bf16: A[row = lane & 15][k = 4 (lane >> 4) + e], B[k = 4 (lane >> 4) + e][col = lane & 15],
// D[row = 4 (lane >> 4) + r][col = lane & 15].  The layers are computed transposed, H^T = W X^T: A = a 16 x 16 block of the
// weight matrix (rows = output features), B = 16 input features x 16 voxel rows.  A lane then holds, for ITS voxel row
// (lane & 15), output features 16 i + 4 (lane >> 4) + r, r = 0..3, of row block i -- which is precisely the B fragment
// of input-feature block i of the next layer: bias, ReLU and the bf16 rounding happen in registers and the three layers
// (and the three transposed products of the backward) chain without a single shuffle or LDS round trip.  A lane's four
// consecutive features are also 16 contiguous bytes of an fp32 row (8 of a bf16 row): x is read, and dx and the factor
// matrices are written, straight from / to the operand layout.
// A wave carries two column blocks (32 voxel rows) per trip so that a weight fragment read from LDS feeds two MFMAs; a
// workgroup (4 waves) takes 128 rows per trip and stays resident (weights are staged into LDS once per workgroup).
// Outputs (forward) and output gradients (backward) travel through a per-wave LDS tile in the five tensors' own layout, 16
// bytes per lane; the backward's products with W^T read the same row-major LDS image through ds_read_b64_tr_b16; the factor
// matrices of the parameter gradients leave as bf16 rows, the activations with a column of ones (bias gradients ride along
// in the weight-gradient products of lara_gemm_tn_bf16).
// 65 MFMAs per 16 rows forward, 130 backward (~50 / ~100 us of matrix-core time per 10^6 rows at this instruction's rate);
// HBM: forward 320 + 88 K bytes per row, backward 320 + 88 K in, 320 + 1008 out.  Measured 150 / 584 us per 10^6 rows
// (3.5 / 3.1 TB/s): a trip's load, MFMA and store phases do not overlap inside a wave.
#include "common.h"
#include "../../include/lara_coarsedec.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned short f2bf(const float f) {  // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(const unsigned short h) { return __uint_as_float((unsigned)h << 16); }

constexpr int CD_F = 80;      // features in, hidden
constexpr int CD_O = 48;      // padded outputs (22 K <= 48)
constexpr int CD_LDW = 84;    // bf16 elements per LDS row of an 80-wide matrix (168 B: 8-byte aligned fragments, rows spread over the banks)
constexpr int CD_PTS = 128;   // voxel rows per workgroup trip
constexpr int CD_NB = 208;    // bias entries: 80 + 80 + 48
constexpr int CD_FA = 88;     // row width of the activation factor matrices: 80 features, a column of ones, 7 of zeros
constexpr int CD_TILE = 16 * CD_O + 16 * 12;  // floats of a wave's IO tile: 16 rows of the five tensors + the backward's copy of `offset` (16 x 3K floats, K <= 4 since 10 K <= CD_O)
constexpr int CD_WG_FWD = 1024, CD_WG_BWD = 512;

struct CdP {
    int M, n_par;
    const float *x, *w1, *b1, *w2, *b2, *w3, *b3;
    float *out[5];            // offset, sh, scaling, rotation, opacity
    const float *dout[5];
    const float *offset_out;
    float *dx;
    unsigned short *xb, *h1, *h2, *dz1, *dz2, *dz3;
    float opacity_shift, scaling_shift;
    int width[5], toff[5];   // floats per row of each output tensor; offset of its 16-row block in a wave's LDS tile
    signed char tensor[CD_O], col[CD_O], act[CD_O];   // per output p of the last layer: which tensor, which column, which activation
};

// ROWS x 80 fp32 [row-major] -> bf16 in LDS, rows >= rows_valid as zeros; transposed: dst[c][r].  The trip count is a
// compile-time constant and the loop is unrolled: all of a thread's loads are in flight together (left rolled, a workgroup
// spent ~25 dependent L2 round trips per matrix before its first MFMA).
template <int ROWS>
__device__ __forceinline__ void stage_matrix(const float *__restrict__ w, const int rows_valid, unsigned short *dst,
                                             const int ld, const bool transposed) {
    constexpr int N = ROWS * CD_F, TRIPS = (N + 255) / 256;
    float v[TRIPS];
#pragma unroll
    for (int u = 0; u < TRIPS; u++) {
        const int idx = u * 256 + (int)threadIdx.x;
        v[u] = (idx < N && idx < rows_valid * CD_F) ? w[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < TRIPS; u++) {
        const int idx = u * 256 + (int)threadIdx.x;
        if (idx >= N) continue;
        const int r = idx / CD_F, c = idx - r * CD_F;
        if (transposed) dst[c * ld + r] = f2bf(v[u]); else dst[r * ld + c] = f2bf(v[u]);
    }
}
__device__ __forceinline__ void stage_bias(const float *__restrict__ b, const int n_valid, const int n, float *dst) {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = i < n_valid ? bf2f(f2bf(b[i])) : 0.f;   // autocast: bias.to(bf16)
}

// acc[i][c] = bias block i + sum_t W[16 i.., 16 t..] . in[t][c]   (W: LDS, row-major [NI*16][ld], K contiguous)
// TR: the product with W^T from the SAME row-major image (rows = the contraction index): gfx950's transposing read
// ds_read_b64_tr_b16 -- the 16 lanes of a group address a [4 rows][16 columns] block (lane l: row l >> 2, columns
// 4 (l & 3)..+3) and lane l receives column l of the four rows -- hands every lane its A fragment of W^T in one instruction.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
template <int NI, int NT, bool TR>
__device__ __forceinline__ void mm(const unsigned short *Ws, const int ld, const float *bs, const s16x4 (&in)[NT][2],
                                   f32x4 (&acc)[NI][2], const int c16, const int q4) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (bs) b = *(const f32x4 *)(bs + 16 * i + 4 * q4);
        acc[i][0] = b; acc[i][1] = b;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            s16x4 a;
            if (TR) {
                const unsigned short *src = Ws + (16 * t + 4 * q4 + (c16 >> 2)) * ld + 16 * i + 4 * (c16 & 3);
                a = __builtin_bit_cast(s16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)src));
            } else {
                a = *(const s16x4 *)(Ws + (16 * i + c16) * ld + 16 * t + 4 * q4);
            }
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, in[t][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, in[t][1], acc[i][1], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ s16x4 pack4(const float a, const float b, const float c, const float d) {
    s16x4 h;
    h[0] = (short)f2bf(a); h[1] = (short)f2bf(b); h[2] = (short)f2bf(c); h[3] = (short)f2bf(d);
    return h;
}

// x rows base + 16 c + (lane & 15), features 16 t + 4 q4 ..+3 -> bf16 fragments (rows beyond M read row M - 1)
__device__ __forceinline__ void load_x(const float *__restrict__ x, const int M, const int base, const int c16, const int q4,
                                       s16x4 (&xb)[5][2]) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const float4 *row = (const float4 *)(x + (size_t)min(base + 16 * c + c16, M - 1) * CD_F);
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const float4 v = row[4 * t + q4];
            xb[t][c] = pack4(v.x, v.y, v.z, v.w);
        }
    }
}

template <int NI>
__device__ __forceinline__ void relu_pack(const f32x4 (&acc)[NI][2], s16x4 (&h)[NI][2]) {
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int c = 0; c < 2; c++)
            h[i][c] = pack4(fmaxf(acc[i][c][0], 0.f), fmaxf(acc[i][c][1], 0.f), fmaxf(acc[i][c][2], 0.f), fmaxf(acc[i][c][3], 0.f));
}

// 8-byte stores of bf16 fragments into a row-major [rows][ld] matrix.  ONES: the matrix is an activation operand of a
// weight-gradient product, stored CD_FA = 88 wide with column 80 = 1 and 81..87 = 0 -- the product dz^T [act | 1] then
// carries the bias gradient (the column sums of dz) in its column 80, and no kernel has to sum dz separately
template <int NI, bool ONES>
__device__ __forceinline__ void store_frags(unsigned short *dst, const int ld, const int base, const int c16, const int q4,
                                            const s16x4 (&h)[NI][2]) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
        unsigned short *row = dst + (size_t)(base + 16 * c + c16) * ld + 4 * q4;
#pragma unroll
        for (int i = 0; i < NI; i++) *(s16x4 *)(row + 16 * i) = h[i][c];
        if (ONES && q4 < 2) {
            s16x4 one = {0, 0, 0, 0};
            if (q4 == 0) one[0] = (short)0x3f80;   // bf16 1.0
            *(s16x4 *)(row + 16 * NI) = one;
        }
    }
}

__global__ void __launch_bounds__(256)
coarse_fwd_kernel(const CdP p) {
    __shared__ __attribute__((aligned(16))) unsigned short W1s[CD_F * CD_LDW], W2s[CD_F * CD_LDW], W3s[CD_O * CD_LDW];
    __shared__ __attribute__((aligned(16))) float bs[CD_NB];
    __shared__ __attribute__((aligned(16))) float tiles[4 * CD_TILE];
    __shared__ int s_width[5], s_toff[5];
    __shared__ signed char s_tensor[CD_O], s_col[CD_O], s_act[CD_O];
    stage_matrix<CD_F>(p.w1, CD_F, W1s, CD_LDW, false);
    stage_matrix<CD_F>(p.w2, CD_F, W2s, CD_LDW, false);
    stage_matrix<CD_O>(p.w3, p.n_par, W3s, CD_LDW, false);
    stage_bias(p.b1, CD_F, CD_F, bs);
    stage_bias(p.b2, CD_F, CD_F, bs + CD_F);
    stage_bias(p.b3, p.n_par, CD_O, bs + 2 * CD_F);
    if (threadIdx.x < CD_O) {
        s_tensor[threadIdx.x] = p.tensor[threadIdx.x]; s_col[threadIdx.x] = p.col[threadIdx.x]; s_act[threadIdx.x] = p.act[threadIdx.x];
    }
    if (threadIdx.x < 5) { s_width[threadIdx.x] = p.width[threadIdx.x]; s_toff[threadIdx.x] = p.toff[threadIdx.x]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, q4 = lane >> 4;
    const int trips = (p.M + CD_PTS - 1) / CD_PTS;
    // where this lane's 12 values of the last layer go in the tile, and through which activation: loop invariants (the
    // memory clobbers of the wave-level LDS hand-offs below would otherwise make every trip fetch the tables again)
    int slot[3][4];
    uint32_t acts = 0u;   // 2 bits per value
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = 16 * i + 4 * q4 + r, t = s_tensor[q];
            slot[i][r] = q < p.n_par ? s_toff[t] + c16 * s_width[t] + s_col[q] : -1;
            acts |= (uint32_t)s_act[q] << (2 * (4 * i + r));
        }
    for (int trip = blockIdx.x; trip < trips; trip += gridDim.x) {
        const int base = trip * CD_PTS + wave * 32;
        if (base >= p.M) continue;
        s16x4 xb[5][2], h[5][2];
        f32x4 acc[5][2], z[3][2];
        load_x(p.x, p.M, base, c16, q4, xb);
        mm<5, 5, false>(W1s, CD_LDW, bs, xb, acc, c16, q4);
        relu_pack<5>(acc, h);
        mm<5, 5, false>(W2s, CD_LDW, bs + CD_F, h, acc, c16, q4);
        relu_pack<5>(acc, xb);
        mm<3, 5, false>(W3s, CD_LDW, bs + 2 * CD_F, xb, z, c16, q4);
        // the last layer's bf16 result, widened (`.float()`), through the split's activations -- parked in the wave's LDS
        // tile in the OUTPUT tensors' own layout (tensor t's 16 rows x width_t floats back to back: the 16 voxel rows of a
        // column block are one contiguous piece of every output tensor), then copied out 16 bytes per lane
        float *tile = tiles + wave * CD_TILE;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int row0 = base + 16 * c, rows = min(16, p.M - row0);
            if (rows <= 0) break;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (slot[i][r] < 0) continue;
                    const int a = (acts >> (2 * (4 * i + r))) & 3u;
                    float v = bf2f(f2bf(z[i][c][r]));
                    if (a == 1) v = 2.0f / (1.0f + __expf(-v)) - 1.0f;
                    else if (a == 2) v += p.scaling_shift;
                    else if (a == 3) v += p.opacity_shift;
                    tile[slot[i][r]] = v;
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const int wd = p.width[t], n = rows * wd;     // floats of this tensor's piece; 16 wd is a multiple of 4
                float *dst = p.out[t] + (size_t)row0 * wd;
                const float *src = tile + p.toff[t];
                for (int j = 4 * lane; j < n; j += 256) {
                    if (j + 4 <= n) *(float4 *)(dst + j) = *(const float4 *)(src + j);
                    else for (int e = j; e < n; e++) dst[e] = src[e];
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is free again
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
coarse_bwd_kernel(const CdP p) {
    __shared__ __attribute__((aligned(16))) unsigned short W1s[CD_F * CD_LDW], W2s[CD_F * CD_LDW], W3s[CD_O * CD_LDW];
    __shared__ __attribute__((aligned(16))) float bs[2 * CD_F];
    __shared__ __attribute__((aligned(16))) float tiles[8 * CD_TILE];
    __shared__ int s_width[5], s_toff[5];
    __shared__ signed char s_tensor[CD_O], s_col[CD_O], s_act[CD_O];
    stage_matrix<CD_F>(p.w1, CD_F, W1s, CD_LDW, false);
    stage_matrix<CD_F>(p.w2, CD_F, W2s, CD_LDW, false);
    stage_matrix<CD_O>(p.w3, p.n_par, W3s, CD_LDW, false);
    stage_bias(p.b1, CD_F, CD_F, bs);
    stage_bias(p.b2, CD_F, CD_F, bs + CD_F);
    if (threadIdx.x < CD_O) {
        s_tensor[threadIdx.x] = p.tensor[threadIdx.x]; s_col[threadIdx.x] = p.col[threadIdx.x]; s_act[threadIdx.x] = p.act[threadIdx.x];
    }
    if (threadIdx.x < 5) { s_width[threadIdx.x] = p.width[threadIdx.x]; s_toff[threadIdx.x] = p.toff[threadIdx.x]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, q4 = lane >> 4;
    const int trips = (p.M + CD_PTS - 1) / CD_PTS;
    int slot[3][4];   // loop invariants: see the forward kernel
    uint32_t is_offset = 0u;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = 16 * i + 4 * q4 + r, t = s_tensor[q];
            slot[i][r] = q < p.n_par ? s_toff[t] + c16 * s_width[t] + s_col[q] : -1;
            is_offset |= (s_act[q] == 1 ? 1u : 0u) << (4 * i + r);
        }
    for (int trip = blockIdx.x; trip < trips; trip += gridDim.x) {
        const int base = trip * CD_PTS + wave * 32;
        s16x4 a[5][2], b[5][2];
        f32x4 acc[5][2];
        uint32_t live1[2] = {0u, 0u}, live2[2] = {0u, 0u};   // ReLU masks: bit 4 i + e of column block c
        // ---- first of all, the incoming gradients on their way: a column block's 16 rows are one contiguous piece of every
        // gradient tensor -- copied into the wave's two LDS tiles 16 bytes per lane (the forward's `offset` output behind them);
        // they are picked up in the operand layout after the forward recomputation, by which time they have long arrived
        float *tile = tiles + wave * 2 * CD_TILE;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int row0 = base + 16 * c, rows = max(0, min(16, p.M - row0));
#pragma unroll
            for (int t = 0; t < 6; t++) {
                const int wd = p.width[t < 5 ? t : 0], n = rows * wd;
                const float *src = t < 5 ? p.dout[t] : (p.dout[0] ? p.offset_out : nullptr);   // (absent gradient: zeros)
                float *dst = tile + c * CD_TILE + (t < 5 ? p.toff[t] : 16 * CD_O);
                for (int j = 4 * lane; j < 16 * wd; j += 256) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (src) {
                        const float *g = src + (size_t)row0 * wd + j;
                        if (j + 4 <= n) v = *(const float4 *)g;
                        else if (j < n) { v.x = g[0]; if (j + 1 < n) v.y = g[1]; if (j + 2 < n) v.z = g[2]; }
                    }
                    *(float4 *)(dst + j) = v;
                }
            }
        }
        // ---- forward again, up to the second hidden layer; the operands of the weight gradients go out as they appear
        load_x(p.x, p.M, base, c16, q4, a);
        store_frags<5, true>(p.xb, CD_FA, base, c16, q4, a);
        mm<5, 5, false>(W1s, CD_LDW, bs, a, acc, c16, q4);
        relu_pack<5>(acc, b);
        store_frags<5, true>(p.h1, CD_FA, base, c16, q4, b);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) live1[c] |= (b[i][c][e] > 0 ? 1u : 0u) << (4 * i + e);   // as a signed short: a positive bf16
        mm<5, 5, false>(W2s, CD_LDW, bs + CD_F, b, acc, c16, q4);
        relu_pack<5>(acc, a);
        store_frags<5, true>(p.h2, CD_FA, base, c16, q4, a);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) live2[c] |= (a[i][c][e] > 0 ? 1u : 0u) << (4 * i + e);
        // ---- dz3: the gradients through the activations, rounded to bf16 (the `.float()`'s backward), in the operand layout
        s16x4 g3[3][2];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the tiles filled at the top of the trip are complete
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float *tl = tile + c * CD_TILE;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[r] = 0.f;
                    if (slot[i][r] < 0) continue;
                    float g = tl[slot[i][r]];
                    if ((is_offset >> (4 * i + r)) & 1u) {   // o = 2 s - 1  ->  do/dz = 2 s (1 - s) = (1 - o^2) / 2   (offset is tensor 0, at offset 0: same index)
                        const float o = tl[16 * CD_O + slot[i][r]];
                        g *= 0.5f * (1.0f - o * o);
                    }
                    v[r] = g;
                }
                g3[i][c] = pack4(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tiles are free for the next trip
        store_frags<3, false>(p.dz3, CD_O, base, c16, q4, g3);
        // ---- dz2 = relu'(h2) . bf16(W3^T dz3)
        mm<5, 3, true>(W3s, CD_LDW, nullptr, g3, acc, c16, q4);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const bool in_range = base + 16 * c + c16 < p.M;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned short h = (in_range && ((live2[c] >> (4 * i + e)) & 1u)) ? f2bf(acc[i][c][e]) : (unsigned short)0;
                    a[i][c][e] = (short)h;
                }
            }
        store_frags<5, false>(p.dz2, CD_F, base, c16, q4, a);
        // ---- dz1 = relu'(h1) . bf16(W2^T dz2)
        mm<5, 5, true>(W2s, CD_LDW, nullptr, a, acc, c16, q4);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const bool in_range = base + 16 * c + c16 < p.M;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned short h = (in_range && ((live1[c] >> (4 * i + e)) & 1u)) ? f2bf(acc[i][c][e]) : (unsigned short)0;
                    b[i][c][e] = (short)h;
                }
            }
        store_frags<5, false>(p.dz1, CD_F, base, c16, q4, b);
        // ---- dx = W1^T dz1, fp32
        mm<5, 5, true>(W1s, CD_LDW, nullptr, b, acc, c16, q4);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int row = base + 16 * c + c16;
            if (row >= p.M) continue;
            float4 *dst = (float4 *)(p.dx + (size_t)row * CD_F) + q4;
#pragma unroll
            for (int i = 0; i < 5; i++) dst[4 * i] = make_float4(acc[i][c][0], acc[i][c][1], acc[i][c][2], acc[i][c][3]);
        }
    }
}

// the split of network.py:268-270: [offset 3 | sh | opacity 1 | scaling 2 | rotation 4] per k -> (tensor, column, activation)
int fill_map(CdP &p, const int K, const int sh_dim) {
    const int per = 10 + sh_dim;
    if (K < 1 || sh_dim < 0 || K * per > CD_O) return LARA2DGS_E_INVALID;
    static_assert(CD_TILE - 16 * CD_O >= 16 * 3 * (CD_O / 10), "the backward's copy of `offset` (16 x 3K floats) must fit the tile's tail for every K with 10 K <= CD_O");
    p.n_par = K * per;
    const int widths[5] = {3, sh_dim, 2, 4, 1};   // output tensors: offset, sh, scaling, rotation, opacity
    for (int t = 0, o = 0; t < 5; t++) { p.width[t] = K * widths[t]; p.toff[t] = o; o += 16 * p.width[t]; }
    for (int q = 0; q < CD_O; q++) { p.tensor[q] = 0; p.col[q] = 0; p.act[q] = 0; }
    for (int k = 0; k < K; k++)
        for (int j = 0; j < per; j++) {
            int t, c, a = 0;
            if (j < 3) { t = 0; c = j; a = 1; }
            else if (j < 3 + sh_dim) { t = 1; c = j - 3; }
            else if (j < 4 + sh_dim) { t = 4; c = 0; a = 3; }
            else if (j < 6 + sh_dim) { t = 2; c = j - 4 - sh_dim; a = 2; }
            else { t = 3; c = j - 6 - sh_dim; }
            const int q = k * per + j;
            p.tensor[q] = (signed char)t; p.col[q] = (signed char)(k * widths[t] + c); p.act[q] = (signed char)a;
        }
    return LARA2DGS_OK;
}

}  // namespace

extern "C" {

int64_t lara_coarse_decoder_padded_rows(int64_t M) { return (M + CD_PTS - 1) / CD_PTS * CD_PTS; }

int lara_coarse_decoder_forward(int32_t M, int32_t K, int32_t sh_dim, const float *x, const float *w1, const float *b1,
                                const float *w2, const float *b2, const float *w3, const float *b3, float opacity_shift,
                                float scaling_shift, float *offset, float *sh, float *scaling, float *rotation,
                                float *opacity, void *stream) {
    if (M < 0 || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) return LARA2DGS_E_INVALID;
    CdP p{};
    if (int rc = fill_map(p, K, sh_dim)) return rc;
    if (M == 0) return LARA2DGS_OK;
    if (!x || !offset || !scaling || !rotation || !opacity || (sh_dim > 0 && !sh)) return LARA2DGS_E_INVALID;
    p.M = M; p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3;
    p.out[0] = offset; p.out[1] = sh; p.out[2] = scaling; p.out[3] = rotation; p.out[4] = opacity;
    p.opacity_shift = opacity_shift; p.scaling_shift = scaling_shift;
    const int trips = (M + CD_PTS - 1) / CD_PTS;
    {
        L2D_PROF("coarse_decoder_fwd", (hipStream_t)stream);
        hipLaunchKernelGGL(coarse_fwd_kernel, dim3(trips < CD_WG_FWD ? trips : CD_WG_FWD), dim3(256), 0, (hipStream_t)stream, p);
    }
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

int lara_coarse_decoder_backward(int32_t M, int32_t K, int32_t sh_dim, const float *x, const float *w1, const float *b1,
                                 const float *w2, const float *b2, const float *w3, const float *offset_out,
                                 const float *d_offset, const float *d_sh, const float *d_scaling, const float *d_rotation,
                                 const float *d_opacity, float *dx, uint16_t *xb, uint16_t *h1, uint16_t *h2,
                                 uint16_t *dz1, uint16_t *dz2, uint16_t *dz3, void *stream) {
    if (M < 0 || !w1 || !b1 || !w2 || !b2 || !w3) return LARA2DGS_E_INVALID;
    CdP p{};
    if (int rc = fill_map(p, K, sh_dim)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return LARA2DGS_OK;
    if (!x || !dx || !xb || !h1 || !h2 || !dz1 || !dz2 || !dz3 || (d_offset && !offset_out)) return LARA2DGS_E_INVALID;
    p.M = M; p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3;
    p.dout[0] = d_offset; p.dout[1] = d_sh; p.dout[2] = d_scaling; p.dout[3] = d_rotation; p.dout[4] = d_opacity;
    p.offset_out = offset_out; p.dx = dx;
    p.xb = xb; p.h1 = h1; p.h2 = h2; p.dz1 = dz1; p.dz2 = dz2; p.dz3 = dz3;
    const int trips = (M + CD_PTS - 1) / CD_PTS;
    const int grid = trips < CD_WG_BWD ? trips : CD_WG_BWD;
    {
        L2D_PROF("coarse_decoder_bwd", s);
        hipLaunchKernelGGL(coarse_bwd_kernel, dim3(grid), dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

}  // extern "C"
