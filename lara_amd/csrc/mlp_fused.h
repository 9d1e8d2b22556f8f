// mlp_fused.h -- the MLP half of LaRa's GroupAttBlock as ONE kernel (lightning/network.py:94 and the norm in front of the
// convolution, :96):
//
//     x2 = x1 + fc2(gelu(fc1(norm2(x1))))          xn3 = norm3(x2),  (mean, rstd) of x2's rows
//
// Rounds 2-5 ran four launches for it (LayerNorm+cast 41 us, fc1 91 us, fc2 98 us, LayerNorm+cast 41 us at LaRa's
// 131 072 token rows): the two products are HBM-bound on what lies BETWEEN them -- the bf16 normalised rows, the 134 MB hidden
// tensor written and read back, the fp32 residual stream read twice more -- and every fusion with ONE neighbour that was tried
// (norm2 into the attention, fc2 + norm3, the products on the ring kernel) measured slower.  Here the whole chain runs per
// 128-row tile and nothing intermediate travels through HBM except what a backward needs:
//
//   * a workgroup of 8 waves owns 128 token rows.  Phase 0: the rows are read once (fp32, one wave per row), normalised,
//     rounded to bf16 and dealt through LDS into MFMA operand fragments that every wave then HOLDS IN REGISTERS for the rest
//     of the kernel (64 VGPRs: its 32 tokens x 256 channels);
//   * the hidden dimension is walked in four chunks of 128 units.  Per chunk: fc1 TRANSPOSED -- acc1[h][token] = W1 . xn^T,
//     wave tile 64 hidden x 32 tokens -- so that a lane ends up with runs of four consecutive hidden units of one token:
//     bias + GELU + bf16 rounding in registers, then 8-byte LDS stores drop the chunk straight into the K-contiguous layout
//     fc2 reads its A operand in (32 KB; no transpose pass, no shuffles); fc2's partial product of the chunk,
//     acc2[token][out] += h_chunk . W2[:, chunk]^T (wave tile 64 tokens x 64 outputs), stays in registers over the chunks;
//   * both weight matrices stream L2 -> LDS with global_load_lds (16 KB slots, a ring of four, two slots in flight behind
//     the one being multiplied; 64-byte rows with the XOR chunk swizzle of mfma_gemm.h's ring kernels), one barrier per slot;
//   * the epilogue bounces acc2 through LDS per wave (32 x 64 at a time), adds bias and the residual row (re-read: L2 / MALL),
//     writes x2 -- and, the tile holding whole rows, forms norm3 in place: row sums as DPP row totals exchanged between the
//     four waves of a wave row, two passes (mean, then centred squares: ln_cast_kernel's arithmetic), bf16 rows + (mean, rstd).
//
// TRAIN = the forward of a training step also leaves norm2's bf16 rows, the pre-activation z and the hidden h (what the
// block's backward reads: lara_groupblock_backward); the hidden chunk is copied out of LDS in 16-byte row-contiguous pieces,
// z goes out as the 8-byte runs the accumulator layout yields.
//
// HBM bytes per token row: 1 KB in + (TRAIN: 0.5 + 1 + 1) + 1 + 0.5 KB out = 5 KB against 8.5 KB for the four launches
// (inference: 2.5 KB against 6.5).  Matrix work: 0.5 MFLOP per row = 68.7 GFLOP per layer at LaRa's size.
#pragma once
#include "mfma_gemm.h"

namespace {

#ifndef MF_TIMING
#define MF_TIMING 0                        // timing-only builds (results invalid; tools/build_variant.sh -DMF_TIMING=n): 1 = no epilogue, 8 = no weight stream,
#endif                                     // 2 = no products (weight stream + MFMAs), 4 = no phase 0
// NW = waves per workgroup: 8 (128 token rows, one workgroup per CU) or 4 (64 rows, 69 KB of LDS: TWO workgroups per CU, whose
// memory phases -- the tile's rows in, the results out -- run under the other one's products; the weights then travel L2 -> LDS twice
// as often, 1 GB per layer at LaRa's size).
#ifndef MF_NW
#define MF_NW 8
#endif
constexpr int MF_SLOT = 16384;             // bytes per weight slot
constexpr int MF_CONST = 5 * 1024;         // b1 (512) | b2 (256) | ln3 gamma (256) | ln3 beta (256) floats
#ifndef MF_SLOTS0
#define MF_SLOTS0 4        // ring slots of the inference instantiation
#endif
#ifndef MF_SLOTS1
#define MF_SLOTS1 4        // ... of the training forward and the backward (z chunk resident)
#endif
template <int NW, int MODE = 1> struct MfCfg {
    static constexpr int TM = 16 * NW;                     // token rows per workgroup
    // (Ring depth, measured on one box, inference / training forward + backward products: 4 slots 171.6 / 455 + 145 us, 5 and 3
    // slots 179.5 / 460 + 153, 6 and 4: 182.8 / 458 + 147, 7 and 5: 183.4 / 456 + 148 -- more bytes in flight do not speed the
    // weight stream up; -DMF_SLOTS0 / -DMF_SLOTS1 for A/B builds.)
    static constexpr int SLOTS = NW == 8 ? (MODE == 0 ? MF_SLOTS0 : MF_SLOTS1) : 3;          // ring
    static constexpr int AHEAD = SLOTS - 2;                // slots in flight behind the one being multiplied (see the loop)
    static constexpr int PW = 16 / NW;                     // one-KB pieces of a slot per wave
    static constexpr int HC = TM * 256;                    // the hidden chunk: [4 panels of 32 units][TM tokens][64 B]
    static constexpr int ZC = MODE == 0 ? 0 : TM * 256;    // the chunk's pre-activations z, same layout (training forward: out; backward: in)
    static constexpr int LDS = HC + ZC + SLOTS * MF_SLOT + MF_CONST;   // (phase 0's TM x 512 B of normalised rows and the
                                                                       // epilogue's bounce tiles alias the chunk + ring region)
    static_assert(HC + SLOTS * MF_SLOT >= TM * 512 && HC + SLOTS * MF_SLOT >= NW * 8704 + 2048, "phase 0 / the epilogue alias chunk + ring");
    static_assert(LDS <= 163840, "one workgroup's LDS");
};

struct MlpP {
    const float *x1;                  // [M, 256] fp32: the block's activations behind the attention step
    float *x2;                        // [M, 256] fp32 out (may be x1: a workgroup reads its rows before it writes them)
    const float *ln2_w, *ln2_b, *b1, *b2, *ln3_w, *ln3_b;
    const unsigned short *w1, *w2;    // bf16 [512, 256], [256, 512]
    unsigned short *xn3;              // [M + 1, 256] bf16 out: norm3(x2); row M is zero-filled (the convolution's padding voxels gather it:
                                      // a 512-byte memset launch per layer until round 6)
    float2 *stats;                    // [M] out: (mean, rstd) of x2's rows
    unsigned short *xn2, *z, *h;      // TRAIN: bf16 [M, 256], [M, 512], [M, 512]
    float eps;
    int M;
    // MODE 2 (the backward of the same chain, see mlp_fused_kernel): w1 = W2^T [512, 256], w2 = W1^T [256, 512]; x1, ln2_w, eps, z as
    // above (z is READ); h = dz out (bf16 [M, 512])
    const unsigned short *gin;        // bf16 [M, 256]: the gradient of x2 (what norm3's backward left)
    float *g;                         // fp32 [M, 256] in / out: the residual stream's gradient; out = g + norm2's backward
    unsigned short *gout;             // bf16 [M, 256] out: the same, rounded
    float *part_ln;                   // [row tiles][3][256]: per-tile column sums of dy xhat, dy, out (dgamma, dbeta of norm2, -)
    float *part_b1;                   // [row tiles][512]: per-tile column sums of dz (the bias gradient of fc1)
};

// MODE 0: inference forward; 1: training forward (TRAIN); 2: the BACKWARD of the same chain, which has the same shape --
//     dz = (g2 . W2) * gelu'(z)            <-> fc1 with W2^T as its weight, the activation a product with gelu'(z) read from HBM
//     dy = dz . W1   (rounded to bf16)      <-> fc2 with W1^T as its weight
//     g  = g + norm2_backward(dy; x1)       <-> the epilogue
// (network.py:94 backwards; rounds 2-5: two products, 175 us, + a LayerNorm backward pass, 98 us, per layer.)  Phase 0 copies the bf16
// gradient rows into the fragment layout (no LayerNorm); the chunk's dz leaves through LDS for the weight gradient of fc1, its
// per-tile column sums (fc1's bias gradient) are taken on the way; the epilogue re-deals the tile so that a wave owns WHOLE rows
// (64 rows x 256 floats at a time through LDS) and runs ln_bwd_kernel's arithmetic on them, operation for operation.
template <int MODE, int NW>
__global__ void __launch_bounds__(64 * NW)
mlp_fused_kernel(const MlpP p) {
    constexpr bool TRAIN = MODE == 1;
    using Cfg = MfCfg<NW, MODE>;
    constexpr int TM = Cfg::TM, PANEL = TM * 64, MF_SLOTS = Cfg::SLOTS, MF_AHEAD = Cfg::AHEAD, PW = Cfg::PW, MF_HC = Cfg::HC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const hc = lds;                       // hidden chunk
    unsigned char *const ring = lds + MF_HC;             // weight slots
    unsigned char *const zc = lds + MF_HC + MF_SLOTS * MF_SLOT;      // z chunk
    float *const cst = (float *)(lds + MF_HC + MF_SLOTS * MF_SLOT + Cfg::ZC);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int M = p.M, bm0 = blockIdx.x * TM;
    // fc1 (transposed) wave tile: hidden rows wh * 64 .. + 64 of the chunk, tokens wt * 32 .. + 32
    const int wh = wave / (NW / 2), wt = wave % (NW / 2);
    // fc2 wave tile: tokens wr * 64 .. + 64, outputs wc * 64 .. + 64
    const int wr = wave >> 2, wc = wave & 3;

#ifdef MF_STAGGER
    // Stagger: the workgroups of a round run their phases in lock-step -- every CU reads its rows at the same time, then every CU
    // streams the weights from L2, then every CU writes -- so each shared resource is idle two thirds of the time.  Holding back every
    // other workgroup of the FIRST round by about half a tile's time puts the two halves of the chip out of phase for the rest of the
    // launch (a tile takes the same time everywhere, so the offset persists).
    if (blockIdx.x < 256 && (blockIdx.x & 1))
        for (int i = 0; i < MF_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
#endif
    // ---- the weight stream: slot n of a tile = (chunk c = n / 8, k = n % 8): k < 4: W1 rows c*128 .. +128, channels 64 k .. +64 as
    //      two 32-channel sub-tiles [2][128 rows][64 B]; k >= 4: W2 rows 0 .. 256, hidden units c*128 + 32 (k-4) .. +32 as [256 rows][64 B].
    //      A slot is 16 one-KB pieces (16 rows x 64 B, lane L -> row L / 4, physical chunk L & 3 = logical chunk ^ ((row >> 2) & 3));
    //      wave w moves pieces 2 w and 2 w + 1.
    const char *W1b = (const char *)p.w1, *W2b = (const char *)p.w2;
    uint32_t off1[PW], off2[PW];    // per piece: byte offset of this lane's 16 bytes at chunk 0, k = 0
#pragma unroll
    for (int q = 0; q < PW; q++) {
        const int piece = PW * wave + q;
        {   // W1 slot: sub-tile piece / 8, rows 16 (piece % 8) + L / 4
            const int sub = piece >> 3, row = 16 * (piece & 7) + (lane >> 2);
            off1[q] = (uint32_t)row * 512u + (uint32_t)sub * 64u + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) << 4);
        }
        {   // W2 slot: rows 16 piece + L / 4
            const int row = 16 * piece + (lane >> 2);
            off2[q] = (uint32_t)row * 1024u + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) << 4);
        }
    }
    // (The piece is inline assembly: behind the compiler's own global_load_lds its waitcnt pass puts vmcnt(0) in front of the next LDS
    // read -- the first build of this kernel waited out every slot it had just requested -- and drains lgkmcnt in front of every MFMA.
    // Invisible to that pass, the pieces are only ever waited for by the counted vmcnt in the loop below; see gemm_ring2_kernel.)
    const uint32_t ring_a = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)ring + wave * (PW * 1024);
#define MF_DMA(base, voff, ldsaddr) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"((uint32_t)(ldsaddr)), "v"((uint32_t)(voff)), "s"(base) : "memory")
    auto issue = [&](const int n) {      // this wave's pieces of slot n (n < 32)
        if (MF_TIMING & 8) return;       // (timing-only: no weight stream -- the products run on whatever the ring holds)
        const int c = n >> 3, k = n & 7;
        const uint32_t dst = ring_a + (n % MF_SLOTS) * MF_SLOT;
#pragma unroll
        for (int q = 0; q < PW; q++) {
            if (k < 4) MF_DMA(W1b, off1[q] + (uint32_t)c * (128u * 512u) + (uint32_t)k * 128u, dst + q * 1024);
            else MF_DMA(W2b, off2[q] + (uint32_t)c * 256u + (uint32_t)(k - 4) * 64u, dst + q * 1024);
        }
    };

    // ---- phase 0: norm2 of the tile's rows -> bf16 -> LDS [8 panels of 32 channels][128 tokens][64 B] (aliases chunk + ring)
    if (MODE >= 2) {
        if (MODE == 2 && tid < 64) ((float4 *)cst)[tid] = ((const float4 *)p.ln2_w)[tid];      // gamma of norm2
    } else
    for (int t = tid; t < 320; t += 64 * NW) {      // the constants: b1 | b2 | gamma3 | beta3
        const float *src = t < 128 ? p.b1 + t * 4 : t < 192 ? p.b2 + (t - 128) * 4 : t < 256 ? p.ln3_w + (t - 192) * 4 : p.ln3_b + (t - 256) * 4;
        ((float4 *)cst)[t] = *(const float4 *)src;
    }
    if (MODE >= 2) {
        uint2 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = ((const uint2 *)(p.gin + (size_t)min(bm0 + wave * 16 + i, M - 1) * 256))[lane];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int tok = wave * 16 + i;
            *(uint2 *)(lds + (lane >> 3) * PANEL + tok * 64 + ((((lane & 7) >> 1) ^ ((tok >> 2) & 3)) << 4) + (lane & 1) * 8) = v[i];
        }
    } else if (!(MF_TIMING & 4)) {
        const float4 g = ((const float4 *)p.ln2_w)[lane], be = ((const float4 *)p.ln2_b)[lane];
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = ((const float4 *)(p.x1 + (size_t)min(bm0 + wave * 16 + i, M - 1) * 256))[lane];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int tok = wave * 16 + i;
            float s = v[i].x + v[i].y + v[i].z + v[i].w;         // (ln_cast_kernel's arithmetic, operation for operation)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
            const float mean = s * (1.0f / 256.0f);
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d4 = v[i].w - mean;
            float q = a * a + b * b + c * c + d4 * d4;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
            const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + p.eps);
            uint2 o;
            o.x = f2bf2(a * rstd * g.x + be.x, b * rstd * g.y + be.y);
            o.y = f2bf2(c * rstd * g.z + be.z, d4 * rstd * g.w + be.w);
            // lane L holds channels 4 L .. 4 L + 3: panel L / 8, 16-byte chunk (L % 8) / 2, half L & 1
            *(uint2 *)(lds + (lane >> 3) * PANEL + tok * 64 + ((((lane & 7) >> 1) ^ ((tok >> 2) & 3)) << 4) + (lane & 1) * 8) = o;
            if (TRAIN && bm0 + tok < M) *(uint2 *)(p.xn2 + (size_t)(bm0 + tok) * 256 + lane * 4) = o;
        }
    }
    __syncthreads();
    // this wave's fc1 B operand: tokens wt * 32 + r, all 256 channels = 16 fragments, kept for the whole kernel
    bf16x8 xb[16];
    {
        const int tok = wt * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 16; ks++)
            xb[ks] = *(const bf16x8 *)(lds + (ks >> 1) * PANEL + tok * 64 + (((2 * (ks & 1) + kh) ^ ((tok >> 2) & 3)) << 4));
    }
    __syncthreads();      // the staging area is ring + chunk from here on

    // ---- the products
    f32x16 acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc2[i][j][e] = 0.f;
    f32x16 acc1[2];
    float *const csum = cst + 256;                 // MODE 2, 3: [NW waves][128] column sums of the chunk's dz
    (void)csum; (void)zc;

    // fragment offsets inside a slot / the chunk: row * 64 + ((K step's chunk pair 2 s + kh) ^ ((row >> 2) & 3)) * 16
    int a1off[2], a2off[2], b2off[2];      // [K step s]: fc1's A rows (W1: wh * 64 + r, + 32 i = + 2048 i), fc2's A rows (chunk: wr * 64 + r), fc2's B rows (W2: wc * 64 + r)
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int ra = wh * 64 + r, rt = wr * 64 + r, rb = wc * 64 + r;
        a1off[s] = ra * 64 + (((2 * s + kh) ^ ((ra >> 2) & 3)) << 4);
        a2off[s] = rt * 64 + (((2 * s + kh) ^ ((rt >> 2) & 3)) << 4);
        b2off[s] = rb * 64 + (((2 * s + kh) ^ ((rb >> 2) & 3)) << 4);
    }

    // One barrier per slot.  Iteration n: issue slot n + AHEAD (into the buffer slot n + AHEAD - SLOTS left: every wave finished
    // reading that one before it arrived at barrier n - 1, which this wave has passed -- hence SLOTS >= AHEAD + 2), wait for this
    // wave's pieces of slot n (counted: the 2 AHEAD younger pieces stay in flight; on gfx9 vector memory operations retire in issue
    // order, so younger STORES in the queue only make the wait conservative), barrier, multiply.
    if (!(MF_TIMING & 2)) {
#pragma unroll
    for (int n = 0; n < MF_AHEAD; n++) issue(n);
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc1[i][e] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int n = c * 8 + k;
            // (wait + barrier as ONE asm statement with a memory clobber: the barrier intrinsic alone does not order the compiler's
            // LDS accesses; lgkmcnt(0): this wave's chunk stores have landed before anybody is told to read them)
            // this wave's pieces of slot n have landed once at most PW * (slots issued behind it) are outstanding
            if (n + MF_AHEAD < 32) issue(n + MF_AHEAD);
            {
                constexpr int kMaxYounger = PW * MF_AHEAD;
                const int younger = PW * (n + MF_AHEAD < 32 ? MF_AHEAD : 31 - n);      // (n is a constant of the unrolled loop)
                static_assert(kMaxYounger <= 12, "add cases below");
                // (the backward's z pieces are requested at k == 0 behind that trip's slot and are needed at k == 3: with AHEAD <= 3 every
                // piece the k == 3 wait leaves outstanding is younger than they are)
                static_assert(MODE < 2 || MF_AHEAD <= 3, "the z chunk's pieces would be among the outstanding ones at k == 3");
                switch (younger) {
                case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                }
            }
            const unsigned char *slot = ring + (n % MF_SLOTS) * MF_SLOT;
            if (MODE >= 2 && k == 0) {
                // the chunk's pre-activations z [TM tokens x 128 units] into LDS, in the chunk buffer's layout, by DMA: 4 TM / 16 one-KB
                // pieces (16 tokens x 32 units each), 16-byte row-contiguous reads (the accumulator layout's own 8-byte gathers used a
                // quarter of every 64-byte sector they touched).  Needed at k == 3; the pieces are older than the weight slots that
                // iteration waits for, so its counted wait covers them.  (The buffer's last readers were the k == 3 epilogue of the
                // chunk before: four barriers ago.)
                const uint32_t zc_a = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)zc;
#pragma unroll
                for (int q = 0; q < (4 * TM / 16) / NW; q++) {
                    const int piece = q * NW + wave, pan = piece / (TM / 16), tok = 16 * (piece % (TM / 16)) + (lane >> 2);
                    const uint32_t off = (uint32_t)min(bm0 + tok, M - 1) * 1024u + (uint32_t)c * 256u + (uint32_t)pan * 64u +
                                         (uint32_t)(((lane & 3) ^ ((tok >> 2) & 3)) << 4);
                    MF_DMA((const char *)p.z, off, zc_a + piece * 1024);
                }
            }
            if (MODE >= 2 && k == 5 && tid < 128) {      // fc1's bias gradient of the chunk: the eight waves' column sums (see k == 4)
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) t += csum[w * 128 + tid];
                p.part_b1[(size_t)blockIdx.x * 512 + c * 128 + tid] = t;
            }
            if (k < 4) {
                // fc1^T: four K steps (sub-tile, s) of two MFMAs: acc1[i] += W1 rows (wh*64 + 32 i + r) . xn tokens (wt*32 + r)
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const unsigned char *sub = slot + (ks >> 1) * 8192;
                    const bf16x8 w0 = *(const bf16x8 *)(sub + a1off[ks & 1]);
                    const bf16x8 w1 = *(const bf16x8 *)(sub + a1off[ks & 1] + 2048);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xb[4 * k + ks], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xb[4 * k + ks], acc1[1], 0, 0, 0);
                }
                if (k == 3) {
                    // the chunk's epilogue: + bias -> z, GELU -> h, bf16, into the chunk buffer in fc2's A layout.  A lane holds, per MFMA
                    // tile i and register group e >> 2, four consecutive hidden units (e & 3) of token wt * 32 + r:
                    // unit u = wh * 64 + 32 i + 8 (e >> 2) + 4 kh + (e & 3) of the chunk -> panel 2 wh + i, 16-byte chunk e >> 2, half kh.
                    // (the buffer was last read by fc2 of the chunk before: its waves have all passed this slot's barrier)
                    const int tok = wt * 32 + r;
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int eg = 0; eg < 4; eg++) {
                            const int u0 = wh * 64 + 32 * i + 8 * eg + 4 * kh;
                            uint2 hh;
                            if (MODE >= 2) {      // dz = (g2 . W2) * gelu'(z)
                                const uint2 zz = *(const uint2 *)(zc + (2 * wh + i) * PANEL + tok * 64 + ((eg ^ ((tok >> 2) & 3)) << 4) + kh * 8);
                                hh.x = f2bf2(acc1[i][4 * eg] * gelu_erf_grad(bf2f((unsigned short)zz.x)),
                                             acc1[i][4 * eg + 1] * gelu_erf_grad(bf2f((unsigned short)(zz.x >> 16))));
                                hh.y = f2bf2(acc1[i][4 * eg + 2] * gelu_erf_grad(bf2f((unsigned short)zz.y)),
                                             acc1[i][4 * eg + 3] * gelu_erf_grad(bf2f((unsigned short)(zz.y >> 16))));
                            } else {
                            const float4 bs = *(const float4 *)(cst + c * 128 + u0);
                            const float z0 = acc1[i][4 * eg] + bs.x, z1 = acc1[i][4 * eg + 1] + bs.y;
                            const float z2 = acc1[i][4 * eg + 2] + bs.z, z3 = acc1[i][4 * eg + 3] + bs.w;
                            if (TRAIN) {      // z leaves through LDS like h (row-contiguous 16-byte stores, see k == 4)
                                uint2 zz;
                                zz.x = f2bf2(z0, z1); zz.y = f2bf2(z2, z3);
                                *(uint2 *)(zc + (2 * wh + i) * PANEL + tok * 64 + ((eg ^ ((tok >> 2) & 3)) << 4) + kh * 8) = zz;
                            }
                            hh.x = f2bf2(gelu_erf(z0), gelu_erf(z1)); hh.y = f2bf2(gelu_erf(z2), gelu_erf(z3));
                            }
                            *(uint2 *)(hc + (2 * wh + i) * PANEL + tok * 64 + ((eg ^ ((tok >> 2) & 3)) << 4) + kh * 8) = hh;
                        }
                }
            } else {
                if (k == 4 && (TRAIN || MODE >= 2)) {
                    // (the barrier of this slot published the chunk) h / dz out, row-contiguous: 128 tokens x 256 bytes.  MODE 2: the
                    // column sums of what is stored (fc1's bias gradient) on the way: a thread's pieces all belong to columns
                    // 8 (tid & 15) .. + 8; its four tokens, then the wave's four lanes of a column group, then (k == 5) the waves
                    float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int idx = u * (64 * NW) + tid, tok = idx >> 4, c16 = idx & 15;
                        const uint4 v = *(const uint4 *)(hc + (c16 >> 2) * PANEL + tok * 64 + (((c16 & 3) ^ ((tok >> 2) & 3)) << 4));
                        if (bm0 + tok < M) {
                            *(uint4 *)(p.h + (size_t)(bm0 + tok) * 512 + c * 128 + c16 * 8) = v;
                            if (TRAIN)
                                *(uint4 *)(p.z + (size_t)(bm0 + tok) * 512 + c * 128 + c16 * 8) =
                                    *(const uint4 *)(zc + (c16 >> 2) * PANEL + tok * 64 + (((c16 & 3) ^ ((tok >> 2) & 3)) << 4));
                            if (MODE >= 2) {
                                cs8[0] += __uint_as_float(v.x << 16); cs8[1] += __uint_as_float(v.x & 0xffff0000u);
                                cs8[2] += __uint_as_float(v.y << 16); cs8[3] += __uint_as_float(v.y & 0xffff0000u);
                                cs8[4] += __uint_as_float(v.z << 16); cs8[5] += __uint_as_float(v.z & 0xffff0000u);
                                cs8[6] += __uint_as_float(v.w << 16); cs8[7] += __uint_as_float(v.w & 0xffff0000u);
                            }
                        }
                    }
                    if (MODE >= 2) {
#pragma unroll
                        for (int q = 0; q < 8; q++) { cs8[q] += __shfl_xor(cs8[q], 16, 64); cs8[q] += __shfl_xor(cs8[q], 32, 64); }
                        if (lane < 16) {
                            *(float4 *)(csum + wave * 128 + lane * 8) = make_float4(cs8[0], cs8[1], cs8[2], cs8[3]);
                            *(float4 *)(csum + wave * 128 + lane * 8 + 4) = make_float4(cs8[4], cs8[5], cs8[6], cs8[7]);
                        }
                    }
                }
                // fc2: two K steps of four MFMAs: acc2[i][j] += h tokens (wr*64 + 32 i + r) . W2 rows (wc*64 + 32 j + r)
                const unsigned char *pan = hc + (k - 4) * PANEL;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const bf16x8 h0 = *(const bf16x8 *)(pan + a2off[s]), h1 = *(const bf16x8 *)(pan + a2off[s] + 2048);
                    const bf16x8 v0 = *(const bf16x8 *)(slot + b2off[s]), v1 = *(const bf16x8 *)(slot + b2off[s] + 2048);
                    acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, v0, acc2[0][0], 0, 0, 0);
                    acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, v1, acc2[0][1], 0, 0, 0);
                    acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, v0, acc2[1][0], 0, 0, 0);
                    acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, v1, acc2[1][1], 0, 0, 0);
                }
            }
        }
    }
    }
    __syncthreads();      // every fragment read is done: chunk + ring become the bounce tiles
    if (MF_TIMING & 1) {
        if (acc2[0][0][0] == 123.456f) p.x2[0] = acc2[0][0][0] + acc2[0][1][1] + acc2[1][0][2] + acc2[1][1][3];
        return;
    }

    if (MODE == 3) {
        // ---- the backward's products alone: dy = bf16(acc2) to HBM (p.gout), row-contiguous through LDS; norm2's backward stays a
        //      pass of its own (ln_bwd_kernel: many small workgroups at HBM speed, where the fused epilogue below serialises its
        //      0.47 GB behind the products of ONE workgroup per CU)
        unsigned short *bounce = (unsigned short *)lds;      // [64][256] bf16
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    bounce[(wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * 256 + wc * 64 + j * 32 + r] = f2bf(acc2[i][j][e]);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; u++) {      // 64 rows x 512 bytes = 2048 sixteen-byte pieces
                const int idx = u * 512 + tid, lrow = idx >> 5, c16 = idx & 31;
                const int row = bm0 + (lrow >> 5) * 64 + i * 32 + (lrow & 31);
                if (row < M) *(uint4 *)(p.gout + (size_t)row * 256 + c16 * 8) = *(const uint4 *)(bounce + lrow * 256 + c16 * 8);
            }
        }
        return;
    }
    if (MODE == 2) {
        // ---- epilogue of the backward: dy = bf16(acc2); g <- g + norm2's backward (ln_bwd_kernel's arithmetic); a wave owns whole rows.
        //      Half i of the tile = the rows wr * 64 + 32 i + (0 .. 31) of both wave rows: 64 rows x 256 floats through LDS,
        //      wave w then takes bounce rows 8 w .. 8 w + 7.
        float *bounce = (float *)lds;                       // [64][256]
        const float4 gam = ((const float4 *)cst)[lane];
        float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f}, po[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float4 xv[8], kv[8];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int lrow = wave * 8 + t, row = min(bm0 + (lrow >> 5) * 64 + i * 32 + (lrow & 31), M - 1);
                xv[t] = ((const float4 *)(p.x1 + (size_t)row * 256))[lane];
                kv[t] = ((const float4 *)(p.g + (size_t)row * 256))[lane];
            }
            __syncthreads();      // (i == 1: the rows of half 0 have been read)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    bounce[(wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh) * 256 + wc * 64 + j * 32 + r] = bf2f(f2bf(acc2[i][j][e]));
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int lrow = wave * 8 + t, row = bm0 + (lrow >> 5) * 64 + i * 32 + (lrow & 31);
                const float4 v = xv[t], k = kv[t], d = ((const float4 *)(bounce + lrow * 256))[lane];
                float s = v.x + v.y + v.z + v.w;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                const float mean = s * (1.0f / 256.0f);
                const float c0 = v.x - mean, c1 = v.y - mean, c2 = v.z - mean, c3 = v.w - mean;
                float q = c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + p.eps);
                const float h0 = c0 * rstd, h1 = c1 * rstd, h2 = c2 * rstd, h3 = c3 * rstd;
                const float a0 = d.x * gam.x, a1 = d.y * gam.y, a2 = d.z * gam.z, a3 = d.w * gam.w;
                float sa = a0 + a1 + a2 + a3, sh = a0 * h0 + a1 * h1 + a2 * h2 + a3 * h3;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sh += __shfl_xor(sh, o, 64); }
                sa *= (1.0f / 256.0f); sh *= (1.0f / 256.0f);
                const float4 rr = make_float4(rstd * (a0 - sa - h0 * sh) + k.x, rstd * (a1 - sa - h1 * sh) + k.y,
                                              rstd * (a2 - sa - h2 * sh) + k.z, rstd * (a3 - sa - h3 * sh) + k.w);
                if (row < M) {
                    ((float4 *)(p.g + (size_t)row * 256))[lane] = rr;
                    uint2 hb;
                    hb.x = f2bf2(rr.x, rr.y); hb.y = f2bf2(rr.z, rr.w);
                    ((uint2 *)(p.gout + (size_t)row * 256))[lane] = hb;
                    pg[0] += d.x * h0; pg[1] += d.y * h1; pg[2] += d.z * h2; pg[3] += d.w * h3;
                    pb[0] += d.x; pb[1] += d.y; pb[2] += d.z; pb[3] += d.w;
                    po[0] += rr.x; po[1] += rr.y; po[2] += rr.z; po[3] += rr.w;
                }
            }
        }
        __syncthreads();
        float *red = (float *)lds;      // [NW][12][64]
#pragma unroll
        for (int c = 0; c < 4; c++) { red[(wave * 12 + c) * 64 + lane] = pg[c]; red[(wave * 12 + 4 + c) * 64 + lane] = pb[c]; red[(wave * 12 + 8 + c) * 64 + lane] = po[c]; }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) t += red[(w * 12 + k) * 64 + lane];
                p.part_ln[(size_t)blockIdx.x * 768 + (k >> 2) * 256 + 4 * lane + (k & 3)] = t;      // quantity k / 4, channel 4 lane + k % 4
            }
        }
        return;
    }
    if (MODE < 2 && blockIdx.x == 0 && tid < 32) ((uint4 *)(p.xn3 + (size_t)M * 256))[tid] = make_uint4(0u, 0u, 0u, 0u);
    // ---- epilogue: x2 = acc2 + b2 + x1, norm3(x2).  Per wave 32 rows x 64 columns per trip through LDS (as ring_epilogue);
    //      a lane then holds four columns (c4) of rows 16 half + 4 q + lg; a row's 256 columns sit in the four waves of its wave row.
    // the residual rows this lane adds below (16 float4: the registers the fc1 operand held until the last chunk), requested in one
    // go -- fetched where they are used, each of the epilogue's four passes began with a round trip to L2 / HBM
    float4 res[16];
    {
        const int lg0 = lane >> 4, col0 = wc * 64 + (lane & 15) * 4;
#pragma unroll
        for (int t = 0; t < 16; t++) {      // t = (i, half, q)
            const int row = min(bm0 + wr * 64 + (t >> 3) * 32 + 16 * ((t >> 2) & 1) + 4 * (t & 3) + lg0, M - 1);
            res[t] = *(const float4 *)(p.x1 + (size_t)row * 256 + col0);
        }
    }
    float *ep = (float *)lds + wave * (32 * 68);
    float *rowsum = (float *)lds + NW * (32 * 68);      // [NW / 4 wave rows][32 rows][4 wave columns]
    const int lg = lane >> 4, c4 = (lane & 15) * 4, col = wc * 64 + c4;
    const float4 b2v = *(const float4 *)(cst + 512 + col), ga = *(const float4 *)(cst + 768 + col), be3 = *(const float4 *)(cst + 1024 + col);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ep[((e & 3) + 8 * (e >> 2) + 4 * kh) * 68 + j * 32 + r] = acc2[i][j][e];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            __syncthreads();      // the bounce tile is written (half 0) / the row sums of the half before have been read
            float4 x[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lr = 16 * half + 4 * q + lg;
                const float4 v = *(const float4 *)(ep + lr * 68 + c4);
                const float4 rs = res[i * 8 + half * 4 + q];
                x[q] = make_float4(v.x + b2v.x + rs.x, v.y + b2v.y + rs.y, v.z + b2v.z + rs.z, v.w + b2v.w + rs.w);
                const float s = row16_total(x[q].x + x[q].y + x[q].z + x[q].w);
                if ((lane & 15) == 0) rowsum[(wr * 32 + lr) * 4 + wc] = s;
            }
            __syncthreads();
            float mean[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lr = 16 * half + 4 * q + lg;
                const float4 s4 = *(const float4 *)(rowsum + (wr * 32 + lr) * 4);
                mean[q] = ((s4.x + s4.y) + (s4.z + s4.w)) * (1.0f / 256.0f);
            }
            __syncthreads();      // the sums have been read: the slots take the centred squares
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lr = 16 * half + 4 * q + lg;
                const float a = x[q].x - mean[q], b = x[q].y - mean[q], c = x[q].z - mean[q], d = x[q].w - mean[q];
                const float s = row16_total(a * a + b * b + c * c + d * d);
                if ((lane & 15) == 0) rowsum[(wr * 32 + lr) * 4 + wc] = s;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lr = 16 * half + 4 * q + lg, row = bm0 + wr * 64 + i * 32 + lr;
                const float4 s4 = *(const float4 *)(rowsum + (wr * 32 + lr) * 4);
                const float rstd = 1.0f / sqrtf(((s4.x + s4.y) + (s4.z + s4.w)) * (1.0f / 256.0f) + p.eps);
                if (row < M) {
                    const size_t o = (size_t)row * 256 + col;
                    *(float4 *)(p.x2 + o) = x[q];
                    uint2 n3;
                    n3.x = f2bf2((x[q].x - mean[q]) * rstd * ga.x + be3.x, (x[q].y - mean[q]) * rstd * ga.y + be3.y);
                    n3.y = f2bf2((x[q].z - mean[q]) * rstd * ga.z + be3.z, (x[q].w - mean[q]) * rstd * ga.w + be3.w);
                    *(uint2 *)(p.xn3 + o) = n3;
                    if (wc == 0 && (lane & 15) == 0) p.stats[row] = make_float2(mean[q], rstd);
                }
            }
        }
    }
}

template <int MODE>
static inline hipError_t launch_mlp_fused(const MlpP &p, hipStream_t s) {
    using Cfg = MfCfg<MF_NW, MODE>;
    static_assert(MODE < 2 || MF_NW == 8, "the backward's epilogue deals 64-row halves to eight waves");
    static const hipError_t attr = hipFuncSetAttribute((const void *)mlp_fused_kernel<MODE, MF_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((mlp_fused_kernel<MODE, MF_NW>), dim3((p.M + Cfg::TM - 1) / Cfg::TM), dim3(64 * MF_NW), Cfg::LDS, s, p);
    return hipGetLastError();
}

}  // namespace
