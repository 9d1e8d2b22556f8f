// group_attn.h -- the fused attention step (LayerNorm, Q projection, per-group softmax attention, output projection + residual)
// of attention.hip (inference) and encoder_bwd.hip (training forward: the variant that keeps LN(x), Q and O for the backward).
#pragma once
#include "mfma_gemm.h"

namespace {

// ---- per-group softmax attention on 16x16x16 bf16 MFMA (step 3 of group_attn_fused2_kernel below) ------------------
// One wave per unit of 4 groups (32 query tokens, 16 key/value tokens); Q [32,256], K|V [16,512] (K in columns 0..255, V in
// 256..511), all bf16.  S^T = K Q^T puts a query's four scores into the four accumulator registers of one lane (softmax
// needs no cross-lane traffic), and those registers ARE the A-operand fragment of the following P.V product.
// Reading the fragments straight from HBM (8-byte pieces of K and Q rows, 2-byte elements of V, O written two bytes at a
// time) ran at 2.1 TB/s; so the operands sit in the wave's own LDS (K|V rows arrive by global_load_lds_dwordx4, 1 KB per
// instruction, half of the 16 heads = 256 bytes per row at a time; Q is produced there), the fragments are taken from
// there, and each head's output is written back over the Q columns it has just consumed.  Rows are 256 B, i.e. exactly one
// bank row: the 16-byte chunk index is XORed with (row & 15) on the DMA source address and on every LDS access, so that the
// 16 rows a fragment read touches use 16 different chunk slots.
// (Rounds 1-4 also shipped this as a stand-alone launch -- `group_attn_kernel`, Q / K|V / O through HBM, 45 us -- for the five-launch
// form of the step; round 5 moved the training forward onto the fused kernel and removed it.)

// wave-wide sum, the same value in every lane: six DPP adds (the total lands in lane 63) and one v_readlane -- the
// __shfl_xor butterfly is six LDS permutes per value, and a LayerNorm row needs two of them back to back
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_part(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_total(float x) {
    x += dpp_part<0xB1, 0xf>(x);    // quad_perm [1,0,3,2]
    x += dpp_part<0x4E, 0xf>(x);    // quad_perm [2,3,0,1]
    x += dpp_part<0x141, 0xf>(x);   // row_half_mirror
    x += dpp_part<0x140, 0xf>(x);   // row_mirror
    x += dpp_part<0x142, 0xa>(x);   // row_bcast:15
    x += dpp_part<0x143, 0xc>(x);   // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// ---- the whole attention step behind the K|V projection, one wave per 4 groups ------------------------------------
// y = x + W_o . attention(LN(x) W_q^T, K, V):  LayerNorm, Q projection, the per-group softmax attention above and
// the output projection with its residual in ONE kernel -- x is read once (and once more from L2 for the residual),
// y written once; xn, Q and O never leave the CU (round 1: ln_cast -> gemm -> group_attn -> gemm, four launches
// and 0.6 GB of bf16 intermediates through HBM per layer).
// A wave owns 32 query rows (4 groups) end to end, so there is no workgroup barrier anywhere:
//   1. LayerNorm of its rows (a row = 64 lanes x 4 channels) -> bf16 into its 16 KB of LDS, laid out as the
//      attention expects it: two half planes [16 heads / 2][32 rows][256 B], 16-byte chunks XOR-swizzled by row;
//   2. Q^T = W_q xn^T on v_mfma_f32_32x32x16_bf16 with the WEIGHT as the A operand (fragments straight from
//      L2: 16 contiguous bytes of a weight row per lane) and the rows as the B operand (16 fragments, read from
//      LDS once and held in registers): the accumulator then holds 4 consecutive features of ONE row per
//      register group -- an 8-byte LDS store puts them where the attention reads them (over the xn it replaces);
//   3. the attention described above, half of the heads at a time (K|V rows by LDS-DMA), O over Q;
//   4. y^T = W_o O^T the same way; the accumulator's 4 consecutive features of a row are 16 bytes of y: the
//      residual is added from x and the result stored.
// Each wave streams both weight matrices (256 KB) from L2 for its 32 rows: 1.07 GB of L2 reads per layer at 4
// scenes -- the price of having no barrier; HBM sees x, K|V and y only.
// (That first cut -- round 2, `group_attn_fused_kernel`, weights read straight from the row-major matrix -- lost to the four
// launches it replaced and was removed in round 5; group_attn_fused2_kernel below is the same design with packed weights.)

// A [256, 256] bf16 weight (row = output feature, K contiguous) in the order group_attn_fused2_kernel's A-operand fragments are
// read in: out[((nt * 16 + ks) * 64 + lane) * 8 + e] = W[nt * 32 + (lane & 31)][ks * 16 + 8 * (lane >> 5) + e].  Straight from
// the row-major matrix a fragment instruction touches 32 rows x 32 bytes -- 64 separate L1 accesses, 23.6 k per wave and unit,
// and the kernel sat on the L1's access rate (TCP_TOTAL_CACHE_ACCESSES 96.6 M per launch = 180 us of 222); packed, the same
// instruction reads 1 KB of consecutive bytes.
// (both weights of the step in ONE launch of 64 workgroups -- 32 per matrix; round 6: two launches per layer until then)
__global__ void __launch_bounds__(256) pack_weight_frag_kernel(const unsigned short *__restrict__ W0, unsigned short *__restrict__ out0,
                                                               const unsigned short *__restrict__ W1, unsigned short *__restrict__ out1) {
    const bool second = blockIdx.x >= 32;
    const unsigned short *W = second ? W1 : W0;
    unsigned short *out = second ? out1 : out0;
    const int t = (blockIdx.x & 31) * 256 + threadIdx.x;   // one 16-byte chunk per thread: 8192 chunks per matrix
    const int lane = t & 63, ks = (t >> 6) & 15, nt = t >> 10;
    *(uint4 *)(out + (size_t)t * 8) = *(const uint4 *)(W + (size_t)(nt * 32 + (lane & 31)) * 256 + ks * 16 + 8 * (lane >> 5));
}

// KEEP (round 5, the TRAINING forward): the three [32 x 256] bf16 blocks the backward needs -- LN(x), Q and the attention's output --
// also leave the wave's LDS for xn_out / q_out / o_out [M, 256] at the moments they are complete (two rows = 1 KB per store
// instruction, un-swizzled on the way): the training forward ran five launches until then because only they kept these rows.
template <bool KEEP>
__global__ void __launch_bounds__(64)
group_attn_fused2_kernel(const float *x /* may alias y: the block runs in place */, const float *__restrict__ gamma,
                        const float *__restrict__ beta, const float eps, const unsigned short *__restrict__ Wq /* packed */,
                        const unsigned short *__restrict__ KV, const unsigned short *__restrict__ Wo /* packed */, float *y, const int G,
                        unsigned short *__restrict__ xn_out, unsigned short *__restrict__ q_out, unsigned short *__restrict__ o_out) {
#ifndef GA_LDS_BYTES
#define GA_LDS_BYTES 24576
#endif
    // (tools/build_variant.sh -DGA_LDS_BYTES=n: occupancy experiments.  Round 6, inference instantiation, one box: 24 KB = six waves per
    //  CU 137.0 us, 32 KB = five 159.0, 40 KB = four 146.7 -- an uneven number of waves per SIMD costs more than two waves buy; eight
    //  (20 KB: K|V a quarter of the heads at a time) would be worth about what four -> six was, 7 %: not built.)
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[GA_LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x, g0 = unit * 4;
    if (g0 >= G) return;
    unsigned char *R = lds_all, *KVr = R + 16384;      // K|V of one half of the heads: K 4 KB | V 4 KB
    const int c16 = lane & 15, q4 = lane >> 4;
    const int q_rows = min(32, (G - g0) * 8), kv_rows = min(16, (G - g0) * 4);  // ragged last unit
    const int srow = lane >> 4, sphys = lane & 15;
    const size_t row0 = (size_t)g0 * 8;

    // ---- 1. LayerNorm -> bf16 rows in LDS
    {
        const float4 gm = ((const float4 *)gamma)[lane], bt = ((const float4 *)beta)[lane];
        const int hh = lane >> 5, chunk = (lane & 31) >> 1, sub = (lane & 1) * 8;
#pragma unroll 1
        for (int rb = 0; rb < 32; rb += 16) {
            float4 v[16];      // sixteen rows in flight per lane (16 KB per wave): the loads' latency is paid twice per unit, not 8 times
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = ((const float4 *)(x + (row0 + min(rb + i, q_rows - 1)) * 256))[lane];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = rb + i;
                const float mean = wave_total((v[i].x + v[i].y) + (v[i].z + v[i].w)) * (1.0f / 256.0f);
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d4 = v[i].w - mean;
                const float rstd = 1.0f / sqrtf(wave_total(a * a + b * b + c * c + d4 * d4) * (1.0f / 256.0f) + eps);
                *(uint2 *)(R + hh * 8192 + r * 256 + ((chunk ^ (r & 15)) << 4) + sub) =
                    make_uint2(f2bf2(a * rstd * gm.x + bt.x, b * rstd * gm.y + bt.y), f2bf2(c * rstd * gm.z + bt.z, d4 * rstd * gm.w + bt.w));
            }
        }
    }
    // row-operand fragments of a 32 x 256 bf16 block held in R: B[k = 16 ks + 8 (lane >> 5) + e][j = lane & 31]
    const int frow = lane & 31, fk = 8 * (lane >> 5);
    auto load_rows = [&](bf16x8 (&bf)[16]) {
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
            const int f = ks * 16 + fk;
            bf[ks] = *(const bf16x8 *)(R + (f >> 7) * 8192 + frow * 256 + ((((f & 127) >> 3) ^ (frow & 15)) << 4));
        }
    };
    // the unit's K|V rows of the first half of the heads start travelling now; they are needed after the Q projection.  (The
    // second half follows through REGISTERS while the first is being consumed -- see the attention loop: 24 KB of LDS per wave
    // = six waves per CU; with both halves resident it was 32 KB = four.)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 4 * i + srow, chunk = sphys ^ (row & 15);
        const unsigned short *src = KV + (size_t)(g0 * 4 + min(row, kv_rows - 1)) * 512 + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(KVr + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 256),
                                         (__attribute__((address_space(3))) void *)(KVr + 4096 + i * 1024), 16, 0, 0);
    }
    // the row block held in R, row-major and un-swizzled, to dst[row0 ..][256]: lane l moves chunk (l & 31) of row 2 i + (l >> 5)
    auto store_rows = [&](unsigned short *dst) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (int i = 0; i < 16; i++) {      // (four rows pairs in flight: unrolled fully, the 64 registers of loads cost the kernel its second wave per SIMD)
            const int r = 2 * i + (lane >> 5), c = lane & 31;
            const uint4 v = *(const uint4 *)(R + (c >> 4) * 8192 + r * 256 + (((c & 15) ^ (r & 15)) << 4));
            if (r < q_rows) *(uint4 *)(dst + (row0 + r) * 256 + c * 8) = v;
        }
    };
    bf16x8 bf[16], wf[16];
    load_rows(bf);
    if (KEEP) store_rows(xn_out);       // (before the Q tiles overwrite the normalised rows)
#pragma unroll
    for (int ks = 0; ks < 16; ks++) wf[ks] = *(const bf16x8 *)(Wq + ((size_t)ks * 64 + lane) * 8);
    // ---- 2. Q^T tiles: 8 tiles of 32 features; the accumulator's registers 4g..4g+3 are features
    //         n0 + 8g + 4 (lane >> 5) + (0..3) of row (lane & 31)
#pragma unroll 1
    for (int nt = 0; nt < 8; nt++) {
        // the weight fragments of the NEXT tile are requested before this tile's MFMAs (a tile is one L2 round trip)
        const unsigned short *wnext = Wq + ((size_t)min(nt + 1, 7) * 1024 + lane) * 8;
        bf16x8 wn[16];
#pragma unroll
        for (int ks = 0; ks < 16; ks++) wn[ks] = *(const bf16x8 *)(wnext + ks * 512);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], bf[ks], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int f = nt * 32 + 8 * g + 4 * (lane >> 5);
            *(uint2 *)(R + (f >> 7) * 8192 + frow * 256 + ((((f & 127) >> 3) ^ (frow & 15)) << 4) + ((f & 7) << 1)) =
                make_uint2(f2bf2(acc[4 * g], acc[4 * g + 1]), f2bf2(acc[4 * g + 2], acc[4 * g + 3]));
        }
#pragma unroll
        for (int ks = 0; ks < 16; ks++) wf[ks] = wn[ks];
    }
    if (KEEP) store_rows(q_out);        // (before the attention writes each head's output over the Q columns it consumed)
    // ---- 3. attention, half of the heads at a time (Q is in place; K|V rows come by LDS-DMA)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K|V (first half) have arrived long ago (wave-private region: no barrier)
    uint4 pk[4], pv[4];                                  // the second half's K|V rows, in the DMA's lane order, on their way
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 4 * i + srow, chunk = sphys ^ (row & 15);
        const unsigned short *src = KV + (size_t)(g0 * 4 + min(row, kv_rows - 1)) * 512 + 128 + chunk * 8;
        pk[i] = *(const uint4 *)src;
        pv[i] = *(const uint4 *)(src + 256);
    }
    for (int hh = 0; hh < 2; hh++) {
        unsigned char *Qr = R + hh * 8192, *Kr = KVr, *Vr = KVr + 4096;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous half's LDS reads are retired
        if (hh == 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                *(uint4 *)(Kr + i * 1024 + lane * 16) = pk[i];
                *(uint4 *)(Vr + i * 1024 + lane * 16) = pv[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll 2
        for (int hl = 0; hl < 8; hl++) {
            const int ck = 2 * hl + (q4 >> 1), sub = (q4 & 1) * 8;
            const s16x4 kf = *(const s16x4 *)(Kr + c16 * 256 + ((ck ^ c16) << 4) + sub);
            s16x4 vf;
            const int cv = 2 * hl + (c16 >> 3), vb = (c16 & 7) * 2;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = 4 * q4 + e;
                vf[e] = *(const short *)(Vr + row * 256 + ((cv ^ row) << 4) + vb);
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int qrow = 16 * t + c16;
                const s16x4 qf = *(const s16x4 *)(Qr + qrow * 256 + ((ck ^ (qrow & 15)) << 4) + sub);
                f32x4 sc = {0.f, 0.f, 0.f, 0.f};
                sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf, sc, 0, 0, 0);
                const bool valid = q4 == 2 * t + (c16 >> 3);
                const float s0 = sc[0] * 0.25f, s1 = sc[1] * 0.25f, s2 = sc[2] * 0.25f, s3 = sc[3] * 0.25f;
                const float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
                const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx), e2 = __expf(s2 - mx), e3 = __expf(s3 - mx);
                const float inv = valid ? 1.0f / (e0 + e1 + e2 + e3) : 0.f;
                const uint2 pp = make_uint2(f2bf2(e0 * inv, e1 * inv), f2bf2(e2 * inv, e3 * inv));
                s16x4 pf;
                __builtin_memcpy(&pf, &pp, 8);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, vf, o, 0, 0, 0);
                const unsigned o01 = f2bf2(o[0], o[1]), o23 = f2bf2(o[2], o[3]);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int orow = 16 * t + 4 * q4 + rr;
                    const unsigned w = rr < 2 ? o01 : o23;
                    *(unsigned short *)(Qr + orow * 256 + ((cv ^ (orow & 15)) << 4) + vb) = (unsigned short)((rr & 1) ? (w >> 16) : w);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KEEP) store_rows(o_out);        // (before the epilogue's fp32 tile is bounced through the row buffer)
    // ---- 4. y^T = W_o O^T, + x, stored 16 bytes per lane
    load_rows(bf);
#pragma unroll
    for (int ks = 0; ks < 16; ks++) wf[ks] = *(const bf16x8 *)(Wo + ((size_t)ks * 64 + lane) * 8);
    // Epilogue through LDS: the accumulator gives a lane 4 consecutive features of ONE row per register group, i.e. a store (and
    // the residual's load) instruction touches 32 rows x 2 pieces of 16 bytes -- 64 separate L1 accesses.  The tile (32 rows x 32
    // features, fp32) is bounced through the row buffer, which step 4 no longer needs once `bf` is loaded: lane l then owns the
    // 16-byte chunk (l & 7) of rows (l >> 3) + 8 p, p = 0..3 -- 128 contiguous bytes per row and 8 rows per instruction, for the
    // residual's loads (requested one tile ahead, like the weight fragments) and for the stores.
    float *T = (float *)R;                           // [32 rows][36 floats]: 144-byte rows spread the 16-byte writes over the banks
    const int erow = lane >> 3, ec = (lane & 7) * 4;
    float4 xr[4], xn4[4];
#pragma unroll
    for (int p4 = 0; p4 < 4; p4++) xr[p4] = *(const float4 *)(x + (row0 + min(erow + 8 * p4, q_rows - 1)) * 256 + ec);
#pragma unroll 1
    for (int nt = 0; nt < 8; nt++) {
        const unsigned short *wnext = Wo + ((size_t)min(nt + 1, 7) * 1024 + lane) * 8;
        bf16x8 wn[16];
#pragma unroll
        for (int ks = 0; ks < 16; ks++) wn[ks] = *(const bf16x8 *)(wnext + ks * 512);
#pragma unroll
        for (int p4 = 0; p4 < 4; p4++) xn4[p4] = *(const float4 *)(x + (row0 + min(erow + 8 * p4, q_rows - 1)) * 256 + min(nt + 1, 7) * 32 + ec);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], bf[ks], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(float4 *)(T + frow * 36 + 8 * g + 4 * (lane >> 5)) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
#pragma unroll
        for (int p4 = 0; p4 < 4; p4++) {
            const int row = erow + 8 * p4;
            const float4 v = *(const float4 *)(T + row * 36 + ec);
            if (row < q_rows)
                *(float4 *)(y + (row0 + row) * 256 + nt * 32 + ec) = make_float4(v.x + xr[p4].x, v.y + xr[p4].y, v.z + xr[p4].z, v.w + xr[p4].w);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ks++) wf[ks] = wn[ks];
#pragma unroll
        for (int p4 = 0; p4 < 4; p4++) xr[p4] = xn4[p4];
    }
}

}  // namespace
