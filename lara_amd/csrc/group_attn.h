// group_attn.h -- the per-group softmax attention kernel of attention.hip, shared with the backward's
// recompute pass (encoder_bwd.hip).
#pragma once
#include "mfma_gemm.h"

namespace {

// ---- per-group softmax attention on 16x16x16 bf16 MFMA ---------------------------------------------
// One wave per unit of 4 groups (32 query tokens, 16 key/value tokens); Q [G*8,256], KV [G*4,512]
// (K in columns 0..255, V in 256..511), O [G*8,256]; all bf16.  S^T = K Q^T puts a query's four
// scores into the four accumulator registers of one lane (softmax needs no cross-lane traffic), and
// those registers ARE the A-operand fragment of the following P.V product.
// Reading the fragments straight from HBM (8-byte pieces of K and Q rows, 2-byte elements of V, O
// written two bytes at a time) ran at 2.1 TB/s; so a wave first moves its unit's rows (4 groups: 32 Q rows, 16 K|V rows; half of the 16 heads = 256 bytes per row
// at a time) into its own 16 KB of LDS with global_load_lds_dwordx4 (1 KB per instruction, contiguous
// in HBM), takes the fragments from there, writes each head's output back over the Q columns it has
// just consumed, and streams the finished 32 x 256 B block out with 16-byte stores.  Rows are 256 B,
// i.e. exactly one bank row: the 16-byte chunk index is XORed with (row & 15) on the DMA source
// address and on every LDS access, so that the 16 rows a fragment read touches use 16 different
// chunk slots.
__global__ void __launch_bounds__(256)
group_attn_kernel(const unsigned short *__restrict__ Q, const unsigned short *__restrict__ KV,
                      unsigned short *__restrict__ O, const int G) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4][16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave, g0 = unit * 4;
    if (g0 >= G) return;
    unsigned char *Qr = lds_all[wave], *Kr = Qr + 8192, *Vr = Qr + 12288;
    const int c16 = lane & 15, q4 = lane >> 4;
    const int q_rows = min(32, (G - g0) * 8), kv_rows = min(16, (G - g0) * 4);  // ragged last unit
    const int srow = lane >> 4, sphys = lane & 15;  // DMA: lane -> (row within the 4-row piece, chunk slot)
    for (int hh = 0; hh < 2; hh++) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous half's LDS reads are retired
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = 4 * i + srow, chunk = sphys ^ (row & 15);
            const unsigned short *src = Q + (size_t)(g0 * 8 + min(row, q_rows - 1)) * 256 + hh * 128 + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(Qr + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = 4 * i + srow, chunk = sphys ^ (row & 15);
            const unsigned short *src = KV + (size_t)(g0 * 4 + min(row, kv_rows - 1)) * 512 + hh * 128 + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(Kr + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 256),
                                             (__attribute__((address_space(3))) void *)(Vr + i * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // wave-private region: no barrier needed
#pragma unroll 2
        for (int hl = 0; hl < 8; hl++) {
            const int ck = 2 * hl + (q4 >> 1), sub = (q4 & 1) * 8;  // chunk / byte of a 4-element K or Q fragment
            const s16x4 kf = *(const s16x4 *)(Kr + c16 * 256 + ((ck ^ c16) << 4) + sub);
            s16x4 vf;
            const int cv = 2 * hl + (c16 >> 3), vb = (c16 & 7) * 2;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = 4 * q4 + e;
                vf[e] = *(const short *)(Vr + row * 256 + ((cv ^ row) << 4) + vb);
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {  // query groups g0+2t, g0+2t+1
                const int qrow = 16 * t + c16;
                const s16x4 qf = *(const s16x4 *)(Qr + qrow * 256 + ((ck ^ (qrow & 15)) << 4) + sub);
                f32x4 sc = {0.f, 0.f, 0.f, 0.f};
                sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf, sc, 0, 0, 0);  // S^T[key = 4 q4 + r][query = c16]
                const bool valid = q4 == 2 * t + (c16 >> 3);  // key group == query group
                const float s0 = sc[0] * 0.25f, s1 = sc[1] * 0.25f, s2 = sc[2] * 0.25f, s3 = sc[3] * 0.25f;
                const float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
                const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx), e2 = __expf(s2 - mx), e3 = __expf(s3 - mx);
                const float inv = valid ? 1.0f / (e0 + e1 + e2 + e3) : 0.f;
                s16x4 pf;
                pf[0] = (short)f2bf(e0 * inv); pf[1] = (short)f2bf(e1 * inv);
                pf[2] = (short)f2bf(e2 * inv); pf[3] = (short)f2bf(e3 * inv);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf, vf, o, 0, 0, 0);  // O[query = 4 q4 + r][d = c16]
                // the head's output replaces the head's Q columns of the same rows (already consumed)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int orow = 16 * t + 4 * q4 + rr;
                    *(unsigned short *)(Qr + orow * 256 + ((cv ^ (orow & 15)) << 4) + vb) = f2bf(o[rr]);
                }
            }
        }
        // stream the 32 x 256 B output block out, 16 bytes per lane, 1 KB per instruction
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = 4 * i + srow, chunk = sphys ^ (row & 15);
            if (row < q_rows)
                *(uint4 *)(O + (size_t)(g0 * 8 + row) * 256 + hh * 128 + chunk * 8) = *(const uint4 *)(Qr + row * 256 + sphys * 16);
        }
    }
}

}  // namespace
