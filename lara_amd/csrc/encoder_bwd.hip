// encoder_bwd.hip -- backward of LaRa's GroupAttBlock / VolTransformer tail (lightning/network.py:81-102,
// :156-163; in the reference this is torch autograd through nn.MultiheadAttention, nn.LayerNorm,
// nn.Linear, nn.Conv3d and nn.ConvTranspose3d under bf16-mixed autocast) on gfx950 matrix cores.
//
// A block keeps NOTHING from its forward except its input rows: the backward re-runs the block's
// forward into scratch (the same kernels as encoder.hip, keeping the intermediates this time) and then
// walks it in reverse.  Gradients w.r.t. activations are NT GEMMs against host-transposed weights (the
// forward's kernels, new epilogues); gradients w.r.t. weights are TN GEMMs (reduction over the token
// rows) in gemm_bf16_tn_kernel below; LayerNorm, the per-group softmax attention and the bias sums have
// their own small kernels.  All matrix products take bf16 operands and accumulate in fp32, like the
// forward; every reduction has a fixed order (partials + a second pass, no float atomics), so results
// are bit-reproducible.
#include <cstdlib>

#include "mfma_gemm.h"
#include "mlp_fused.h"
#include "group_attn.h"
#include "../../include/lara_groupattn.h"

namespace {

// ---- C[N, T*Kc] (+)= A[M, N]^T . B[M or gathered, Kc]: weight gradients -------------------------------
// Both operands are row-major with the REDUCTION index (token row m) as the slow one, the opposite of
// what an MFMA fragment wants (8 consecutive k per lane).  A lane therefore loads one dword (columns
// 2r, 2r+1) from each of the 8 rows of its k-slice and re-packs the halves: the low halves are the
// fragment of column 2r, the high halves of column 2r+1 -- two MFMA tiles (even / odd columns) per 64
// operand columns, no LDS, no barriers.  The 8 rows m0 + 8 kh + e of a k-slice are one 2x2x2 voxel group
// in the group-major token order, which also makes the gathered variant cheap: for the convolution's
// weight gradient, B rows are the (dz, dy, dx) neighbours of the A rows, looked up in a table
// nbr[tap][m] (row index, or the index of a zeroed row) that is built once per volume shape.
// Workgroup: 4 waves as 2 x 2, tile 128 (A columns) x 128 (B columns), wave tile 64 x 64; grid.y splits
// the token rows, every split writes its own partial tile (summed by accum_partials_kernel).
struct TnP {
    const unsigned short *A, *B;
    float *part;     // [splits][N][T * Kc]
    const int *nbr;  // GATHER: [T][M] byte offsets of the B rows
    int M, lda, ldb, N, Kc, T, chunk, tiles;
};
// Several products in ONE launch (gemm_tn_group): a workgroup finds its product through the cumulative slot counts (a slot =
// the eight workgroups -- one per XCD -- of one (tile, split / 8) pair).  The five weight gradients of a block's linear layers
// are 4 + 4 + 28 + 8 + 8 tiles: launched one by one each needs 128-256 splits to fill the device, and writes and re-reads
// 52-67 MB of partial tiles -- as much as its operands; together 24 splits do.
constexpr int TN_GROUP_MAX = 6;
struct TnGroup {
    TnP p[TN_GROUP_MAX];
    int first_slot[TN_GROUP_MAX + 1];
    int count;
};

// v_perm_b32 picks bytes of {S0 (bytes 4-7), S1 (bytes 0-3)}: one instruction per fragment dword
__device__ __forceinline__ bf16x8 pack_lo(const uint32_t *w) {  // the low halves of 8 dwords
    union { uint32_t u[4]; bf16x8 v; } r;
#pragma unroll
    for (int d = 0; d < 4; d++) r.u[d] = __builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x05040100u);
    return r.v;
}
__device__ __forceinline__ bf16x8 pack_hi(const uint32_t *w) {  // the high halves
    union { uint32_t u[4]; bf16x8 v; } r;
#pragma unroll
    for (int d = 0; d < 4; d++) r.u[d] = __builtin_amdgcn_perm(w[2 * d + 1], w[2 * d], 0x07060302u);
    return r.v;
}

template <bool GATHER>
__global__ void __launch_bounds__(256)
gemm_bf16_tn_kernel(const TnP p) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, kh = lane >> 5;
    // Workgroup -> (tile, split), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (each
    // with its own L2), so id L runs on XCD L % 8.  All tiles of one split read the SAME token rows (for the
    // convolution: the same A rows under every tap, and B rows from the same 4x4x4 voxel neighbourhoods), so a
    // split's tiles are dealt to one XCD, where they run side by side and share those rows through its L2
    // instead of each pulling them over the fabric (27 x 4 tiles: 7.2 GB of operand reads per launch).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // (the integer division runs on the vector unit: tell the compiler its result is wave-uniform)
    const int sq = __builtin_amdgcn_readfirstlane(slot / p.tiles);
    const int tile = slot - sq * p.tiles, split = sq * 8 + xcd;
    const int ctiles = (p.Kc + 127) / 128 * p.T;  // column tiles (a tile never straddles two taps)
    const int ct = tile % ctiles, nt = tile / ctiles;
    const int tpt = (p.Kc + 127) / 128;  // tiles per tap
    const int tap = ct / tpt, kc0 = (ct - tap * tpt) * 128;
    const int nbase = nt * 128 + (wave >> 1) * 64, kbase = kc0 + (wave & 1) * 64;
    const int an = min(nbase + 2 * r, p.N - 2), bk = min(kbase + 2 * r, p.Kc - 2);
    const int m_start = split * p.chunk, m_end = min(p.M, m_start + p.chunk);
    // operands are < 4 GB: a uniform base + 32-bit byte offsets per lane (24-bit multiplies for the gathered
    // rows, running adds for the dense ones) -- 64-bit address arithmetic was 2/3 of this loop's VALU work
    const char *Ab = (const char *)p.A, *Bb = (const char *)p.B;
    const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u, bcol = (uint32_t)bk * 2u;
    const int *nb = GATHER ? p.nbr + (size_t)tap * p.M : nullptr;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // per-lane byte offsets of the 8 rows of its k-slice inside a 16-row slice; the slice itself advances
    // through a uniform (scalar) base pointer, so the loop has no per-lane address arithmetic for dense operands
    uint32_t offa[8], offb[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        offa[e] = (uint32_t)(8 * kh + e) * lda2 + (uint32_t)an * 2u;
        offb[e] = (uint32_t)(8 * kh + e) * ldb2 + bcol;
    }
    uint32_t a0[8], b0[8], a1[8], b1[8], a2[8], b2[8], a3[8], b3[8];
    // Gathered rows: the 16 table entries of a slice (byte offsets of B rows) are wave-uniform (two groups of 8:
    // k-slices 0 and 1), so they come through the scalar cache (s_load, its own counter: waiting for them does
    // not drain the vector loads in flight) one slice ahead of their use.
    int tn[16];
    auto table = [&](const int m0) {
        const int *t = nb + min(m0, p.M - 16);
#pragma unroll
        for (int e = 0; e < 16; e++) tn[e] = t[e];
    };
    if (GATHER) table(m_start);
    // buffer loads: base in a scalar resource descriptor, per-lane offset register that never changes, and the
    // slice offset in the scalar offset operand
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)Ab, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)Bb, 0, -1, 0x00020000);
    auto load = [&](const int m0, uint32_t *a, uint32_t *b) {  // slices in order (the table runs one ahead)
        const int sa = (int)((uint32_t)m0 * lda2), sb = GATHER ? 0 : (int)((uint32_t)m0 * ldb2);
        if (GATHER) {
#pragma unroll
            for (int e = 0; e < 8; e++) offb[e] = (uint32_t)(kh ? tn[8 + e] : tn[e]) + bcol;
            table(m0 + 16);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            a[e] = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)offa[e], sa, 0);
            b[e] = __builtin_amdgcn_raw_buffer_load_b32(rb, (int)offb[e], sb, 0);
        }
    };
    auto mma = [&](const uint32_t *a, const uint32_t *b) {
        const bf16x8 fa0 = pack_lo(a), fa1 = pack_hi(a), fb0 = pack_lo(b), fb1 = pack_hi(b);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
    };
    // Four register sets: three 16-row slices are in flight while one is multiplied.  With ~3 waves per SIMD
    // (64 accumulator + ~100 vector registers) one slice of look-ahead left the matrix cores waiting on HBM
    // latency most of the time.
    const int steps = max(0, (m_end - m_start + 15) >> 4), full = steps & ~3;  // (a split may start beyond M)
    if (full > 0) {
        load(m_start, a0, b0);
        load(m_start + 16, a1, b1);
        load(m_start + 32, a2, b2);
    }
    for (int i = 0; i < full; i += 4) {
        const bool more = i + 4 < full;
        load(m_start + 16 * (i + 3), a3, b3);
        mma(a0, b0);
        if (more) load(m_start + 16 * (i + 4), a0, b0);
        mma(a1, b1);
        if (more) load(m_start + 16 * (i + 5), a1, b1);
        mma(a2, b2);
        if (more) load(m_start + 16 * (i + 6), a2, b2);
        mma(a3, b3);
    }
    for (int i = full; i < steps; i++) {  // at most three left-over slices (the last split of a ragged M)
        load(m_start + 16 * i, a0, b0);
        mma(a0, b0);
    }
    // accumulator (ta, tb)[reg]: A column nbase + 2 i + ta with i = (reg&3) + 8 (reg>>2) + 4 kh, B column
    // kbase + 2 (lane&31) + tb: the two tb's of a lane are neighbours -> float2 stores
    const int ldc = p.T * p.Kc;
    float *out = p.part + (size_t)split * p.N * ldc + (size_t)tap * p.Kc;
    const int col = kbase + 2 * r;
    if (col < p.Kc) {
#pragma unroll
        for (int ta = 0; ta < 2; ta++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int n = nbase + 2 * ((e & 3) + 8 * (e >> 2) + 4 * kh) + ta;
                if (n < p.N) *(float2 *)(out + (size_t)n * ldc + col) = make_float2(acc[ta][0][e], acc[ta][1][e]);
            }
    }
}

// ---- the same product, operands staged through LDS --------------------------------------------------------
// The direct kernel above spends its time in the texture path: sixteen 4-byte loads per lane and 16-row slice,
// every operand column fetched by two waves (rocprof: 57 M vector-memory instructions and 228 M L1 accesses per
// convolution weight gradient, matrix cores 18 % busy).  Here a stage of 32 token rows x 128 columns of each
// operand (2 x 8 KB) travels global -> LDS with global_load_lds_dwordx4 (row-major image, four 1 KB pieces per
// wave and stage), and the k-strided fragments come out of LDS with gfx950's transposing read
// ds_read_b64_tr_b16: the 16 lanes of a group pass the addresses of a [4 rows][16 columns] block (lane l: row
// l >> 2, columns 4 (l & 3) ..+3) and lane l receives column l of the four rows -- half of an MFMA fragment
// (measured with tools/ubench/tr_read.hip).  LDS rows are 256 B = one bank row, so the 16-byte chunk index is
// XORed with (row & 3) << 2 on the DMA source address and on the reads: the 4 rows x 64 B that 32 lanes touch
// then cover all 64 banks once.  Two stages of 16 KB (the next one is in flight while this one is multiplied; 32 KB
// per workgroup keeps four workgroups per CU, so the 108 x 8 tiles of the convolution run as one round), one
// barrier per stage (8 MFMAs per wave).
// Needs M % 32 == 0 and 16-byte aligned rows (lda, ldb, N, Kc multiples of 8); the host falls back to the
// direct kernel otherwise.  Same workgroup -> (tile, split) mapping, same partial-tile output.
// (Round 5 measured deeper rings under a counted vmcnt on the same box -- four 16-row stages in the same 32 KB: 754 us, the barrier
// per four MFMAs costs more than the prefetch returns; three / four 32-row stages at three / two workgroups per CU: 897 / 843 us, the
// 864 workgroups no longer fit the device at once -- against 498-517 us for this form: two stages, four workgroups per CU.)
constexpr int TL_ROWS = 32, TL_STAGES = 2, TL_TILE = TL_ROWS * 256, TL_STAGE = 2 * TL_TILE;
constexpr int TL_WROWS = TL_ROWS / 4, TL_PIECES = TL_WROWS / 4;      // rows a wave moves per stage, as 4-row pieces

template <bool GATHER>
__global__ void __launch_bounds__(256)
gemm_bf16_tn_lds_kernel(const TnGroup grp) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[TL_STAGES * TL_STAGE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7;
    int slot = blockIdx.x >> 3, which = 0;
    while (which + 1 < grp.count && slot >= grp.first_slot[which + 1]) which++;
    slot -= grp.first_slot[which];
    const TnP &p = grp.p[which];
    const int sq = __builtin_amdgcn_readfirstlane(slot / p.tiles);
    const int tile = slot - sq * p.tiles, split = sq * 8 + xcd;
    const int tpt = (p.Kc + 127) / 128, ctiles = tpt * p.T;
    const int ct = tile % ctiles, nt = tile / ctiles;
    const int tap = ct / tpt, kc0 = (ct - tap * tpt) * 128;
    const int m_start = split * p.chunk, m_end = min(p.M, m_start + p.chunk);
    const int nst = max(0, (m_end - m_start) / TL_ROWS);  // chunk and M are multiples of 32
    const char *Ab = (const char *)p.A, *Bb = (const char *)p.B;
    const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
    const int *nb = GATHER ? p.nbr + (size_t)tap * p.M : nullptr;

    // DMA: wave w moves rows 8w .. 8w+7 of a stage, as two 4-row pieces per operand; lane L lands in row
    // 4 piece + (L >> 4), physical chunk L & 15, and fetches the logical chunk (L & 15) ^ ((row & 3) << 2)
    const int drow = lane >> 4;  // row within a piece == row & 3
    const int dchunk = (lane & 15) ^ (drow << 2);
    const uint32_t acol2 = (uint32_t)min(nt * 128 + dchunk * 8, p.N - 8) * 2u;
    const uint32_t bcol2 = (uint32_t)min(kc0 + dchunk * 8, p.Kc - 8) * 2u;
    int tn[TL_WROWS];  // GATHER: byte offsets of this wave's B rows of the next stage to be issued
    auto table = [&](const int st) {
        // (a provably wave-uniform address in the constant address space: the table arrives by s_load instead of a vector load whose
        // vmcnt(0) the compiler put in front of every use -- round 5: the convolution's weight gradient 537-580 -> 498-517 us)
        const int *t = nb + __builtin_amdgcn_readfirstlane(min(m_start + st * TL_ROWS, p.M - TL_ROWS) + TL_WROWS * wave);
        const __attribute__((address_space(4))) int *t4 = (const __attribute__((address_space(4))) int *)t;   // read-only table: constant address space -> s_load
#pragma unroll
        for (int e = 0; e < TL_WROWS; e++) tn[e] = t4[e];
    };
    auto issue = [&](const int st) {  // this wave's four pieces of stage st
        unsigned char *buf = lds + (st % TL_STAGES) * TL_STAGE;
        const uint32_t row0 = (uint32_t)(m_start + st * TL_ROWS + TL_WROWS * wave);
#pragma unroll
        for (int q = 0; q < TL_PIECES; q++) {
            const uint32_t arow = row0 + 4 * q + drow;
            uint32_t boff;
            if (GATHER) {
                const int t0 = tn[4 * q], t1 = tn[4 * q + 1], t2 = tn[4 * q + 2], t3 = tn[4 * q + 3];
                boff = (uint32_t)(drow == 0 ? t0 : drow == 1 ? t1 : drow == 2 ? t2 : t3);
            } else {
                boff = arow * ldb2;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Ab + (arow * lda2 + acol2)),
                                             (__attribute__((address_space(3))) void *)(buf + (TL_PIECES * wave + q) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Bb + (boff + bcol2)),
                                             (__attribute__((address_space(3))) void *)(buf + TL_TILE + (TL_PIECES * wave + q) * 1024), 16, 0, 0);
        }
    };

    // fragment addresses inside a stage: k-step s (rows 16 s ..), fragment half q (rows +4 q), 32-column tile t
    const int g = lane >> 4, l = lane & 15, kh = g >> 1;
    const int frow = 8 * kh + (l >> 2);                       // + 16 s + 4 q
    const int fcolA = (wave >> 1) * 64 + 16 * (g & 1) + 4 * (l & 3);  // + 32 t
    const int fcolB = (wave & 1) * 64 + 16 * (g & 1) + 4 * (l & 3);
    auto faddr = [&](const int row, const int col) { return row * 256 + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + (col & 7) * 2; };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    if (GATHER && nst > 0) table(0);
    if (nst > 0) { issue(0); if (GATHER) table(1); }
    for (int st = 0; st < nst; st++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of stage st have landed
        __builtin_amdgcn_s_barrier();  // everybody's pieces visible; everybody is done reading stage st-1
        if (st + 1 < nst) {  // refill the other buffer while this one is multiplied
            issue(st + 1);
            if (GATHER) table(st + 2);
        }
        const unsigned char *buf = lds + (st % TL_STAGES) * TL_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < TL_ROWS / 16; s2++) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                union { bf16x4 h[2]; bf16x8 v; } ua, ub;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int row = 16 * s2 + 4 * q + frow;
                    ua.h[q] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4 *)(buf + faddr(row, fcolA + 32 * t)));
                    ub.h[q] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4 *)(buf + TL_TILE + faddr(row, fcolB + 32 * t)));
                }
                fa[t] = ua.v; fb[t] = ub.v;
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment reads (and the table) done before the next barrier
    }
    // accumulator (ta, tb)[reg]: A column nbase + 32 ta + (reg&3) + 8 (reg>>2) + 4 (lane>>5), B column kbase + 32 tb + (lane&31)
    const int ldc = p.T * p.Kc;
    float *out = p.part + (size_t)split * p.N * ldc + (size_t)tap * p.Kc;
    const int nbase = nt * 128 + (wave >> 1) * 64, kbase = kc0 + (wave & 1) * 64;
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < 2; tb++) {
            const int col = kbase + 32 * tb + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int n = nbase + 32 * ta + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n < p.N && col < p.Kc) out[(size_t)n * ldc + col] = acc[ta][tb][e];
            }
        }
}

// ---- the convolution's weight gradient on a 256 x 256 tile ---------------------------------------------
// gemm_bf16_tn_lds_kernel<true> for N = Kc = 256, restructured like gemm_ring2_kernel (mfma_gemm.h): one workgroup of 8 waves
// (2 x 4, wave tile 128 x 64) owns ALL 256 x 256 outputs of one (tap, token split), so a stage of 32 token rows (2 x 16 KB) feeds
// 16 MFMAs per wave instead of 8 -- half the LDS-DMA pieces, fragment reads and barriers per MFMA -- and travels through a ring of
// four stages (128 KB, one workgroup per CU): stage st+3 is issued during stage st, a piece at a time between MFMAs (wave row 0
// in the first K step, row 1 in the second), each wave waits for its pieces of stage st+1 with a counted vmcnt, one barrier per
// stage, fragment reads (ds_read_b64_tr_b16, the same [row][512 B] image with the 16-byte chunk index XORed by (row & 3) << 2)
// dealt over the MFMAs of the K step before.  The gathered rows' byte offsets (four per wave and stage) come through the scalar
// cache one stage ahead.  T x splits workgroups, dealt so that one XCD gets consecutive (split, tap) pairs: the 27 taps of a split
// read the same rows of A and overlapping rows of B through that XCD's L2.
struct TnRingP {
    const unsigned short *A, *B;
    float *part;     // [splits][N][T * Kc]
    const int *nbr;  // gathered B rows: [T][M] byte offsets (then N = Kc = 256), or null
    int M, lda, ldb, T, ntile, ktile, Kc;   // N = 256 ntile, ktile = ceil(Kc / 256) (Kc % 8 == 0; gathered: Kc = 256); T taps (1 without nbr)
    int chunk, splits;                  // chunk: token rows per split, a multiple of 128
};
// several products in one launch (the weight gradients of a block's linear layers): a workgroup finds its product through the
// cumulative workgroup counts
struct TnRingGroup {
    TnRingP p[TN_GROUP_MAX];
    int first_wg[TN_GROUP_MAX + 1];
    int count;
};
constexpr int TR_TILE = 32 * 512, TR_STAGE = 2 * TR_TILE, TR_RING = 4;

template <bool GATHER>
__global__ void __launch_bounds__(512)
gemm_tn_ring_kernel(const TnRingGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // TR_RING * TR_STAGE bytes
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int id = blockIdx.x, which = 0;
    while (which + 1 < grp.count && id >= grp.first_wg[which + 1]) which++;
    id -= grp.first_wg[which];
    const TnRingP &p = grp.p[which];
    // workgroup -> (split, tile): ids go round-robin over the 8 XCDs; an XCD takes a run of consecutive pairs (the tiles of a
    // split read the same rows of A, and the same or overlapping rows of B, through that XCD's L2)
    const int tiles = p.T * p.ntile * p.ktile, total = tiles * p.splits, xcd = id & 7, q8 = total >> 3, rem = total & 7;
    const int pi = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (id >> 3);
    const int split = pi / tiles, tile = pi - split * tiles;
    const int tap = GATHER ? tile : 0, nt = GATHER ? 0 : tile / p.ktile, kt = GATHER ? 0 : tile - nt * p.ktile;
    const int m_start = split * p.chunk, m_end = min(p.M, m_start + p.chunk);
    const int groups = max(0, (m_end - m_start) >> 7);   // four stages each
    const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;

    // staging: wave w moves rows 4w .. 4w+3 of a stage of each operand, as two 2-row pieces; lane L lands in row 4w + 2q + (L >> 5),
    // physical chunk L & 31, and fetches the logical chunk (L & 31) ^ ((row & 3) << 2)
    const int hrow = lane >> 5;
    uint32_t aoffq[2], boffq[2];   // (gathered: boffq is the byte offset inside the row only)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int rr = 2 * q + hrow;   // == row & 3
        const uint32_t lc = (uint32_t)((lane & 31) ^ (rr << 2));
        aoffq[q] = (uint32_t)(m_start + 4 * wave + rr) * lda2 + (uint32_t)nt * 512u + lc * 16u;
        // (a ragged last column tile fetches in-bounds columns again; its surplus accumulator columns are not stored)
        boffq[q] = GATHER ? lc * 16u : (uint32_t)(m_start + 4 * wave + rr) * ldb2 + (uint32_t)min(kt * 256 + (int)lc * 8, p.Kc - 8) * 2u;
    }
    const char *pa = (const char *)p.A, *pb = (const char *)p.B;   // + the stage being issued (pb: plain rows only)
    const __attribute__((address_space(4))) int *pt = (const __attribute__((address_space(4))) int *)(
        GATHER ? p.nbr + (size_t)tap * p.M + m_start + 4 * wave : nullptr);   // + the stage whose offsets are loaded
    int tn[2][4];   // [stage parity]: byte offsets of this wave's four B rows
    const uint32_t mine = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)ring + wave * 2048;
#define TR_DMA(base, voff, lds)                                                                      \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                      \
                 :: "s"((uint32_t)(lds)), "v"(voff), "s"(base) : "memory")
#define TR_DMA_A(slot, q) TR_DMA(pa, aoffq[q], mine + (slot) * TR_STAGE + (q) * 1024)
#define TR_DMA_B(slot, q, par)                                                                       \
    if (GATHER) {                                                                                    \
        const uint32_t bo_ = (uint32_t)(hrow ? tn[par][2 * (q) + 1] : tn[par][2 * (q)]) + boffq[q];  \
        TR_DMA(pb, bo_, mine + (slot) * TR_STAGE + TR_TILE + (q) * 1024);                             \
    } else {                                                                                         \
        TR_DMA(pb, boffq[q], mine + (slot) * TR_STAGE + TR_TILE + (q) * 1024);                        \
    }
#define TR_NEXT_STAGE { pa += 32 * lda2; if (!GATHER) pb += 32 * ldb2; }
#define TR_TABLE(par)                                                                                \
    if (GATHER) {                                                                                    \
        tn[par][0] = pt[0]; tn[par][1] = pt[1]; tn[par][2] = pt[2]; tn[par][3] = pt[3];              \
        pt += 32;                                                                                    \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // fragment addresses inside a stage (gemm_bf16_tn_lds_kernel's, with 512-byte rows): K step s covers rows 16 s .., a
    // fragment is two transposing reads (rows + 0 and + 4); ring slots 2 and 3 get base registers of their own
    const int g16 = lane >> 4, l = lane & 15, kh = g16 >> 1;
    const int frow = 8 * kh + (l >> 2);
    auto faddr = [&](const int col) { return frow * 512 + ((((col >> 3) ^ ((frow & 3) << 2))) << 4) + (col & 7) * 2; };
    int aaddr[4][2], baddr[2][2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        aaddr[i][0] = faddr(wr * 128 + 32 * i + 16 * (g16 & 1) + 4 * (l & 3));
        aaddr[i][1] = aaddr[i][0] + 2 * TR_STAGE;
        asm volatile("" : "+v"(aaddr[i][1]));
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        baddr[j][0] = TR_TILE + faddr(wc * 64 + 32 * j + 16 * (g16 & 1) + 4 * (l & 3));
        baddr[j][1] = baddr[j][0] + 2 * TR_STAGE;
        asm volatile("" : "+v"(baddr[j][1]));
    }
    bf16x8 fa[2][4], fb[2][2];
#define TR_SB __builtin_amdgcn_sched_barrier(0)
#define TR_M(set, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0)
#define TR_RD(dst, addr)                                                                                                        \
    {                                                                                                                           \
        union { bf16x4 h[2]; bf16x8 v; } u_;                                                                                    \
        u_.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(ring + (addr)));          \
        u_.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(ring + (addr) + 4 * 512)); \
        dst = u_.v;                                                                                                             \
    }
#define TR_RA(set, i, slot, step) TR_RD(fa[set][i], ((slot) & 1) * TR_STAGE + (step) * (16 * 512) + aaddr[i][(slot) >> 1])
#define TR_RB(set, j, slot, step) TR_RD(fb[set][j], ((slot) & 1) * TR_STAGE + (step) * (16 * 512) + baddr[j][(slot) >> 1])
    // one K step: the eight MFMAs of fragment set USE, between them the fragment reads of the step after it (set USE ^ 1, ring slot
    // `slot`, K step `step`) and, if ISSUE, the four pieces of the stage three further on (ring slot `is`, offsets of parity `par`);
    // TAB: the offsets of the stage after that one are requested behind the seventh MFMA (of the SECOND K step: the scalar load
    // shares lgkmcnt with the fragment reads, and the first K step ends in the drain before the barrier)
#define TR_HALF(USE, READ, slot, step, ISSUE, is, par, TAB)               \
    TR_M(USE, 0, 0); TR_SB;                                              \
    if (READ) TR_RA(USE ^ 1, 0, slot, step);                             \
    TR_SB; TR_M(USE, 0, 1); TR_SB;                                       \
    if (READ) TR_RB(USE ^ 1, 0, slot, step);                             \
    if (ISSUE) TR_DMA_A(is, 0);                                          \
    TR_SB; TR_M(USE, 1, 0); TR_SB;                                       \
    if (READ) TR_RB(USE ^ 1, 1, slot, step);                             \
    if (ISSUE) TR_DMA_A(is, 1);                                          \
    TR_SB; TR_M(USE, 1, 1); TR_SB;                                       \
    if (READ) TR_RA(USE ^ 1, 1, slot, step);                             \
    if (ISSUE) { TR_DMA_B(is, 0, par) }                                  \
    TR_SB; TR_M(USE, 2, 0); TR_SB;                                       \
    if (READ) TR_RA(USE ^ 1, 2, slot, step);                             \
    if (ISSUE) { TR_DMA_B(is, 1, par) TR_NEXT_STAGE }                    \
    TR_SB; TR_M(USE, 2, 1); TR_SB;                                       \
    if (READ) TR_RA(USE ^ 1, 3, slot, step);                             \
    TR_SB; TR_M(USE, 3, 0); TR_SB;                                       \
    if (TAB) TR_TABLE(par ^ 1);                                          \
    TR_SB; TR_M(USE, 3, 1); TR_SB;
    // stage t of a group sits in ring slot t; I1 / I2: the first / second K step issues stage t + 3 (slot (t + 3) & 3, offsets of
    // parity (t + 3) & 1); TAB: request the offsets of stage t + 4.  VM = this wave's younger pieces allowed in flight when its
    // pieces of the NEXT stage must have landed (-1: there is no next stage)
#define TR_TILE_(t, I1, I2, TAB, VM)                                                            \
    TR_HALF(0, true, t, 1, I1, ((t) + 3) & 3, ((t) + 3) & 1, false)                             \
    if ((VM) >= 0) {                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
        if ((VM) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                         \
        else if ((VM) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_s_barrier();                                                           \
    }                                                                                           \
    TR_SB;                                                                                      \
    TR_HALF(1, (VM) >= 0, ((t) + 1) & 3, 0, I2, ((t) + 3) & 3, ((t) + 3) & 1, TAB)
#define TR_LOOP(I1, I2, VMS)                                      \
    for (int g = 0; g + 1 < groups; g++) {                        \
        TR_TILE_(0, I1, I2, true, VMS)                            \
        TR_TILE_(1, I1, I2, true, VMS)                            \
        TR_TILE_(2, I1, I2, true, VMS)                            \
        TR_TILE_(3, I1, I2, true, VMS)                            \
    }                                                             \
    TR_TILE_(0, I1, I2, false, VMS)                               \
    TR_TILE_(1, false, false, false, 4)                           \
    TR_TILE_(2, false, false, false, 0)                           \
    TR_TILE_(3, false, false, false, -1)

    if (groups > 0) {
        // prologue: stages 0, 1, 2 (and the offsets of stage 3)
        TR_TABLE(0)
        TR_DMA_A(0, 0); TR_DMA_A(0, 1); TR_DMA_B(0, 0, 0) TR_DMA_B(0, 1, 0) TR_NEXT_STAGE
        TR_TABLE(1)
        TR_DMA_A(1, 0); TR_DMA_A(1, 1); TR_DMA_B(1, 0, 1) TR_DMA_B(1, 1, 1) TR_NEXT_STAGE
        TR_TABLE(0)
        TR_DMA_A(2, 0); TR_DMA_A(2, 1); TR_DMA_B(2, 0, 0) TR_DMA_B(2, 1, 0) TR_NEXT_STAGE
        TR_TABLE(1)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        TR_RA(0, 0, 0, 0); TR_RB(0, 0, 0, 0); TR_RB(0, 1, 0, 0); TR_RA(0, 1, 0, 0); TR_RA(0, 2, 0, 0); TR_RA(0, 3, 0, 0);
        TR_SB;
        if (wr == 0) {
            TR_LOOP(true, false, 8)
        } else {
            TR_LOOP(false, true, 4)
        }
    }
#undef TR_LOOP
#undef TR_TILE_
#undef TR_HALF
#undef TR_RA
#undef TR_RB
#undef TR_RD
#undef TR_M
#undef TR_SB
#undef TR_TABLE
#undef TR_NEXT_STAGE
#undef TR_DMA_B
#undef TR_DMA_A
#undef TR_DMA
    // accumulator (i, j)[reg]: A column 256 nt + wr*128 + 32 i + (reg&3) + 8 (reg>>2) + 4 (lane>>5), B column 256 kt + wc*64 + 32 j + (lane&31)
    const int ldc = p.T * p.Kc;
    float *out = p.part + ((size_t)split * p.ntile + nt) * 256 * ldc + (size_t)tap * p.Kc + kt * 256;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = wc * 64 + 32 * j + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int n = wr * 128 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (GATHER || kt * 256 + col < p.Kc) out[(size_t)n * ldc + col] = acc[i][j][e];
            }
        }
}

// dst[i] += sum over parts of part[s * stride + i]; 64 elements per workgroup, blockDim / 64 slices of the
// parts (256 threads when there are many elements and few parts, 1024 for the opposite)
__global__ void __launch_bounds__(1024)
accum_partials_kernel(float *__restrict__ dst, const float *__restrict__ part, const int n, const int parts,
                      const size_t stride) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6, nsl = blockDim.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < n) {
        // sixteen loads in flight per thread: a slice walks up to 128 parts, and one load + add per part at a time is pure latency
        int q = sl;
        for (; q + 15 * nsl < parts; q += 16 * nsl) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = part[(size_t)(q + u * nsl) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] += v[u + 8];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] += v[u + 4];
            s += (v[0] + v[2]) + (v[1] + v[3]);
        }
        for (; q < parts; q += nsl) s += part[(size_t)q * stride + c];
    }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && c < n) {
        float t = red[0][lane];
        for (int k = 1; k < nsl; k++) t += red[k][lane];
        dst[c] += t;
    }
}

// the same for the weight-gradient partial tiles (many elements, tens of parts): four consecutive elements per
// thread (16-byte loads), 256 elements per workgroup, 4 slices of the parts; n % 4 == 0
__global__ void __launch_bounds__(256)
accum_partials4_kernel(float *__restrict__ dst, const float *__restrict__ part, const int n, const int parts,
                       const size_t stride) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n)
        for (int q = sl; q < parts; q += 4) {
            const float4 v = *(const float4 *)(part + (size_t)q * stride + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && c < n) {
        float4 d = *(float4 *)(dst + c);
#pragma unroll
        for (int k = 0; k < 4; k++) { const float4 v = red[k][lane]; d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
        *(float4 *)(dst + c) = d;
    }
}

// the same for the products of one gemm_tn_group launch: workgroup -> product through the cumulative workgroup counts
struct AccGroup {
    float *dst[TN_GROUP_MAX];
    const float *part[TN_GROUP_MAX];
    int n[TN_GROUP_MAX], parts[TN_GROUP_MAX], first_wg[TN_GROUP_MAX + 1];
    int count;
};
__global__ void __launch_bounds__(256)
accum_partials4_group_kernel(const AccGroup a) {
    __shared__ float4 red[4][64];
    int wg = blockIdx.x, which = 0;
    while (which + 1 < a.count && wg >= a.first_wg[which + 1]) which++;
    wg -= a.first_wg[which];
    float *dst = a.dst[which];
    const float *part = a.part[which];
    const int n = a.n[which], parts = a.parts[which];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = (wg * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n)
        for (int q = sl; q < parts; q += 4) {
            const float4 v = *(const float4 *)(part + (size_t)q * n + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && c < n) {
        float4 d = *(float4 *)(dst + c);
#pragma unroll
        for (int k = 0; k < 4; k++) { const float4 v = red[k][lane]; d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
        *(float4 *)(dst + c) = d;
    }
}

// few elements, hundreds of parts (the coarse decoder's 80 x 80 weight gradients: 25 workgroups of the kernel above would
// each walk 1024 parts, 256 dependent loads per thread): 16 slices of the parts per workgroup instead of 4
__global__ void __launch_bounds__(1024)
accum_partials4_wide_kernel(float *__restrict__ dst, const float *__restrict__ part, const int n, const int parts,
                            const size_t stride) {
    __shared__ float4 red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n)
        for (int q = sl; q < parts; q += 16) {
            const float4 v = *(const float4 *)(part + (size_t)q * stride + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && c < n) {
        float4 d = *(float4 *)(dst + c);
#pragma unroll
        for (int k = 0; k < 16; k++) { const float4 v = red[k][lane]; d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
        *(float4 *)(dst + c) = d;
    }
}

// the three column sums of ln_bwd_kernel's partials in one launch: blockIdx.y = quantity (its dst may be null)
__global__ void __launch_bounds__(1024)
accum_ln_partials_kernel(float *d0, float *d1, float *d2, const float *__restrict__ part, const int parts) {
    __shared__ float red[16][64];
    float *dst = blockIdx.y == 0 ? d0 : blockIdx.y == 1 ? d1 : d2;
    if (!dst) return;
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;      // (sixteen loads in flight per thread, as in accum_partials_kernel)
    int q = sl;
    for (; q + 240 < parts; q += 256) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = part[(size_t)(q + 16 * u) * 768 + blockIdx.y * 256 + c];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] += v[u + 8];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] += v[u + 4];
        s += (v[0] + v[2]) + (v[1] + v[3]);
    }
    for (; q < parts; q += 16) s += part[(size_t)q * 768 + blockIdx.y * 256 + c];
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0) {
        float t = red[0][lane];
        for (int k = 1; k < 16; k++) t += red[k][lane];
        dst[c] += t;
    }
}

// Several of those column reductions in one launch (a block's three LayerNorm backward passes and its b1 gradient leave eight of
// them; one by one they are eight-to-twelve-workgroup kernels that run alone on the device, 17 us each): dst_k[c] += sum over
// q < parts_k of part_k[q * stride_k + c], 64 columns per workgroup, 16 slices of the parts, sixteen loads in flight per thread.
constexpr int RED_GROUP_MAX = 8;
struct RedGroup {
    float *dst[RED_GROUP_MAX];
    const float *part[RED_GROUP_MAX];
    int n[RED_GROUP_MAX], parts[RED_GROUP_MAX], stride[RED_GROUP_MAX], first_wg[RED_GROUP_MAX + 1];
    int count;
};
__global__ void __launch_bounds__(1024)
reduce_group_kernel(const RedGroup g) {
    __shared__ float red[16][64];
    int wg = blockIdx.x, which = 0;
    while (which + 1 < g.count && wg >= g.first_wg[which + 1]) which++;
    wg -= g.first_wg[which];
    const float *part = g.part[which];
    const int n = g.n[which], parts = g.parts[which];
    const size_t stride = (size_t)g.stride[which];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = wg * 64 + lane;
    float s = 0.f;
    if (c < n) {
        int q = sl;
        for (; q + 240 < parts; q += 256) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = part[(size_t)(q + 16 * u) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] += v[u + 8];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] += v[u + 4];
            s += (v[0] + v[2]) + (v[1] + v[3]);
        }
        for (; q < parts; q += 16) s += part[(size_t)q * stride + c];
    }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && c < n) {
        float t = red[0][lane];
        for (int k = 1; k < 16; k++) t += red[k][lane];
        g.dst[which][c] += t;
    }
}

// nbr[tap][m] = BYTE offset (rows of `row_bytes`) of the token row of voxel(m) + (dz, dy, dx), or of row
// `zero_row` outside the volume
__global__ void __launch_bounds__(256)
neighbour_table_kernel(int *__restrict__ nbr, const int M, const int R, const int zero_row, const int row_bytes) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    int b, d, h, w;
    token_to_voxel(m, R, b, d, h, w);
    for (int tap = 0; tap < 27; tap++) {
        const int nd = d + tap / 9 - 1, nh = h + (tap / 3) % 3 - 1, nw = w + tap % 3 - 1;
        const bool in = (unsigned)nd < (unsigned)R && (unsigned)nh < (unsigned)R && (unsigned)nw < (unsigned)R;
        nbr[(size_t)tap * M + m] = (in ? voxel_to_token(b, nd, nh, nw, R) : zero_row) * row_bytes;
    }
}

// ---- LayerNorm(256) backward: one wave per row, 32 rows per wave, 128 rows per workgroup --------------
// dx = rstd (a - mean(a) - xhat mean(a xhat)), a = dy gamma;  out = dx (+ skip) as fp32 and, optionally,
// bf16; per-workgroup partial column sums of dy xhat (dgamma), dy (dbeta) and out (the bias gradient of
// the linear layer that produced the LayerNorm's input) go to part[block][3][256].
// dx_out may alias dy or skip (same element only): a row's loads are issued before the previous row's stores.
// DY_BF16: dy is the bf16 output of the dX product in front (what the reference's LayerNorm backward receives under
// autocast: the gradient of a bf16 matmul, up-cast) -- half the bytes of that product's store and of this load.
constexpr int LNB_ROWS = 128;
template <bool DY_BF16>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const void *dy_, const float *__restrict__ x, const float *__restrict__ gamma, const float eps,
              const float *skip, float *dx_out, unsigned short *__restrict__ dx_bf16, float *__restrict__ part,
              const int tokens) {
    const float *dy = (const float *)dy_;
    const unsigned short *dyh = (const unsigned short *)dy_;
    auto load_dy = [&](const size_t tok, const int lane) {
        if (DY_BF16) {
            const ushort4 h = ((const ushort4 *)(dyh + tok * 256))[lane];
            return make_float4(bf2f(h.x), bf2f(h.y), bf2f(h.z), bf2f(h.w));
        }
        return ((const float4 *)(dy + tok * 256))[lane];
    };
    __shared__ float red[4][12][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 g = ((const float4 *)gamma)[lane];
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f}, po[4] = {0.f, 0.f, 0.f, 0.f};
    const int tok0 = blockIdx.x * LNB_ROWS + wave * (LNB_ROWS / 4);
    const int nrow = min(LNB_ROWS / 4, tokens - tok0);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 vn = z4, dn = z4, kn = z4;
    if (nrow > 0) {
        vn = ((const float4 *)(x + (size_t)tok0 * 256))[lane];
        dn = load_dy((size_t)tok0, lane);
        if (skip) kn = ((const float4 *)(skip + (size_t)tok0 * 256))[lane];
    }
    for (int i = 0; i < nrow; i++) {
        const int tok = tok0 + i;
        const float4 v = vn, d = dn, k = kn;
        if (i + 1 < nrow) {
            vn = ((const float4 *)(x + (size_t)(tok + 1) * 256))[lane];
            dn = load_dy((size_t)(tok + 1), lane);
            if (skip) kn = ((const float4 *)(skip + (size_t)(tok + 1) * 256))[lane];
        }
        float s = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.0f / 256.0f);
        const float c0 = v.x - mean, c1 = v.y - mean, c2 = v.z - mean, c3 = v.w - mean;
        float q = c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + eps);
        const float h0 = c0 * rstd, h1 = c1 * rstd, h2 = c2 * rstd, h3 = c3 * rstd;
        const float a0 = d.x * g.x, a1 = d.y * g.y, a2 = d.z * g.z, a3 = d.w * g.w;
        float sa = a0 + a1 + a2 + a3, sh = a0 * h0 + a1 * h1 + a2 * h2 + a3 * h3;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sh += __shfl_xor(sh, o, 64); }
        sa *= (1.0f / 256.0f); sh *= (1.0f / 256.0f);
        const float4 r = make_float4(rstd * (a0 - sa - h0 * sh) + k.x, rstd * (a1 - sa - h1 * sh) + k.y,
                                     rstd * (a2 - sa - h2 * sh) + k.z, rstd * (a3 - sa - h3 * sh) + k.w);
        ((float4 *)(dx_out + (size_t)tok * 256))[lane] = r;
        if (dx_bf16) {
            ushort4 hb;
            hb.x = f2bf(r.x); hb.y = f2bf(r.y); hb.z = f2bf(r.z); hb.w = f2bf(r.w);
            ((ushort4 *)(dx_bf16 + (size_t)tok * 256))[lane] = hb;
        }
        pg[0] += d.x * h0; pg[1] += d.y * h1; pg[2] += d.z * h2; pg[3] += d.w * h3;
        pb[0] += d.x; pb[1] += d.y; pb[2] += d.z; pb[3] += d.w;
        po[0] += r.x; po[1] += r.y; po[2] += r.z; po[3] += r.w;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) { red[wave][c][lane] = pg[c]; red[wave][4 + c][lane] = pb[c]; red[wave][8 + c][lane] = po[c]; }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float t = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
            // quantity k / 4, channel 4 * lane + k % 4
            part[(size_t)blockIdx.x * 768 + (k >> 2) * 256 + 4 * lane + (k & 3)] = t;
        }
    }
}

// ---- column sums of a bf16 matrix [rows, C] (bias gradients): 64 rows per workgroup -> part[block][C] ----
__global__ void __launch_bounds__(256)
colsum_bf16_kernel(const unsigned short *__restrict__ src, float *__restrict__ part, const int rows, const int C) {
    const int r0 = blockIdx.x * 64, r1 = min(rows, r0 + 64);
    for (int c2 = threadIdx.x; c2 < C / 2; c2 += 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int rr = r0; rr < r1; rr++) {
            const uint32_t w = *(const uint32_t *)(src + (size_t)rr * C + 2 * c2);
            s0 += __uint_as_float(w << 16);
            s1 += __uint_as_float(w & 0xffff0000u);
        }
        part[(size_t)blockIdx.x * C + 2 * c2] = s0;
        part[(size_t)blockIdx.x * C + 2 * c2 + 1] = s1;
    }
}

__global__ void __launch_bounds__(256)
cast_bf16_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, const size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = ((const float4 *)src)[i];
    ushort4 h;
    h.x = f2bf(v.x); h.y = f2bf(v.y); h.z = f2bf(v.z); h.w = f2bf(v.w);
    ((ushort4 *)dst)[i] = h;
}

// gradient of the transposed convolution's output [scenes, 2R, 2R, 2R, Cout] fp32, gathered into the rows
// of its GEMM form: dog[m][tap * Cout + co] (bf16), tap = (i*2 + j)*2 + k
__global__ void __launch_bounds__(256)
deconv_grad_gather_kernel(const float *__restrict__ dout, unsigned short *__restrict__ dog, const int M, const int R,
                          const int Cout) {
    const int c4n = 2 * Cout;  // float4 groups per row (8 * Cout / 4)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * c4n) return;
    const int m = (int)(i / c4n), c = (int)(i - (size_t)m * c4n) * 4;
    const int tap = c / Cout, co = c - tap * Cout;
    int b, d, h, w;
    token_to_voxel(m, R, b, d, h, w);
    const size_t R2 = 2 * (size_t)R;
    const size_t o = ((((size_t)b * R2 + 2 * d + (tap >> 2)) * R2 + 2 * h + ((tap >> 1) & 1)) * R2 + 2 * w + (tap & 1)) * Cout + co;
    const float4 v = *(const float4 *)(dout + o);
    ushort4 hb;
    hb.x = f2bf(v.x); hb.y = f2bf(v.y); hb.z = f2bf(v.z); hb.w = f2bf(v.w);
    *(ushort4 *)(dog + (size_t)m * 8 * Cout + c) = hb;
}

// ---- backward of the per-group softmax attention (16 heads x head_dim 16, 8 queries x 4 keys) -----------
// Two groups per workgroup, 128 threads each.  Phase 1, thread = (head, query): scores, softmax, dP, dS
// and its dQ row; phase 2, thread = (head, key, half of the head's 16 channels): dK and dV as sums over
// the 8 queries.  The group's Q, dO and K|V rows (4 KB each) sit in LDS; dQ and dK|dV leave through LDS
// so that global traffic is whole 512- / 1024-byte rows.
__global__ void __launch_bounds__(256)
group_attn_bwd_kernel(const unsigned short *__restrict__ Q, const unsigned short *__restrict__ KV,
                      const unsigned short *__restrict__ dO, unsigned short *__restrict__ dQ,
                      unsigned short *__restrict__ dKV, const int G, const int lddkv /* elements between dK|dV rows (>= 512) */) {
    __shared__ __attribute__((aligned(16))) unsigned short s_q[2][8 * 256], s_do[2][8 * 256], s_kv[2][4 * 512], s_dq[2][8 * 256];
    __shared__ float s_p[2][16 * 8 * 4], s_ds[2][16 * 8 * 4];
    const int gl = threadIdx.x >> 7, t = threadIdx.x & 127;
    const int g = blockIdx.x * 2 + gl;
    const bool active = g < G;
    if (active) {
        const uint4 *gq = (const uint4 *)(Q + (size_t)g * 8 * 256), *gd = (const uint4 *)(dO + (size_t)g * 8 * 256);
        const uint4 *gk = (const uint4 *)(KV + (size_t)g * 4 * 512);
        ((uint4 *)s_q[gl])[t] = gq[t]; ((uint4 *)s_q[gl])[t + 128] = gq[t + 128];
        ((uint4 *)s_do[gl])[t] = gd[t]; ((uint4 *)s_do[gl])[t + 128] = gd[t + 128];
        ((uint4 *)s_kv[gl])[t] = gk[t]; ((uint4 *)s_kv[gl])[t + 128] = gk[t + 128];
    }
    __syncthreads();
    if (active) {
        const int h = t >> 3, i = t & 7;
        float qv[16], dv[16];
#pragma unroll
        for (int d = 0; d < 16; d++) { qv[d] = bf2f(s_q[gl][i * 256 + h * 16 + d]); dv[d] = bf2f(s_do[gl][i * 256 + h * 16 + d]); }
        float sc[4], dp[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int d = 0; d < 16; d++) {
                a += qv[d] * bf2f(s_kv[gl][j * 512 + h * 16 + d]);
                b += dv[d] * bf2f(s_kv[gl][j * 512 + 256 + h * 16 + d]);
            }
            sc[j] = a * 0.25f; dp[j] = b;
        }
        const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        float pr[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) { pr[j] = __expf(sc[j] - mx); sum += pr[j]; }
        const float inv = 1.0f / sum;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) { pr[j] *= inv; dot += pr[j] * dp[j]; }
        float ds[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ds[j] = pr[j] * (dp[j] - dot) * 0.25f;  // includes the 1/sqrt(head_dim) of the scores
            s_p[gl][(h * 8 + i) * 4 + j] = pr[j];
            s_ds[gl][(h * 8 + i) * 4 + j] = ds[j];
        }
#pragma unroll
        for (int d = 0; d < 16; d++) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) a += ds[j] * bf2f(s_kv[gl][j * 512 + h * 16 + d]);
            s_dq[gl][i * 256 + h * 16 + d] = f2bf(a);
        }
    }
    __syncthreads();
    float dk[8], dvv[8];
    const int h2 = t >> 3, j2 = (t >> 1) & 3, d0 = (t & 1) * 8;
    if (active) {
#pragma unroll
        for (int d = 0; d < 8; d++) { dk[d] = 0.f; dvv[d] = 0.f; }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float dsv = s_ds[gl][(h2 * 8 + i) * 4 + j2], pv = s_p[gl][(h2 * 8 + i) * 4 + j2];
#pragma unroll
            for (int d = 0; d < 8; d++) {
                dk[d] += dsv * bf2f(s_q[gl][i * 256 + h2 * 16 + d0 + d]);
                dvv[d] += pv * bf2f(s_do[gl][i * 256 + h2 * 16 + d0 + d]);
            }
        }
    }
    __syncthreads();  // phase 1's reads of K|V are done: reuse its LDS for dK|dV
    if (active) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
            s_kv[gl][j2 * 512 + h2 * 16 + d0 + d] = f2bf(dk[d]);
            s_kv[gl][j2 * 512 + 256 + h2 * 16 + d0 + d] = f2bf(dvv[d]);
        }
    }
    __syncthreads();
    if (active) {
        uint4 *oq = (uint4 *)(dQ + (size_t)g * 8 * 256);
        // (dK|dV rows: 64 uint4 each, `lddkv` elements apart; t and t + 128 are rows t >> 6 and 2 + (t >> 6))
        unsigned short *ok = dKV + ((size_t)g * 4 + (t >> 6)) * lddkv + (t & 63) * 8;
        oq[t] = ((const uint4 *)s_dq[gl])[t]; oq[t + 128] = ((const uint4 *)s_dq[gl])[t + 128];
        *(uint4 *)ok = ((const uint4 *)s_kv[gl])[t]; *(uint4 *)(ok + 2 * (size_t)lddkv) = ((const uint4 *)s_kv[gl])[t + 128];
    }
}

// ---- host helpers -------------------------------------------------------------------------------------
constexpr size_t TN_PART_BYTES = 128ull << 20;  // partial-tile buffer of the weight-gradient GEMMs

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

inline int cu_count() {   // the ring kernels run one workgroup per CU: launches are sized to one round
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        return v > 0 ? v : 256;
    }();
    return cus;
}

struct TnJob {
    const unsigned short *A; int lda, N;
    const unsigned short *B; int ldb, Kc;
    int M;
    float *dst;
};
int gemm_tn_group(const TnJob *jobs, int count, float *part, hipStream_t s);
// a plain product gemm_tn_group runs on the 256 x 256 ring kernel (its own conditions, restated: gemm_tn hands such a product over)
inline bool tn_ring_shape(int M, int N, int Kc, int lda, int ldb, const float *dst) {
    return M >= 128 && !(M & 127) && !(N & 255) && !(Kc & 7) && !(lda & 7) && !(ldb & 7) && (size_t)(M + 1) * lda * 2 < (1ull << 32) &&
           (size_t)(M + 1) * ldb * 2 < (1ull << 32) && M < (1 << 24) && ldb < (1 << 23) && (((uintptr_t)dst) & 15) == 0;
}

// dst[N, T*Kc] += A^T . B; `part` holds TN_PART_BYTES
int gemm_tn(const unsigned short *A, int lda, int N, const unsigned short *B, int ldb, int Kc, int T, const int *nbr,
            int M, float *dst, float *part, hipStream_t s) {
    if ((M & 15) || (N & 1) || (Kc & 1) || (T > 1 && (Kc & 127))) return LARA2DGS_E_INVALID;
    // 32-bit byte offsets and 24-bit row multiplies in the kernel
    if ((size_t)(M + 1) * lda * 2 >= (1ull << 32) || (size_t)(M + 1) * ldb * 2 >= (1ull << 32) || M >= (1 << 24) || ldb >= (1 << 23))
        return LARA2DGS_E_INVALID;
    const size_t out_bytes = (size_t)N * T * Kc * 4;
    const int n = N * T * Kc;
    if (!nbr && T == 1 && tn_ring_shape(M, N, Kc, lda, ldb, dst)) {
        const TnJob j{A, lda, N, B, ldb, Kc, M, dst};
        return gemm_tn_group(&j, 1, part, s);
    }
    if (nbr && N == 256 && Kc == 256 && (M & 127) == 0 && !(lda & 7) && !(ldb & 7) && (((uintptr_t)dst | (uintptr_t)part) & 15) == 0) {
        // the convolution's weight gradient: one 256 x 256 workgroup per (tap, split), one round of workgroups over the device
        const int cus = cu_count();
        static const hipError_t attr = hipFuncSetAttribute((const void *)gemm_tn_ring_kernel<true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, TR_RING * TR_STAGE);
        if (attr != hipSuccess) return LARA2DGS_E_LAUNCH;
        const int g128 = M / 128;
        int splits = max(1, min(min(cus / T, g128), (int)(TN_PART_BYTES / out_bytes)));
        const int chunk = ((g128 + splits - 1) / splits) * 128;
        splits = (M + chunk - 1) / chunk;   // every split has rows
        TnRingGroup g{};
        TnRingP &q = g.p[0];
        q.A = A; q.B = B; q.part = part; q.nbr = nbr; q.M = M; q.lda = lda; q.ldb = ldb; q.T = T; q.ntile = q.ktile = 1; q.Kc = 256;
        q.chunk = chunk; q.splits = splits;
        g.count = 1; g.first_wg[0] = 0; g.first_wg[1] = T * splits;
        hipLaunchKernelGGL(gemm_tn_ring_kernel<true>, dim3(T * splits), dim3(512), TR_RING * TR_STAGE, s, g);
        hipLaunchKernelGGL(accum_partials4_kernel, dim3((n / 4 + 63) / 64), dim3(256), 0, s, dst, part, n, splits, (size_t)n);
        return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
    }
    const int tiles = ((N + 127) / 128) * ((Kc + 127) / 128) * T;
    // about 1024 workgroups per launch (4 per CU: one round), in multiples of 8 splits (one per XCD, see the
    // kernel), at least 256 token rows each; a split that starts beyond M writes zeros
    const int want = 1024;
    int splits = max(1, min(want / tiles, (int)(TN_PART_BYTES / out_bytes)));
    splits = max(8, splits & ~7);
    while (splits > 8 && M / splits < 256) splits -= 8;
    const int chunk = (((M + splits - 1) / splits) + 63) & ~63;
    TnP p{};
    p.A = A; p.B = B; p.part = part; p.nbr = nbr; p.M = M; p.lda = lda; p.ldb = ldb; p.N = N; p.Kc = Kc; p.T = T;
    p.chunk = chunk; p.tiles = tiles;
    if ((size_t)splits * out_bytes > TN_PART_BYTES) return LARA2DGS_E_INVALID;
    const bool staged = !((M & 31) || (N & 7) || (Kc & 7) || (lda & 7) || (ldb & 7));
    if (staged) {
        TnGroup g{};
        g.p[0] = p; g.count = 1; g.first_slot[0] = 0; g.first_slot[1] = tiles * splits / 8;
        if (nbr) hipLaunchKernelGGL((gemm_bf16_tn_lds_kernel<true>), dim3(tiles * splits), dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_bf16_tn_lds_kernel<false>), dim3(tiles * splits), dim3(256), 0, s, g);
    } else {
        if (nbr) hipLaunchKernelGGL((gemm_bf16_tn_kernel<true>), dim3(tiles * splits), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_tn_kernel<false>), dim3(tiles * splits), dim3(256), 0, s, p);
    }
    if ((n & 3) == 0 && (((uintptr_t)dst | (uintptr_t)part) & 15) == 0 && (n / 4 + 63) / 64 < 128 && splits >= 64)
        hipLaunchKernelGGL(accum_partials4_wide_kernel, dim3((n / 4 + 63) / 64), dim3(1024), 0, s, dst, part, n, splits, (size_t)n);
    else if ((n & 3) == 0 && (((uintptr_t)dst | (uintptr_t)part) & 15) == 0)
        hipLaunchKernelGGL(accum_partials4_kernel, dim3((n / 4 + 63) / 64), dim3(256), 0, s, dst, part, n, splits, (size_t)n);
    else
        hipLaunchKernelGGL(accum_partials_kernel, dim3((n + 63) / 64), dim3(256), 0, s, dst, part, n, splits, (size_t)n);
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

// dst_k[N_k, Kc_k] += A_k^T . B_k for several products that are ready at the same time, as one launch + one reduction launch
// (plain products only: no gathered rows, T = 1).  Falls back to one launch per product where the staged kernel's alignment
// conditions do not hold.
int gemm_tn_group(const TnJob *jobs, int count, float *part, hipStream_t s) {
    bool ok = count >= 1 && count <= TN_GROUP_MAX;
    int tiles_all = 0;
    for (int k = 0; k < count && ok; k++) {
        const TnJob &j = jobs[k];
        ok = !((j.M & 31) || (j.N & 7) || (j.Kc & 7) || (j.lda & 7) || (j.ldb & 7)) && ((j.N * j.Kc) & 3) == 0 &&
             (size_t)(j.M + 1) * j.lda * 2 < (1ull << 32) && (size_t)(j.M + 1) * j.ldb * 2 < (1ull << 32) && j.M < (1 << 24) &&
             j.ldb < (1 << 23) && (((uintptr_t)j.dst) & 15) == 0;      // (the same 24-bit row multiplies as gemm_tn)
        tiles_all += ((j.N + 127) / 128) * ((j.Kc + 127) / 128);
    }
    if (!ok) {
        for (int k = 0; k < count; k++) {
            const int rc = gemm_tn(jobs[k].A, jobs[k].lda, jobs[k].N, jobs[k].B, jobs[k].ldb, jobs[k].Kc, 1, nullptr, jobs[k].M,
                                   jobs[k].dst, part, s);
            if (rc) return rc;
        }
        return LARA2DGS_OK;
    }
    // every product in 256 x 256 tiles and 128-row groups: the ring kernel, one round of workgroups (one per CU), every workgroup the
    // same number of token rows (the largest chunk that keeps the launch within the CU count)
    bool ringable = true;
    long units = 0;
    for (int k = 0; k < count; k++) {
        const TnJob &j = jobs[k];
        ringable = ringable && tn_ring_shape(j.M, j.N, j.Kc, j.lda, j.ldb, j.dst);   // (a ragged last column tile is fine)
        units += (long)(j.N / 256) * ((j.Kc + 255) / 256) * (j.M / 128);
    }
    if (ringable) {
        static const hipError_t attr = hipFuncSetAttribute((const void *)gemm_tn_ring_kernel<false>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, TR_RING * TR_STAGE);
        if (attr != hipSuccess) return LARA2DGS_E_LAUNCH;
        const int cus = cu_count();
        TnRingGroup g{};
        AccGroup a{};
        g.count = a.count = count;
        for (int cg = (int)max(1l, (units + cus - 1) / cus);; cg++) {   // cg: 128-row groups per workgroup
            int wgs = 0, rwg = 0;
            size_t off_f = 0;
            for (int k = 0; k < count; k++) {
                const TnJob &j = jobs[k];
                const int tiles = (j.N / 256) * ((j.Kc + 255) / 256), n = j.N * j.Kc, splits = (j.M / 128 + cg - 1) / cg;
                TnRingP &q = g.p[k];
                q.A = j.A; q.B = j.B; q.part = part + off_f; q.nbr = nullptr; q.M = j.M; q.lda = j.lda; q.ldb = j.ldb; q.T = 1;
                q.ntile = j.N / 256; q.ktile = (j.Kc + 255) / 256; q.Kc = j.Kc; q.chunk = cg * 128; q.splits = splits;
                g.first_wg[k] = wgs;
                wgs += tiles * splits;
                a.dst[k] = j.dst; a.part[k] = part + off_f; a.n[k] = n; a.parts[k] = splits; a.first_wg[k] = rwg;
                rwg += (n / 4 + 63) / 64;
                off_f += (size_t)splits * n;
            }
            g.first_wg[count] = wgs;
            a.first_wg[count] = rwg;
            if (wgs <= cus && off_f * 4 <= TN_PART_BYTES) break;
            if (cg > (1 << 20)) return LARA2DGS_E_INVALID;
        }
        hipLaunchKernelGGL(gemm_tn_ring_kernel<false>, dim3(g.first_wg[count]), dim3(512), TR_RING * TR_STAGE, s, g);
        hipLaunchKernelGGL(accum_partials4_group_kernel, dim3(a.first_wg[count]), dim3(256), 0, s, a);
        return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
    }
    // about 1280 workgroups (five per CU: the kernel is latency-bound below four), splits in multiples of 8 (one per XCD), at
    // least 256 token rows per split
    const int want = 1280;
    int base = max(8, (want / tiles_all) & ~7);
    TnGroup g{};
    AccGroup a{};
    g.count = a.count = count;
    size_t off_f = 0;     // floats
    for (;;) {
        off_f = 0;
        int slot = 0, wg = 0;
        for (int k = 0; k < count; k++) {
            const TnJob &j = jobs[k];
            int splits = base;
            while (splits > 8 && j.M / splits < 256) splits -= 8;
            const int tiles = ((j.N + 127) / 128) * ((j.Kc + 127) / 128), n = j.N * j.Kc;
            TnP &p = g.p[k];
            p.A = j.A; p.B = j.B; p.part = part + off_f; p.nbr = nullptr; p.M = j.M; p.lda = j.lda; p.ldb = j.ldb; p.N = j.N; p.Kc = j.Kc;
            p.T = 1; p.chunk = (((j.M + splits - 1) / splits) + 63) & ~63; p.tiles = tiles;
            g.first_slot[k] = slot;
            slot += tiles * splits / 8;
            a.dst[k] = j.dst; a.part[k] = part + off_f; a.n[k] = n; a.parts[k] = splits; a.first_wg[k] = wg;
            wg += (n / 4 + 63) / 64;
            off_f += (size_t)splits * n;
        }
        g.first_slot[count] = slot;
        a.first_wg[count] = wg;
        if (off_f * 4 <= TN_PART_BYTES || base == 8) break;
        base -= 8;
    }
    if (off_f * 4 > TN_PART_BYTES) return LARA2DGS_E_INVALID;
    hipLaunchKernelGGL((gemm_bf16_tn_lds_kernel<false>), dim3(g.first_slot[count] * 8), dim3(256), 0, s, g);
    hipLaunchKernelGGL(accum_partials4_group_kernel, dim3(a.first_wg[count]), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

// LayerNorm backward + the reductions of its partial sums into dgamma, dbeta, dbias (any may be null)
// pending column reductions of a block (reduce_group_kernel)
struct RedList {
    RedGroup g{};
    int wgs = 0;
    bool overflow = false;
    void add(float *dst, const float *part, int n, int parts, int stride) {
        if (!dst) return;
        if (g.count >= RED_GROUP_MAX) { overflow = true; return; }
        const int k = g.count++;
        g.dst[k] = dst; g.part[k] = part; g.n[k] = n; g.parts[k] = parts; g.stride[k] = stride; g.first_wg[k] = wgs;
        wgs += (n + 63) / 64;
        g.first_wg[g.count] = wgs;
    }
    int launch(hipStream_t s) {
        if (overflow) return LARA2DGS_E_INVALID;
        if (g.count) hipLaunchKernelGGL(reduce_group_kernel, dim3(wgs), dim3(1024), 0, s, g);
        g.count = 0; wgs = 0;
        return LARA2DGS_OK;
    }
};

// `later`: the three reductions are queued there (the caller launches them with others) instead of launched here; `part` must
// then stay untouched until it does
int ln_bwd(const void *dy, bool dy_bf16, const float *x, const float *gamma, float eps, const float *skip, float *dx,
           unsigned short *dx_bf16, float *dgamma, float *dbeta, float *dbias, float *part, int M, hipStream_t s,
           RedList *later = nullptr) {
    const int blocks = (M + LNB_ROWS - 1) / LNB_ROWS;
    if (dy_bf16) hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(blocks), dim3(256), 0, s, dy, x, gamma, eps, skip, dx, dx_bf16, part, M);
    else hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(blocks), dim3(256), 0, s, dy, x, gamma, eps, skip, dx, dx_bf16, part, M);
    if (later) {
        later->add(dgamma, part, 256, blocks, 768);
        later->add(dbeta, part + 256, 256, blocks, 768);
        later->add(dbias, part + 512, 256, blocks, 768);
    } else {
        hipLaunchKernelGGL(accum_ln_partials_kernel, dim3(4, 3), dim3(1024), 0, s, dgamma, dbeta, dbias, part, blocks);
    }
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

int colsum_bf16(const unsigned short *src, int rows, int C, float *dst, float *part, hipStream_t s) {
    const int blocks = (rows + 63) / 64;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(blocks), dim3(256), 0, s, src, part, rows, C);
    hipLaunchKernelGGL(accum_partials_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, dst, part, C, blocks, (size_t)C);
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

template <int EPI>
void gemm_nt(const unsigned short *A, const unsigned short *W, void *C, int M, int N, int K, const float *resid,
             unsigned short *C2, hipStream_t s, float *colsum = nullptr) {
    GemmP p{};
    p.A = A; p.W = W; p.C = C; p.resid = resid; p.C2 = C2; p.M = M; p.N = N; p.K = K; p.colsum = colsum;
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, EPI>), dim3((M + 127) / 128, (N + 127) / 128), dim3(256), 0, s, p);
}

// What a block's forward leaves behind for its backward (lara_groupblock_forward_train), in bytes from the
// start of the block's save area: 7.2 KB per token row, 0.94 GB per layer at 4 scenes x 32^3 -- 11 GB for the
// 12 layers, which a 288 GB part holds without thinking (recomputing them instead costs 0.6 ms per layer).
struct SaveWs {
    size_t xn1, q, kv, o, x1, xn2, z, h, x2, xn3, stats, wpack, total;
};
SaveWs save_layout(int64_t M) {
    SaveWs w{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = up256(o + bytes); return at; };
    const size_t b256 = (size_t)M * 512, f256 = (size_t)M * 1024, b512 = (size_t)M * 1024;
    w.xn1 = take(b256); w.q = take(b256); w.kv = take(b256); w.o = take(b256);
    w.x1 = take(f256); w.xn2 = take(b256); w.z = take(b512); w.h = take(b512); w.x2 = take(f256);
    w.xn3 = take(b256 + 512);  // + the zeroed row (index M) the gathers read outside the volume
    w.stats = take((size_t)M * 8);
    w.wpack = take(2 * 256 * 256 * 2);   // W_q and W_o in the fused attention kernel's fragment order
    w.total = o;
    return w;
}
// scratch of lara_groupblock_backward (followed by a SaveWs for the recompute mode)
struct BwdWs {
    size_t gb, tmpf, dzb, dq, dkv, dob, nbr, lnpart, tnpart, save, gb3, gb2, lnpart4, total;
};
BwdWs bwd_layout(int64_t M) {
    BwdWs w{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = up256(o + bytes); return at; };
    const size_t b256 = (size_t)M * 512, f256 = (size_t)M * 1024, b512 = (size_t)M * 1024;
    w.gb = take(b256 + 512);   // + a zeroed row, as above
    w.tmpf = take(f256); w.dzb = take(b512); w.dq = take(b256); w.dkv = take(b256); w.dob = take(b256);
    w.nbr = take((size_t)27 * M * 4);
    w.lnpart = take(((size_t)(M + 63) / 64) * 768 * 4);
    w.tnpart = take(TN_PART_BYTES);
    w.save = take(save_layout(M).total);
    w.gb3 = take(b256); w.gb2 = take(b256);
    w.lnpart4 = take(4 * ((size_t)(M + LNB_ROWS - 1) / LNB_ROWS) * 768 * 4);   // four sets of per-workgroup partial column sums, reduced together   // bf16(g) behind norm3's / norm2's backward, kept for the block's grouped weight gradients
    w.total = o;
    return w;
}

// network.py:88-95 of one block, every intermediate kept in `save` (x_in is not modified)
int block_forward_keep(int M, int cond_dim, const float *x_in, const unsigned short *cond_bf16,
                       const lara_groupblock_weights *w, char *save, hipStream_t s) {
    const SaveWs L = save_layout(M);
    const int G = M / 8, Mkv = G * 4, lnb = (M + 3) / 4;
    unsigned short *xn1 = (unsigned short *)(save + L.xn1), *q = (unsigned short *)(save + L.q), *kv = (unsigned short *)(save + L.kv);
    unsigned short *o = (unsigned short *)(save + L.o), *xn2 = (unsigned short *)(save + L.xn2), *z = (unsigned short *)(save + L.z);
    unsigned short *h = (unsigned short *)(save + L.h), *xn3 = (unsigned short *)(save + L.xn3);
    float *x1 = (float *)(save + L.x1), *x2 = (float *)(save + L.x2);
    // the attention step: the K|V projection + ONE fused kernel (LayerNorm, Q projection, attention, output projection +
    // residual) that also leaves LN(x), Q and the attention's output for the backward (group_attn_fused2_kernel<true>; until
    // round 5 the training forward ran the five separate launches because only they kept those rows: 284 -> ~240 us per layer)
    {
        GemmP p{};
        p.A = cond_bf16; p.W = w->wkv; p.C = kv; p.M = Mkv; p.N = 512; p.K = cond_dim;
        if (launch_gemm_ring<0, 0>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
    {
        unsigned short *wqp = (unsigned short *)(save + L.wpack), *wop = wqp + 65536;
        hipLaunchKernelGGL(pack_weight_frag_kernel, dim3(64), dim3(256), 0, s, w->wq, wqp, w->wo, wop);
        hipLaunchKernelGGL(group_attn_fused2_kernel<true>, dim3((G + 3) / 4), dim3(64), 0, s, x_in, w->ln1_w, w->ln1_b, w->eps, wqp, kv, wop,
                           x1, G, xn1, q, o);
    }
#ifndef LARA_MLP_UNFUSED
    (void)lnb;
    {   // norm2 -> fc1 -> GELU -> fc2 -> + x1 -> norm3 as one kernel per 128-row tile (mlp_fused.h); leaves xn2, z, h, x2, xn3, stats
        MlpP p{};
        p.x1 = x1; p.x2 = x2; p.ln2_w = w->ln2_w; p.ln2_b = w->ln2_b; p.b1 = w->b1; p.b2 = w->b2; p.ln3_w = w->ln3_w; p.ln3_b = w->ln3_b;
        p.w1 = w->w1; p.w2 = w->w2; p.xn3 = xn3; p.stats = (float2 *)(save + L.stats); p.xn2 = xn2; p.z = z; p.h = h; p.eps = w->eps; p.M = M;
        if (launch_mlp_fused<1>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
#else       // (rounds 2-5: four launches; tools/build_variant.sh -DLARA_MLP_UNFUSED for A/B runs)
    hipLaunchKernelGGL(ln_cast_kernel, dim3(lnb), dim3(256), 0, s, x1, w->ln2_w, w->ln2_b, w->eps, xn2, (float2 *)nullptr, M);
    {
        GemmP p{};
        p.A = xn2; p.W = w->w1; p.C = h; p.C2 = z; p.bias = w->b1; p.M = M; p.N = 512; p.K = 256;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 6>), dim3((M + 127) / 128, 4), dim3(256), 0, s, p);
    }
    {
        GemmP p{};
        p.A = h; p.W = w->w2; p.C = x2; p.resid = x1; p.bias = w->b2; p.M = M; p.N = 256; p.K = 512;
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<0, 3>), dim3((M + 127) / 128, 2), dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(ln_cast_kernel, dim3(lnb), dim3(256), 0, s, x2, w->ln3_w, w->ln3_b, w->eps, xn3,
                       (float2 *)(save + L.stats), M);
#endif
#ifdef LARA_MLP_UNFUSED
    if (hipMemsetAsync(xn3 + (size_t)M * 256, 0, 512, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
#endif      // (the fused kernel zero-fills row M of xn3: mlp_fused.h)
    return hipGetLastError() == hipSuccess ? LARA2DGS_OK : LARA2DGS_E_LAUNCH;
}

bool block_weights_ok(const lara_groupblock_weights *w) {
    return w && w->ln1_w && w->ln1_b && w->wq && w->wkv && w->wo && w->ln2_w && w->ln2_b && w->w1 && w->b1 && w->w2 &&
           w->b2 && w->ln3_w && w->ln3_b && w->wconv;
}

}  // namespace

extern "C" {

int64_t lara_groupblock_backward_workspace_bytes(int32_t scenes, int32_t R) {
    if (scenes < 0 || R < 4 || (R & 1)) return LARA2DGS_E_INVALID;
    return (int64_t)bwd_layout((int64_t)scenes * R * R * R).total;
}

int64_t lara_groupblock_save_bytes(int32_t scenes, int32_t R) {
    if (scenes < 0 || R < 4 || (R & 1)) return LARA2DGS_E_INVALID;
    return (int64_t)save_layout((int64_t)scenes * R * R * R).total;
}

int lara_groupblock_forward_train(int32_t scenes, int32_t R, int32_t cond_dim, const float *x_in, float *x_out,
                                  const uint16_t *cond_bf16, const lara_groupblock_weights *w, void *saved,
                                  void *stream) {
    if (scenes < 0 || R < 4 || (R & 1) || cond_dim <= 0 || (cond_dim % 32) != 0 || !block_weights_ok(w)) return LARA2DGS_E_INVALID;
    if (scenes == 0) return LARA2DGS_OK;
    if (!x_in || !x_out || x_in == x_out || !cond_bf16 || !saved) return LARA2DGS_E_INVALID;
    const int64_t M64 = (int64_t)scenes * R * R * R;
    if (M64 * 2056 + 512 >= (1ll << 32)) return LARA2DGS_E_INVALID;
    const int M = (int)M64;
    hipStream_t s = (hipStream_t)stream;
    const SaveWs L = save_layout(M);
    char *save = (char *)saved;
    int rc;
    {
        L2D_PROF("gbt_forward", s);
        if ((rc = block_forward_keep(M, cond_dim, x_in, cond_bf16, w, save, s))) return rc;
    }
    {
        L2D_PROF("gb_conv3d", s);
        GemmP p{};
        p.A = (const unsigned short *)(save + L.xn3); p.W = w->wconv; p.C = x_out; p.resid = (const float *)(save + L.x2);
        p.M = M; p.N = 256; p.K = 27 * 256; p.R = R; p.Cin = 256; p.stats = (const float2 *)(save + L.stats);
        p.gamma = w->ln3_w; p.beta = w->ln3_b; p.zero_off = (uint32_t)((size_t)M * 512);
        if (launch_gemm_ring<1, 4>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_groupblock_backward(int32_t scenes, int32_t R, int32_t cond_dim, const float *x_in,
                             const uint16_t *cond_bf16, const lara_groupblock_weights *w,
                             const lara_groupblock_weights_t *wt, const void *saved, float *g, float *dcond,
                             const lara_groupblock_grads *dw, int32_t chained, uint16_t *dkv_ext, int32_t lddkv,
                             void *workspace, void *stream) {
    if (scenes < 0 || R < 4 || (R & 1) || cond_dim <= 0 || (cond_dim % 32) != 0 || !block_weights_ok(w) || !wt || !dw)
        return LARA2DGS_E_INVALID;
    if (scenes == 0) return LARA2DGS_OK;
    if (!x_in || !cond_bf16 || !g || (!dcond && !dkv_ext) || !workspace) return LARA2DGS_E_INVALID;
    if (dkv_ext && (lddkv < 512 || (lddkv & 7))) return LARA2DGS_E_INVALID;
    if (!wt->wq_t || !wt->wkv_t || !wt->wo_t || !wt->w1_t || !wt->w2_t ||
        !wt->wconv_t || !dw->ln1_w || !dw->ln1_b || !dw->wq || !dw->wkv || !dw->wo || !dw->ln2_w || !dw->ln2_b ||
        !dw->w1 || !dw->b1 || !dw->w2 || !dw->b2 || !dw->ln3_w || !dw->ln3_b || !dw->wconv)
        return LARA2DGS_E_INVALID;
    const int64_t M64 = (int64_t)scenes * R * R * R;
    if (M64 * 2056 + 512 >= (1ll << 32)) return LARA2DGS_E_INVALID;
    const int M = (int)M64, G = M / 8, Mkv = G * 4;
    hipStream_t s = (hipStream_t)stream;
    const BwdWs L = bwd_layout(M);
    const SaveWs S = save_layout(M);
    char *ws = (char *)workspace;
    int rc;
    if (!saved) {  // nothing kept from the forward: run it again into our own save area
        L2D_PROF("gbb_recompute", s);
        if ((rc = block_forward_keep(M, cond_dim, x_in, cond_bf16, w, ws + L.save, s))) return rc;
    }
    const char *sv = saved ? (const char *)saved : ws + L.save;
    const unsigned short *xn1 = (const unsigned short *)(sv + S.xn1), *q = (const unsigned short *)(sv + S.q);
    const unsigned short *kv = (const unsigned short *)(sv + S.kv), *o = (const unsigned short *)(sv + S.o);
    const unsigned short *xn2 = (const unsigned short *)(sv + S.xn2), *h = (const unsigned short *)(sv + S.h);
    const unsigned short *xn3 = (const unsigned short *)(sv + S.xn3);
    unsigned short *z = (unsigned short *)(sv + S.z);  // (read only; the GEMM parameter block is not const-correct)
    const float *x1 = (const float *)(sv + S.x1), *x2 = (const float *)(sv + S.x2);
    unsigned short *gb = (unsigned short *)(ws + L.gb);
    unsigned short *gb3 = (unsigned short *)(ws + L.gb3), *gb2 = (unsigned short *)(ws + L.gb2);
    unsigned short *dzb = (unsigned short *)(ws + L.dzb), *dq = (unsigned short *)(ws + L.dq);
    // dK|dV: into the caller's all-layers buffer (its dcond product runs once, after the sweep) or into the workspace
    unsigned short *dkv = dkv_ext ? dkv_ext : (unsigned short *)(ws + L.dkv);
    const int ld_dkv = dkv_ext ? lddkv : 512;
    unsigned short *dob = (unsigned short *)(ws + L.dob);
    unsigned short *tmpb = (unsigned short *)(ws + L.tmpf);   // dX of the MLP / of the Q projection, bf16 [M, 256]
    float *lnpart = (float *)(ws + L.lnpart), *tnpart = (float *)(ws + L.tnpart);
    // the block's four sets of per-workgroup partial column sums (three LayerNorm backward passes, the b1 gradient) live side by
    // side and are reduced by ONE launch at the end of the block
    const size_t lnset = ((size_t)(M + LNB_ROWS - 1) / LNB_ROWS) * 768;
    float *lnp4 = (float *)(ws + L.lnpart4);
    RedList red;
    (void)lnpart;
    int *nbr = (int *)(ws + L.nbr);
    if (!chained) {  // (a chained call finds the zero row, the neighbour table and bf16(g) where the call before left them)
        if (hipMemsetAsync(gb + (size_t)M * 256, 0, 512, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
        hipLaunchKernelGGL(neighbour_table_kernel, dim3((M + 255) / 256), dim3(256), 0, s, nbr, M, R, M, 512);
        hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(((size_t)M * 64 + 255) / 256)), dim3(256), 0, s, g, gb, (size_t)M * 64);
        L2D_CHECK_LAUNCH();
    }
    // ---- x_out = pn + cnn(pn), pn = norm3(x2)  (network.py:94-100) ----
    {
        L2D_PROF("gbb_dw_conv", s);
        if ((rc = gemm_tn(gb, 256, 256, xn3, 256, 256, 27, nbr, M, dw->wconv, tnpart, s))) return rc;
    }
    // d pn = g + cnn^T(g): the same implicit GEMM with the taps mirrored and in/out swapped; and behind it the backward of
    // pn = norm3(x2) -- in the product's epilogue where the tile holds whole rows (EPI 9, mfma_gemm.h), as a pass of its own otherwise.
    // The bf16 copy of g that each LayerNorm backward leaves goes to its own buffer (gb3 behind norm3, gb2 behind norm2, gb --
    // the one the next block's convolution gathers from -- behind norm1), so that the five weight gradients of the block's linear
    // layers, whose operands are then all alive at the end of the block, run as ONE grouped product (gemm_tn_group).
    {
        GemmP p{};
        p.A = gb; p.W = wt->wconv_t; p.C = g; p.resid = g; p.M = M; p.N = 256; p.K = 27 * 256;
        p.R = R; p.Cin = 256; p.zero_off = (uint32_t)((size_t)M * 512);
        if (ring2_shape(p, 1)) {
            L2D_PROF("gbb_dx_conv", s);
            p.C2 = gb3; p.lnx = x2; p.stats = (const float2 *)(sv + S.stats); p.gamma = w->ln3_w; p.colsum = lnp4;
            if (launch_gemm_ring<1, 9>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
            const int tiles = (M + RT - 1) / RT;
            red.add(dw->ln3_w, lnp4, 256, tiles, 768);
            red.add(dw->ln3_b, lnp4 + 256, 256, tiles, 768);
            red.add(dw->b2, lnp4 + 512, 256, tiles, 768);
        } else {
            {
                L2D_PROF("gbb_dx_conv", s);
                if (launch_gemm_ring<1, 1>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
            }
            L2D_PROF("gbb_ln_bwd", s);
            if ((rc = ln_bwd(g, false, x2, w->ln3_w, w->eps, nullptr, g, gb3, dw->ln3_w, dw->ln3_b, dw->b2, lnp4, M, s, &red))) return rc;
        }
    }
    L2D_CHECK_LAUNCH();
    // ---- x2 = x1 + mlp(norm2(x1)) ----
#ifndef LARA_MLP_UNFUSED
    {
        // dz = (g2 W2) * gelu'(z) and dy = dz W1 as ONE kernel per 128-row tile (mlp_fused.h): dz stays in LDS between the two products
        // (it still leaves for the weight gradient of fc1), its per-tile column sums are the pieces of db1.  LARA_MLP_BWD_LN: norm2's
        // backward in the same kernel's epilogue (MODE 2) -- measured slower than the pass of its own (profiles/r06_mlp_fused_phases.txt)
        L2D_PROF("gbb_dx_mlp", s);
        MlpP p{};
        p.gin = gb3; p.w1 = wt->w2_t; p.w2 = wt->w1_t; p.z = z; p.h = dzb; p.x1 = x1; p.ln2_w = w->ln2_w; p.eps = w->eps;
        p.part_b1 = lnp4 + lnset; p.M = M;
        const int tiles = (M + 127) / 128;
#ifdef LARA_MLP_BWD_LN
        p.g = g; p.gout = gb2; p.part_ln = lnp4 + 2 * lnset;
        if (launch_mlp_fused<2>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
        red.add(dw->ln2_w, lnp4 + 2 * lnset, 256, tiles, 768);
        red.add(dw->ln2_b, lnp4 + 2 * lnset + 256, 256, tiles, 768);
#else
        p.gout = tmpb;
        if (launch_mlp_fused<3>(p, s) != hipSuccess) return LARA2DGS_E_LAUNCH;
#endif
        red.add(dw->b1, lnp4 + lnset, 512, tiles, 512);
    }
#ifndef LARA_MLP_BWD_LN
    {
        L2D_PROF("gbb_ln_bwd", s);
        if ((rc = ln_bwd(tmpb, true, x1, w->ln2_w, w->eps, g, g, gb2, dw->ln2_w, dw->ln2_b, nullptr, lnp4 + 2 * lnset, M, s, &red))) return rc;
    }
#endif
#else
    {
        L2D_PROF("gbb_dx_mlp", s);
        // dz = (g2 W2) * gelu'(z); its column sums per 128-row tile (= the pieces of db1) come out of the same epilogue
        gemm_nt<7>(gb3, wt->w2_t, dzb, M, 512, 256, nullptr, z, s, lnp4 + lnset);
        red.add(dw->b1, lnp4 + lnset, 512, (M + 127) / 128, 512);
        gemm_nt<0>(dzb, wt->w1_t, tmpb, M, 256, 512, nullptr, nullptr, s);   // bf16: see ln_bwd_kernel
    }
    {
        L2D_PROF("gbb_ln_bwd", s);
        if ((rc = ln_bwd(tmpb, true, x1, w->ln2_w, w->eps, g, g, gb2, dw->ln2_w, dw->ln2_b, nullptr, lnp4 + 2 * lnset, M, s, &red))) return rc;
    }
#endif
    L2D_CHECK_LAUNCH();
    // ---- x1 = x0 + cross_attn(norm1(x0), cond, cond) ----
    {
        L2D_PROF("gbb_dx_attn", s);
        gemm_nt<0>(gb2, wt->wo_t, dob, M, 256, 256, nullptr, nullptr, s);
        hipLaunchKernelGGL(group_attn_bwd_kernel, dim3((G + 1) / 2), dim3(256), 0, s, q, kv, dob, dq, dkv, G, ld_dkv);
        if (!dkv_ext) gemm_nt<1>(dkv, wt->wkv_t, dcond, Mkv, cond_dim, 512, dcond, nullptr, s);
        gemm_nt<0>(dq, wt->wq_t, tmpb, M, 256, 256, nullptr, nullptr, s);
    }
    {
        L2D_PROF("gbb_ln_bwd", s);
        if ((rc = ln_bwd(tmpb, true, x_in, w->ln1_w, w->eps, g, g, gb, dw->ln1_w, dw->ln1_b, nullptr, lnp4 + 3 * lnset, M, s, &red))) return rc;
        if ((rc = red.launch(s))) return rc;      // dgamma / dbeta of the three LayerNorms, b2, b1: eight column reductions, one launch
    }
    {
        L2D_PROF("gbb_dw_linear", s);      // dW2, dW1, dWo, dWq, dWkv
        const TnJob jobs[5] = {{gb3, 256, 256, h, 512, 512, M, dw->w2},
                               {dzb, 512, 512, xn2, 256, 256, M, dw->w1},
                               {gb2, 256, 256, o, 256, 256, M, dw->wo},
                               {dq, 256, 256, xn1, 256, 256, M, dw->wq},
                               {dkv, ld_dkv, 512, cond_bf16, cond_dim, cond_dim, Mkv, dw->wkv}};
        if ((rc = gemm_tn_group(jobs, 5, tnpart, s))) return rc;
    }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int64_t lara_voltrans_head_backward_workspace_bytes(int32_t scenes, int32_t R, int32_t Cout) {
    if (scenes < 0 || R < 4 || (R & 1) || Cout <= 0 || (Cout & 15)) return LARA2DGS_E_INVALID;
    const size_t M = (size_t)scenes * R * R * R;
    return (int64_t)(up256(M * 512) + up256(M * 16 * Cout) + up256(M * 1024) + up256(((M + 63) / 64) * 4 * (size_t)max(768, 8 * Cout)) +
                     TN_PART_BYTES);
}

int lara_voltrans_head_backward(int32_t scenes, int32_t R, const float *x, const float *ln_w, const float *ln_b,
                                float eps, const uint16_t *wdeconv_t, int32_t Cout, const float *dout, float *g,
                                float *d_ln_w, float *d_ln_b, float *d_wdeconv, float *d_bias8, void *workspace,
                                void *stream) {
    if (scenes < 0 || R < 4 || (R & 1) || Cout <= 0 || (Cout & 15)) return LARA2DGS_E_INVALID;  // K = 8 Cout in steps of 32... 128
    if (scenes == 0) return LARA2DGS_OK;
    if (!x || !ln_w || !ln_b || !wdeconv_t || !dout || !g || !d_ln_w || !d_ln_b || !d_wdeconv || !d_bias8 || !workspace)
        return LARA2DGS_E_INVALID;
    const int M = scenes * R * R * R, N8 = 8 * Cout;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    size_t off = 0;
    unsigned short *xn = (unsigned short *)(ws + off); off += up256((size_t)M * 512);
    unsigned short *dog = (unsigned short *)(ws + off); off += up256((size_t)M * 16 * Cout);
    float *tmpf = (float *)(ws + off); off += up256((size_t)M * 1024);
    float *lnpart = (float *)(ws + off); off += up256(((size_t)(M + 63) / 64) * 4 * (size_t)max(768, N8));
    float *tnpart = (float *)(ws + off);
    int rc;
    L2D_PROF("vtb_head", s);
    hipLaunchKernelGGL(ln_cast_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, ln_w, ln_b, eps, xn, (float2 *)nullptr, M);
    {
        const size_t n = (size_t)M * 2 * Cout;
        hipLaunchKernelGGL(deconv_grad_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dout, dog, M, R, Cout);
    }
    if ((rc = colsum_bf16(dog, M, N8, d_bias8, lnpart, s))) return rc;
    if ((rc = gemm_tn(dog, N8, N8, xn, 256, 256, 1, nullptr, M, d_wdeconv, tnpart, s))) return rc;
    gemm_nt<0>(dog, wdeconv_t, tmpf, M, 256, N8, nullptr, nullptr, s);   // (bf16 rows in the fp32-sized region)
    if ((rc = ln_bwd(tmpf, true, x, ln_w, eps, nullptr, g, nullptr, d_ln_w, d_ln_b, nullptr, lnpart, M, s))) return rc;
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

/* ---- unit entry points (parity tests of the individual kernels) ---- */

int lara_gemm_nt_bf16(int32_t M, int32_t N, int32_t K, const uint16_t *A, const uint16_t *W, void *C, int32_t c_fp32,
                      void *stream) {
    if (M < 0 || N < 0 || K <= 0 || (K & 31)) return LARA2DGS_E_INVALID;
    if (M == 0 || N == 0) return LARA2DGS_OK;
    if (!A || !W || !C) return LARA2DGS_E_INVALID;
    if ((size_t)M * K * 2 >= (1ull << 32) || (size_t)N * K * 2 >= (1ull << 32)) return LARA2DGS_E_INVALID;  // 32-bit operand offsets
    GemmP p{};
    p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K;
    const hipError_t e = c_fp32 ? launch_gemm_ring<0, 8>(p, (hipStream_t)stream) : launch_gemm_ring<0, 0>(p, (hipStream_t)stream);
    if (e != hipSuccess) { l2d_set_hip_error(e); return LARA2DGS_E_LAUNCH; }
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

int lara_gemm_tn_bf16(int32_t M, int32_t N, int32_t Kc, const uint16_t *A, const uint16_t *B, float *dst,
                      void *workspace, void *stream) {
    if (M <= 0 || N <= 0 || Kc <= 0 || !A || !B || !dst || !workspace) return LARA2DGS_E_INVALID;
    return gemm_tn(A, N, N, B, Kc, Kc, 1, nullptr, M, dst, (float *)workspace, (hipStream_t)stream);
}

int64_t lara_gemm_tn_workspace_bytes(void) { return (int64_t)TN_PART_BYTES; }

int lara_layernorm256_backward(int32_t rows, const float *dy, const float *x, const float *gamma, float eps,
                               const float *skip, float *dx, float *dgamma, float *dbeta, void *workspace,
                               void *stream) {
    if (rows <= 0 || !dy || !x || !gamma || !dx || !dgamma || !dbeta || !workspace) return LARA2DGS_E_INVALID;
    return ln_bwd(dy, false, x, gamma, eps, skip, dx, nullptr, dgamma, dbeta, nullptr, (float *)workspace, rows, (hipStream_t)stream);
}

int lara_groupattn_core_backward(int32_t G, const uint16_t *q, const uint16_t *kv, const uint16_t *d_o, uint16_t *dq,
                                 uint16_t *dkv, void *stream) {
    if (G < 0) return LARA2DGS_E_INVALID;
    if (G == 0) return LARA2DGS_OK;
    if (!q || !kv || !d_o || !dq || !dkv) return LARA2DGS_E_INVALID;
    hipLaunchKernelGGL(group_attn_bwd_kernel, dim3((G + 1) / 2), dim3(256), 0, (hipStream_t)stream, q, kv, d_o, dq, dkv, G, 512);
    L2D_CHECK_LAUNCH();
    return LARA2DGS_OK;
}

}  // extern "C"
