// micro-benchmark: native ds_add_f32 in the composite backward's access pattern (16 DPP quads of a wave park 4 consecutive
// floats each in a 96-byte slot), against the plain ds_write_b32 it would replace, with 1 / 2 / 4 / 16 quads of the wave
// landing on the SAME slot in one instruction; and a determinism probe: is the order in which the lanes of ONE wave
// instruction are added to one address fixed (run to run, workgroup to workgroup, under LDS traffic of the other waves)?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_fadd.hip -o /tmp/lds_fadd && /tmp/lds_fadd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

constexpr int SLOTS = 384, SLOT_F = 24;

// SHARE = quads of a wave that hit the same slot; MODE 0 = ds_write_b32, 1 = ds_add_f32
template <int MODE, int SHARE>
__global__ void __launch_bounds__(256) rate_kernel(float *out, int iters) {
    __shared__ float pool[SLOTS * SLOT_F];
    for (int i = threadIdx.x; i < SLOTS * SLOT_F; i += 256) pool[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, quad = lane >> 2;
    float v = 1.0f + lane * 1e-3f;
    unsigned h = (quad / SHARE) * 2654435761u + wave * 40503u;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        float *ps = pool + ((h >> 8) % SLOTS) * SLOT_F + (lane & 3);
        if (MODE == 0) {
            ps[0] = v; ps[4] = v; ps[8] = v; ps[12] = v; ps[16] = v;
            if (!(lane & 2)) ps[20] = v;
        } else {
            atomicAdd(ps, v); atomicAdd(ps + 4, v); atomicAdd(ps + 8, v); atomicAdd(ps + 12, v); atomicAdd(ps + 16, v);
            if (!(lane & 2)) atomicAdd(ps + 20, v);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = pool[5];
}

template <int MODE, int SHARE>
void run_rate(const char *name) {
    float *out;
    hipMalloc(&out, 4096 * 4);
    const int iters = 2048, blocks = 768 * 4;      // 3 workgroups per CU resident (48 KB of LDS each would allow it), 4 rounds
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((rate_kernel<MODE, SHARE>), dim3(blocks), dim3(256), 0, 0, out, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL((rate_kernel<MODE, SHARE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double trips_per_cu = (double)blocks * 4 * iters / 256.0;     // wave-trips (6 LDS instructions each) per CU
    printf("%-44s %8.3f ms  %7.1f CU-clocks (2.4 GHz) per wave-trip of 6 LDS instructions\n", name, ms, ms * 1e-3 * 2.4e9 / trips_per_cu);
    hipFree(out);
}

// determinism: every workgroup adds the same values; lanes of one instruction collide on a few addresses; waves 1-3 keep the
// LDS busy with unrelated traffic of data-dependent length.  All workgroups of all launches must agree bit for bit.
__global__ void __launch_bounds__(256) det_kernel(float *out, const float *vals, int iters, unsigned seed) {
    __shared__ float acc[64];
    __shared__ float noise[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) noise[i] = 0.f;
    if (threadIdx.x < 64) acc[threadIdx.x] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        for (int i = 0; i < iters; i++) {
            const float v = vals[(i * 64 + lane) & 4095];
            atomicAdd(&acc[(lane & 3) + 4 * ((lane >> 2) % 3)], v);       // 16 quads onto 3 slots: 5-6 lanes per address
        }
    } else {
        unsigned h = seed * 747796405u + blockIdx.x * 2891336453u + threadIdx.x;
        const int n = iters * (1 + (int)((seed + blockIdx.x + wave) % 3));
        for (int i = 0; i < n; i++) {
            h = h * 1664525u + 1013904223u;
            atomicAdd(&noise[(h >> 10) & 4095], 1.0f);
        }
    }
    __syncthreads();
    if (threadIdx.x < 12) out[blockIdx.x * 12 + threadIdx.x] = acc[threadIdx.x];
    if (threadIdx.x == 255) out[blockIdx.x * 12] += 0.f * noise[7];
}

int main() {
    run_rate<0, 1>("ds_write_b32, every quad its own slot");
    run_rate<1, 1>("ds_add_f32,   every quad its own slot");
    run_rate<1, 2>("ds_add_f32,   2 quads per slot");
    run_rate<1, 4>("ds_add_f32,   4 quads per slot");
    run_rate<1, 16>("ds_add_f32,   all 16 quads on one slot");
    run_rate<0, 16>("ds_write_b32, all 16 quads on one slot");

    const int blocks = 2048, iters = 512, launches = 8;
    float *vals, *out;
    hipMalloc(&vals, 4096 * 4);
    hipMalloc(&out, (size_t)blocks * 12 * 4);
    float hv[4096];
    srand(7);
    for (int i = 0; i < 4096; i++) hv[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);   // mixed magnitudes and signs
    hipMemcpy(vals, hv, sizeof hv, hipMemcpyHostToDevice);
    float *res = (float *)malloc((size_t)blocks * 12 * 4), ref[12];
    long long differing = 0;
    for (int l = 0; l < launches; l++) {
        hipLaunchKernelGGL(det_kernel, dim3(blocks), dim3(256), 0, 0, out, vals, iters, (unsigned)l);
        hipMemcpy(res, out, (size_t)blocks * 12 * 4, hipMemcpyDeviceToHost);
        if (l == 0) memcpy(ref, res, sizeof ref);
        for (int b = 0; b < blocks; b++) differing += memcmp(res + b * 12, ref, sizeof ref) != 0;
    }
    // the same sums in lane order and in reverse lane order on the host: which one (if either) is the hardware's?
    float fwd[12] = {0}, rev[12] = {0};
    for (int i = 0; i < iters; i++) {
        for (int lane = 0; lane < 64; lane++) fwd[(lane & 3) + 4 * ((lane >> 2) % 3)] += hv[(i * 64 + lane) & 4095];
        for (int lane = 63; lane >= 0; lane--) rev[(lane & 3) + 4 * ((lane >> 2) % 3)] += hv[(i * 64 + lane) & 4095];
    }
    printf("determinism: %lld of %d workgroup results differ from the first (%d launches x %d workgroups, %d colliding instructions each)\n",
           differing, launches * blocks, launches, blocks, iters);
    printf("order: matches ascending-lane host sum: %s, descending-lane: %s  (gpu %.9g, asc %.9g, desc %.9g)\n",
           memcmp(ref, fwd, sizeof ref) ? "no" : "yes", memcmp(ref, rev, sizeof ref) ? "no" : "yes", ref[0], fwd[0], rev[0]);
    return 0;
}
