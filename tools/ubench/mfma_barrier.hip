// How much of the matrix-core rate survives a workgroup barrier every N MFMAs?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/ubench/mfma_barrier.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NW, int PER_BARRIER, bool BARRIER>
__global__ void __launch_bounds__(NW * 64) k(float *out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < PER_BARRIER; m++) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NW, int PB, bool BAR>
void run(const char *name, int wgs_per_cu) {
    float *out; hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    const int grid = 256 * wgs_per_cu, iters = 16384 / PB * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NW, PB, BAR>), dim3(grid), dim3(NW * 64), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NW, PB, BAR>), dim3(grid), dim3(NW * 64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * NW * iters * PB * 32768.0;
    printf("%-44s %8.1f TFLOP/s\n", name, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run<4, 16, false>("4 waves/CU, no barrier", 1);
    run<8, 16, false>("8 waves/CU, no barrier", 1);
    run<8, 16, true>("8 waves/CU, barrier per 16 MFMAs", 1);
    run<8, 32, true>("8 waves/CU, barrier per 32 MFMAs", 1);
    run<8, 64, true>("8 waves/CU, barrier per 64 MFMAs", 1);
    run<4, 16, true>("2 x 4 waves/CU, barrier per 16 MFMAs", 2);
    run<4, 32, true>("2 x 4 waves/CU, barrier per 32 MFMAs", 2);
    return 0;
}
