// micro-benchmark: LDS atomic throughput (float vs int, conflict-free vs strided vs same address)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE, int PATTERN>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    __shared__ float acc[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx;
    if (PATTERN == 0) idx = wave * 64 + lane;                       // conflict free, distinct
    else if (PATTERN == 1) idx = (lane >> 2) * 23 * 5 + (lane & 3) + wave * 2048;  // quad pattern
    else if (PATTERN == 2) idx = wave * 64 + (lane & 31);           // pairs share an address
    else idx = wave;                                                // all lanes same address
    float v = 1.0f + lane * 1e-3f;
    for (int i = 0; i < iters; i++) {
        const int a = (idx + i * 64 * (PATTERN == 3 ? 0 : 1)) & 8191;
        if (MODE == 0) atomicAdd(&acc[a], v);
        else if (MODE == 1) atomicAdd((unsigned *)&acc[a], (unsigned)i);
        else if (MODE == 2) acc[a] += v;                             // plain RMW (racy, for rate only)
        else if (MODE == 3) atomicAdd((unsigned long long *)&acc[a & ~1], (unsigned long long)i);
        else if (MODE == 4) atomicAdd((double *)&acc[a & ~1], (double)v);
        else if (MODE == 5) __hip_atomic_fetch_add(&acc[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 6) v += atomicAdd(&acc[a], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5];
}
template <int MODE, int PATTERN> void run(const char *name) {
    float *out; hipMalloc(&out, 4096 * 4);
    const int iters = 4096, blocks = 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instrs = (double)blocks * 4 * iters;
    // cycles per wave-instruction per CU (4 blocks/CU resident assumed: 1024 blocks / 256 CUs)
    const double cyc = ms * 1e-3 * 2.1e9 / (wave_instrs / 256.0);
    printf("%-28s %8.3f ms  %6.1f CU-cycles per wave-instr (64 lanes)\n", name, ms, cyc);
    hipFree(out);
}
int main() {
    run<0, 0>("f32 add, distinct");
    run<0, 1>("f32 add, quad pattern");
    run<0, 2>("f32 add, 2 lanes/addr");
    run<0, 3>("f32 add, same addr");
    run<1, 0>("u32 add, distinct");
    run<1, 1>("u32 add, quad pattern");
    run<1, 3>("u32 add, same addr");
    run<3, 0>("u64 add, distinct");
    run<2, 0>("plain rmw, distinct");
    run<4, 0>("f64 add, distinct");
    run<4, 3>("f64 add, same addr");
    run<5, 0>("f32 add wg-scope, distinct");
    run<6, 0>("f32 add rtn, distinct");
    return 0;
}
