// Micro-benchmark (round 5): can matrix-core work and vector-ALU work of DIFFERENT waves share a SIMD at full rate each?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/coissue.hip -o tools/ubench/coissue
// VERDICT r4 "next" #3 asks whether an MFMA-bound kernel (the encoder's convolution: MFMA pipe busy ~0.45, VALU ~0.15) and a
// VALU-bound one (composite_bwd: VALU issue 0.82, no MFMA) can be co-resident on the same SIMDs and overlap.  This is the
// hardware-level upper bound of that idea, with nothing else in the way (no memory traffic, no LDS traffic, no barriers):
//   kernel M: every wave issues back-to-back INDEPENDENT v_mfma_f32_32x32x16_bf16 (8 accumulators = 128 registers), wm waves per SIMD
//   kernel V: every wave issues back-to-back independent v_fma_f32 (32 chains), wv waves per SIMD
// both sized to occupy every CU of the device at the stated waves per SIMD (dynamic LDS caps the workgroups per CU so that
// BOTH fit a CU together), timed alone and together on two streams.  together ~ max(alone) -> the pipes are shared by
// time-slicing only at the issue port; together ~ sum -> no co-issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) kM(float *out, int iters) {
    extern __shared__ char lds[];
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 c[8];
    for (int k = 0; k < 8; k++) for (int e = 0; e < 16; e++) c[k][e] = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; k++) for (int e = 0; e < 16; e++) s += c[k][e];
    if (s == 123.456f) { out[0] = s; lds[0] = 1; }
}

__global__ void __launch_bounds__(256) kV(float *out, int iters) {
    extern __shared__ char lds[];
    float x[32];
    for (int k = 0; k < 32; k++) x[k] = 1.0f + 1e-3f * (threadIdx.x + k);
    const float m = 0.9999f, d = 1e-4f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(d));
    }
    float s = 0.f;
    for (int k = 0; k < 32; k++) s += x[k];
    if (s == 123.456f) { out[0] = s; lds[0] = 1; }
}

static float timed(hipStream_t sa, hipStream_t sb, bool runM, bool runV, int wgM, int ldsM, int itM, int wgV, int ldsV, int itV, float *out) {
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    CK(hipStreamWaitEvent(sb, e0, 0));
    if (runM) hipLaunchKernelGGL(kM, dim3(wgM), dim3(256), ldsM, sa, out, itM);
    if (runV) hipLaunchKernelGGL(kV, dim3(wgV), dim3(256), ldsV, sb, out, itV);
    CK(hipEventRecord(e2, sb));
    CK(hipStreamWaitEvent(sa, e2, 0));
    CK(hipEventRecord(e1, sa));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float *out; CK(hipMalloc(&out, 64));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void *)kM, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)kV, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("{\"device\": \"%s\", \"cus\": %d, \"cases\": [\n", p.name, cus);
    // (waves per SIMD of M, of V): a workgroup = 4 waves = one wave per SIMD; LDS per workgroup chosen so that exactly wm (wv)
    // workgroups of a kernel fit a CU next to the other kernel's share
    const int cfg[][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {1, 3}};
    for (unsigned q = 0; q < sizeof(cfg) / sizeof(cfg[0]); q++) {
        const int wm = cfg[q][0], wv = cfg[q][1];
        // shares of the 160 KB: M gets 96 KB, V 64 KB; per workgroup = share / count (minus a little, so that count + 1 do not fit)
        const int ldsM = 96 * 1024 / wm - 512, ldsV = 64 * 1024 / wv - 512;
        const int wgM = cus * wm, wgV = cus * wv;
        const int itM = 20000 / wm, itV = 40000 / wv;      // ~ a few ms each
        float tM = 1e9f, tV = 1e9f, tB = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            const float a = timed(sa, sb, true, false, wgM, ldsM, itM, wgV, ldsV, itV, out);
            const float b = timed(sa, sb, false, true, wgM, ldsM, itM, wgV, ldsV, itV, out);
            const float c = timed(sa, sb, true, true, wgM, ldsM, itM, wgV, ldsV, itV, out);
            if (rep) { tM = a < tM ? a : tM; tV = b < tV ? b : tV; tB = c < tB ? c : tB; }
        }
        const double mfma_tflops = 2.0 * 32 * 32 * 16 * 8.0 * itM * (double)wgM * 4 / (tM * 1e-3) / 1e12;
        const double fma_tflops = 2.0 * 64 * 32.0 * itV * (double)wgV * 4 / (tV * 1e-3) / 1e12;
        printf("  {\"mfma_waves_per_simd\": %d, \"valu_waves_per_simd\": %d, \"mfma_alone_ms\": %.3f, \"valu_alone_ms\": %.3f, \"together_ms\": %.3f, "
               "\"together_over_sum\": %.3f, \"together_over_max\": %.3f, \"mfma_alone_TFLOPs\": %.0f, \"valu_alone_TFLOPs\": %.1f}%s\n",
               wm, wv, tM, tV, tB, tB / (tM + tV), tB / (tM > tV ? tM : tV), mfma_tflops, fma_tflops,
               q + 1 < sizeof(cfg) / sizeof(cfg[0]) ? "," : "");
    }
    printf("]}\n");
    return 0;
}
