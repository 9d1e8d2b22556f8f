// micro-benchmark behind DESIGN.md section 3.2.1: what ONE compositing step costs per 64 (pixel, entry) hits in
// three formulations, evaluation of alpha / depth excluded (it is common to all of them):
//   A  lane = pixel, each lane blends its own next hit sequentially (the quad-SIMT walk's blend):
//      T, 3 distortion terms, 10 accumulated channels -> ~20 full-rate FMAs
//   B  lanes = 64 consecutive hits of a pixel-major hit list with ARBITRARY segment boundaries: segmented prefix
//      product of (1 - alpha), segmented prefix sums of the two distortion moments, and a segmented REDUCE of the 10
//      channels (6 steps each)
//   C  lanes = hits, FIXED 4-lane segments (a DPP quad = 4 consecutive hits of one pixel): 2-step quad scans for
//      T / M1 / M2, private per-lane accumulators
// Output: shader clocks per step and per hit, one wave per SIMD and four.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/segscan_step.hip -o tools/ubench/segscan_step
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CTRL>
__device__ __forceinline__ float dpp(float x, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, 0xf, 0xf, false));
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const float *__restrict__ in, float *__restrict__ out, int iters, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    // per-hit inputs (as the evaluation would have produced them)
    float alpha = 0.05f + 0.001f * lane, depth = 1.5f + 0.01f * lane, c[6];
    for (int i = 0; i < 6; i++) c[i] = in[(threadIdx.x + 64 * i) & 1023];
    const bool head = (lane % 23) == 0;            // MODE B: a new pixel starts at this lane
    float T = 1.f, M1 = 0.f, M2 = 0.f, dist = 0.f, D = 0.f, acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        alpha = alpha * 0.999f + 1e-5f;            // (keeps the compiler from hoisting the step)
        const float om = 1.f - alpha;
        const float mm = 100.f / 99.8f * (1.f - 0.2f * __builtin_amdgcn_rcpf(depth));
        if (MODE == 0) {
            const float w = alpha * T, A = 1.f - T;
            dist += (mm * mm * A + M2 - 2.f * mm * M1) * w;
            D += depth * w; M1 += mm * w; M2 += mm * mm * w;
#pragma unroll
            for (int i = 0; i < 6; i++) acc[i] += c[i] * w;
            T *= om;
        } else if (MODE == 1) {
            // segmented inclusive scans over the wave (6 steps), heads cut the propagation
            float p = om, s1, s2;
            bool h = head;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float y = __shfl_up(p, d, 64);
                const bool hy = __shfl_up((int)h, d, 64);
                if (lane >= d && !h) p *= y;
                h = h || (lane >= d && hy);
            }
            const float Tb = T * __shfl_up(p, 1, 64);       // exclusive
            const float w = alpha * Tb;
            s1 = mm * w; s2 = mm * mm * w; h = head;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float y1 = __shfl_up(s1, d, 64), y2 = __shfl_up(s2, d, 64);
                const bool hy = __shfl_up((int)h, d, 64);
                if (lane >= d && !h) { s1 += y1; s2 += y2; }
                h = h || (lane >= d && hy);
            }
            dist += (mm * mm * (1.f - Tb) + (M2 + s2 - mm * mm * w) - 2.f * mm * (M1 + s1 - mm * w)) * w;
            // segmented reduce of the channels (suffix form: every head ends up with its segment's sum)
            float r[8] = {c[0] * w, c[1] * w, c[2] * w, c[3] * w, c[4] * w, c[5] * w, depth * w, dist};
            bool t = head;   // tail flags would mirror; the cost is what matters here
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const bool hy = __shfl_down((int)t, d, 64);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float y = __shfl_down(r[i], d, 64);
                    if (lane + d < 64 && !hy) r[i] += y;
                }
                t = t || hy;
            }
#pragma unroll
            for (int i = 0; i < 6; i++) acc[i] += head ? r[i] : 0.f;
            D += head ? r[6] : 0.f;
            T = p; M1 += s1; M2 += s2;
        } else {
            // quad = 4 consecutive hits of one pixel: exclusive quad scans (2 DPP steps each), private accumulators
            float p = om;
            p *= dpp<0x90>(p, 1.f) * ((lane & 1) ? 1.f : 0.f) + ((lane & 1) ? 0.f : 1.f);     // quad_perm [0,0,1,2] gated
            p *= ((lane & 2) ? dpp<0x40>(p, 1.f) : 1.f);
            const float incl = p, excl = incl * __builtin_amdgcn_rcpf(om);
            const float Tb = T * excl, w = alpha * Tb;
            float s1 = mm * w, s2 = mm * mm * w;
            s1 += (lane & 1) ? dpp<0x90>(s1, 0.f) : 0.f; s2 += (lane & 1) ? dpp<0x90>(s2, 0.f) : 0.f;
            s1 += (lane & 2) ? dpp<0x40>(s1, 0.f) : 0.f; s2 += (lane & 2) ? dpp<0x40>(s2, 0.f) : 0.f;
            dist += (mm * mm * (1.f - Tb) + (M2 + s2 - mm * mm * w) - 2.f * mm * (M1 + s1 - mm * w)) * w;
            D += depth * w;
#pragma unroll
            for (int i = 0; i < 6; i++) acc[i] += c[i] * w;
            T *= dpp<0xFF>(incl, 1.f);               // quad_perm [3,3,3,3]: the quad's total
            M1 += dpp<0xFF>(s1, 0.f); M2 += dpp<0xFF>(s2, 0.f);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = T + M1 + M2 + dist + D;
    for (int i = 0; i < 6; i++) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char *name, int wps) {
    float *in, *out; unsigned long long *cyc;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 4 * 4 * 256 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(in, 0, 4096);
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, 10, cyc);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per SIMD: wps waves share it; time per step per SIMD = ms / (iters * wps)
    const double ns_step = ms * 1e6 / ((double)iters * wps);
    printf("%-34s waves/SIMD %d: one wave %7.1f clk/step | SIMD throughput %7.1f ns per 64 hits = %5.2f ns per hit\n", name, wps,
           (double)c / iters, ns_step, ns_step / 64.0);
    (void)hipFree(in); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; w *= 4) {
        run<0>("A lane=pixel sequential blend", w);
        run<1>("B lanes=hits, general segments", w);
        run<2>("C lanes=hits, 4-lane segments", w);
    }
    return 0;
}
