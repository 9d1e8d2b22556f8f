// What does ds_read_b64_tr_b16 deliver?  LDS holds u16 values equal to their own index; every lane passes
// the byte address 8 * lane (pattern 0) or a scattered one (pattern 1) and prints the four values it gets.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void k(unsigned short *out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // pattern 1: lane l of a 16-lane group -> row (l >> 2) of 64 elements, chunk (l & 3); groups 1 KB apart
    const int elem = pattern == 0 ? 4 * lane : (lane >> 4) * 512 + ((lane & 15) >> 2) * 64 + (lane & 3) * 4;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(lds + elem));
    union { bf16x4 v; unsigned short u[4]; } c; c.v = v;
    for (int e = 0; e < 4; e++) out[pattern * 256 + lane * 4 + e] = c.u[e];
}
int main() {
    unsigned short *d, h[512];
    hipMalloc(&d, sizeof(h));
    for (int p = 0; p < 2; p++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int p = 0; p < 2; p++) {
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; l++) printf("lane %2d: %4d %4d %4d %4d%s", l, h[p*256+l*4], h[p*256+l*4+1], h[p*256+l*4+2], h[p*256+l*4+3], (l & 1) ? "\n" : "   ");
    }
    return 0;
}
