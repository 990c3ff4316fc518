// Micro-benchmark (development): which SIMD does wave w of a workgroup land on?  HW_ID bits [5:4] = SIMD, [11:8] = CU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    extern __shared__ float lds[];
    const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
    __syncthreads();
}
int main() {
    unsigned *d, h[16 * 8];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nt : {320, 384, 512}) {
        hipMemset(d, 0xff, sizeof(h));
        hipLaunchKernelGGL(k, dim3(8), dim3(nt), 160 * 1024, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%d threads:", nt);
        for (int b = 0; b < 3; b++) { printf("  wg%d simd:", b); for (int w = 0; w < nt / 64; w++) printf(" %u", (h[b * 16 + w] >> 4) & 3); }
        printf("\n");
    }
    return 0;
}
