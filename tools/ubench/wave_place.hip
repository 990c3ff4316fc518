// Micro-benchmark (development): does a workgroup's wave w land on SIMD w % 4?  One workgroup per CU; wave roles:
// 'O' = long VALU chain (object chains, 340 instr x 50 sweeps), 'J' = short one (joints, 120 x 50), '-' = exits.
// Build: hipcc -O3 --offload-arch=gfx950 -o wave_place wave_place.hip ; run: ./wave_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const int *roles, float *out, int lds_bytes_marker) {
    extern __shared__ float lds[];
    const int w = threadIdx.x >> 6;
    const int r = roles[w];
    if (r == 0) return;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    const int n = r == 2 ? 340 * 50 : 120 * 50;
    for (int i = 0; i < n / 4; i++) {     // 4 dependent VALU instructions per trip
        a = fmaf(a, b, c); a = fmaf(a, b, c); a = fmaf(a, b, c); a = fmaf(a, b, c);
    }
    if (a == 123.456f) out[threadIdx.x] = a + lds[0];
}
static float run(const char *pattern, int nthreads, size_t lds) {
    int roles[16] = {0};
    for (int i = 0; i < (int)strlen(pattern); i++) roles[i] = pattern[i] == 'O' ? 2 : (pattern[i] == 'J' ? 1 : 0);
    int *d; float *o;
    hipMalloc(&d, sizeof(roles)); hipMalloc(&o, 4096);
    hipMemcpy(d, roles, sizeof(roles), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(256), dim3(nthreads), lds, 0, d, o, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipFree(d); hipFree(o);
    return best;
}
int main() {
    const size_t lds = 160 * 1024;      // one workgroup per CU
    printf("O---            %.1f us (object wave alone)\n", run("O---", 256, lds) * 1e3f);
    printf("J---            %.1f us (joint wave alone)\n", run("J---", 256, lds) * 1e3f);
    printf("OJJJJ           %.1f us (5 waves)\n", run("OJJJJ", 320, lds) * 1e3f);
    printf("OJJJ-J--        %.1f us (8 waves, J5 should share SIMD1)\n", run("OJJJ-J--", 512, lds) * 1e3f);
    printf("JJJJO           %.1f us\n", run("JJJJO", 320, lds) * 1e3f);
    printf("JJJJ (4 x both) %.1f us (today: every wave does both)\n", 0.0f);
    printf("OOOO            %.1f us\n", run("OOOO", 256, lds) * 1e3f);
    printf("OJ (same SIMD?) %.1f us ; O---J %.1f us\n", run("OJ", 128, lds) * 1e3f, run("O---J", 320, lds) * 1e3f);
    return 0;
}
