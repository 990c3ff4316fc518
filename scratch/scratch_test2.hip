#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
struct Big { float a[400]; int n; };
// many live values -> spills; by-value struct argument like the library
__global__ void __launch_bounds__(64) k(Big B, float *out, int n) {
    float v[300];
#pragma unroll
    for (int i = 0; i < 300; i++) v[i] = B.a[i] + threadIdx.x;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 300; i++) v[i] = v[i] * 1.0001f + v[(i + 1) % 300] * 0.0001f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 300; i++) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    Big B; for (int i = 0; i < 400; i++) B.a[i] = i * 0.01f; B.n = 3;
    float *d; hipMalloc(&d, 64 * 92 * 4);
    for (int rep = 0; rep < 20; rep++) {
        hipLaunchKernelGGL(k, dim3(92), dim3(64), 0, 0, B, d, 10);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("rep %d err %s\n", rep, hipGetErrorString(e)); return 1; }
    }
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    // cpu ref for thread 0
    float v[300]; for (int i = 0; i < 300; i++) v[i] = B.a[i];
    for (int it = 0; it < 10; it++) for (int i = 0; i < 300; i++) v[i] = v[i] * 1.0001f + v[(i + 1) % 300] * 0.0001f;
    float s = 0; for (int i = 0; i < 300; i++) s += v[i];
    printf("gpu %f cpu %f\n", h[0], s);
    return 0;
}
