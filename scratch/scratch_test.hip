#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(int *out, int n) {
    int a[300];
    for (int i = 0; i < 300; i++) a[i] = i * (threadIdx.x + 1);
    int s = 0;
    for (int i = 0; i < n; i++) s += a[(i * 7 + threadIdx.x) % 300];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    int *d; hipMalloc(&d, 64 * 92 * 4);
    for (int rep = 0; rep < 50; rep++) {
        hipLaunchKernelGGL(k, dim3(92), dim3(64), 0, 0, d, 300);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("rep %d err %s\n", rep, hipGetErrorString(e)); return 1; }
    }
    int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    long ref = 0; for (int i = 0; i < 300; i++) ref += ((i * 7) % 300) * 1;
    printf("ok h[0]=%d ref=%ld\n", h[0], ref);
    return 0;
}
