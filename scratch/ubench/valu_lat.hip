// Micro-benchmark: cycles per instruction of one wave per SIMD for dependent / independent VALU chains and DPP adds.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void __launch_bounds__(64) k(float *out, unsigned long long *cyc, int mode, float a, float b) {
    float x = threadIdx.x * a, y = x + 1, z = x + 2, w = x + 3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; it++) {
        if (mode == 0) { REP64(x = fmaf(x, a, b);) }                                       // dependent fma chain
        else if (mode == 1) { REP64(x = fmaf(x, a, b); y = fmaf(y, a, b); z = fmaf(z, a, b); w = fmaf(w, a, b);) }   // 4 independent chains
        else if (mode == 2) { REP64(x += dpp<0x128>(x);) }                                 // dependent DPP row_ror:8 add
        else if (mode == 3) { REP64(x += dpp<0x128>(x); x += dpp<0x124>(x); x += dpp<0x122>(x); x += dpp<0x121>(x); x = fmaf(x, a, b);) }
        else if (mode == 4) { REP64(x = __builtin_amdgcn_fmed3f(x, a, b); x = x - a;) }
        else if (mode == 5) { REP64(x = (threadIdx.x & 1) ? x * a : y; y = fmaf(x, a, b);) } // cndmask in chain
        else if (mode == 6) { REP64(x += dpp<0x128>(x); y += dpp<0x128>(y);) }              // two independent DPP chains
        else if (mode == 7) { REP64(x += dpp<0xB1>(x);) }       // quad_perm [1,0,3,2]
        else if (mode == 8) { REP64(x += dpp<0x141>(x);) }      // row_half_mirror
        else if (mode == 9) { REP64(x += dpp<0x140>(x);) }      // row_mirror
        else if (mode == 10) { REP64(x += dpp<0x111>(x);) }     // row_shr:1
        else if (mode == 11) { REP64(x += dpp<0x150>(x);) }     // row_newbcast:0
        else if (mode == 12) { REP64(x += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x041F));) }   // ds_swizzle xor 1
        else if (mode == 13) { REP64(x += dpp<0x128>(x); y = fmaf(y, a, b); z = fmaf(z, a, b); w = fmaf(w, a, b);) }   // dpp + 3 independent fma
        else if (mode == 14) { REP64(x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x128, 0xf, 0xf, true));) }  // v_mov_dpp + add
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x + y + z + w;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h[16];
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 128);
    const int ninstr[15] = {64 * 64, 64 * 64 * 4, 64 * 64, 64 * 64 * 5, 64 * 64 * 2, 64 * 64 * 3, 64 * 64 * 2, 4096, 4096, 4096, 4096, 4096, 4096 * 2, 4096 * 4, 4096 * 2};
    const char *name[15] = {"dependent fma", "4 independent fma chains", "dependent dpp add", "4 dpp adds + fma (group_sum)", "med3 + sub", "cmp-free cndmask + mul + fma", "2 independent dpp chains", "quad_perm add", "row_half_mirror add", "row_mirror add", "row_shr:1 add", "row_newbcast add", "ds_swizzle + add", "dpp add + 3 indep fma", "v_mov_dpp + add"};
    for (int waves = 1; waves <= 1; waves++)
        for (int m = 0; m < 15; m++) {
            hipLaunchKernelGGL(k, dim3(1024 * waves), dim3(64), 0, 0, out, cyc, m, 1.0001f, 0.5f);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-32s %6.2f cycles per instruction\n", waves, name[m], (double)h[m] / ninstr[m]);
        }
    return 0;
}
