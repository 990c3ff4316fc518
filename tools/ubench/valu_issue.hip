// VALU and LDS instruction issue-rate micro-benchmark for the `roofline.valu` peak of bench.py and for
// the LDS instruction costs of DESIGN.md 5 (profiles/r04_valu_issue.txt, r04_valu_lds_issue.txt).  `valu_issue [first mode]`.
// W waves per SIMD (W = 1, 2, 4, 6, 8) on every SIMD of the chip run independent chains of one instruction (or a pattern of two); reported:
//   * cycles per wave-instruction seen by one wave (s_memtime), and the same per SIMD (= wave cycles / W),
//   * chip-wide wave-instructions per second from HIP-event wall time (what bench.py divides SQ_INSTS_VALU by).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_issue.hip -o /tmp/valu_issue ; run on the GPU box.
// A workgroup of 256*W threads puts W waves on each SIMD of its CU (waves w and w+4 share a SIMD, simd_map.hip); 96 KB of
// dynamic LDS keeps a second workgroup off the CU, so "W waves per SIMD" is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP2(x) x x
#define REP4(x) x x x x
#define S8(x) x x x x x x x x      // the 8-instruction body eight times in ONE asm statement (the compiler puts an s_nop between statements)
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

enum { M_FMA = 0, M_PKFMA, M_CVT, M_RCP, M_MULLO, M_ADD64, M_CNDMASK, M_CMP, M_DPPMOV, M_BPERM, M_MIN3, M_MIX, M_FMA_SGPR, M_CNDMASK_SMASK, M_FMA_LIT, M_CNDMASK_AFTER_CMP, M_SALU_MIX, M_ADD_VV, M_MUL_VV, M_FMAC, M_MOV, M_ADD_E64, M_FMA_ADD_ALT, M_CMP_ADD_ALT, M_ADDU32, M_FMA_DISTINCT, M_CVT_ADD_ALT, M_CNDMASK_ADD_ALT, M_DS_READ_U16, M_DS_READ_B32, M_DS_READ_B64, M_DS_READ_B128, M_DS_READ_B32_RAND, M_DS_READ_B128_RAND, M_DS_WRITE_B32, M_DS_WRITE_B128, M_DS_MIN_U64_16, M_DS_ADD_RTN_1, M_DS_SWIZZLE, M_DS_READ_B32_BCAST, M_COUNT };
static const char *mode_name[M_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_i32", "v_rcp_f32", "v_mul_lo_u32", "v_lshl_add_u64",
                                         "v_cndmask_b32", "v_cmp_lt_f32", "v_mov_b32 dpp", "ds_bpermute_b32", "v_min3_f32",
                                         "raster mix (sub,mul,fma,cmp,cndmask)",
                                         "v_fma_f32 with an SGPR operand", "v_cndmask_b32, mask in an SGPR pair (s_mov)", "v_add_f32 with a literal constant",
                                         "v_cmp + v_cndmask pairs", "fma x2 + s_and_b64/s_add pairs (SALU beside VALU)",
                                         "v_add_f32 e32, VGPR operands", "v_mul_f32 e32, VGPR operands", "v_fmac_f32 e32", "v_mov_b32 e32", "v_add_f32 e64", "v_fma_f32 / v_add_f32 alternating", "v_cmp_lt_f32 e32 / v_add_f32 alternating", "v_add_u32 e32", "v_fma_f32, four distinct registers", "v_cvt_f32_i32 / v_add_f32 alternating", "v_cndmask_b32 (SGPR mask) / v_add_f32 alternating",
                                         "ds_read_u16, lane-linear", "ds_read_b32, lane-linear", "ds_read_b64, lane-linear", "ds_read_b128, lane-linear", "ds_read_b32, scattered", "ds_read_b128, scattered (16-byte aligned)", "ds_write_b32, lane-linear", "ds_write_b128, lane-linear", "ds_min_u64, 16 lanes, scattered", "ds_add_rtn_u32, one lane", "ds_swizzle_b32", "ds_read_b32, one address (broadcast)"};
static const int mode_ops[M_COUNT] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};      // instructions per inner block

__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int mode, int iters, float a, float b) {
    extern __shared__ float lds[];
    float x0 = threadIdx.x * a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    // packed pairs / 64-bit values
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    v2f pa = {a, a}, pb = {b, b};
    unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, q4 = q0 + 4, q5 = q0 + 5, q6 = q0 + 6, q7 = q0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const int ia = (int)(a * 3.0f) | 1;
    const int baddr = ((threadIdx.x + 1) & 63) << 2;
    const int lin2 = (threadIdx.x & 63) * 2 + (threadIdx.x >> 6) * 128, lin4 = lin2 * 2, lin8 = lin2 * 4, lin16 = lin2 * 8;
    const int rnd4 = (int)(((threadIdx.x * 2654435761u) >> 8) & 0x1ffc), rnd16 = rnd4 & ~15, rnd8 = rnd4 & ~7;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f w0 = {x0, x1, x2, x3}, w1 = w0, w2 = w0, w3 = w0;
    unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)iters; int scnt = iters;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, independent of the DVFS state
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        switch (mode) {
        case M_FMA:
            REP2(asm volatile(S8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_PKFMA:
            REP2(asm volatile(S8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"): "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));)
            break;
        case M_CVT:
            REP2(asm volatile(S8("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n"
                               "v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
            break;
        case M_RCP:
            REP2(asm volatile(S8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                               "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
            break;
        case M_MULLO:
            REP2(asm volatile(S8("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                               "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"): "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia));)
            break;
        case M_ADD64:
            REP2(asm volatile(S8("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3\n"
                               "v_lshl_add_u64 %4, %4, 1, %4\n v_lshl_add_u64 %5, %5, 1, %5\n v_lshl_add_u64 %6, %6, 1, %6\n v_lshl_add_u64 %7, %7, 1, %7\n"): "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));)
            break;
        case M_CNDMASK:
            REP2(asm volatile(S8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
            break;
        case M_CMP:
            REP2(asm volatile(S8("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                               "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
            break;
        case M_DPPMOV:
            REP2(asm volatile(S8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
            break;
        case M_BPERM:
            REP2(asm volatile(S8("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                               "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(baddr));)
            break;
        case M_MIN3:
            REP2(asm volatile(S8("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                               "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_MIX:     // the instruction mix of a sample-point test: subtract, multiply, fused multiply-add, compare, select
            REP2(asm volatile(S8("v_sub_f32 %0, %1, %8\n v_mul_f32 %2, %0, %9\n v_fma_f32 %3, %2, %8, %0\n v_cmp_lt_f32 vcc, %3, %9\n"
                               "v_cndmask_b32 %4, %5, %6, vcc\n v_sub_f32 %5, %7, %9\n v_fma_f32 %6, %4, %8, %5\n v_mul_f32 %7, %6, %8\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");)
            break;
        case M_FMA_SGPR:
            REP2(asm volatile(S8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));)
            break;
        case M_CNDMASK_SMASK:
            REP2(asm volatile(S8("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                               "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(smask));)
            break;
        case M_FMA_LIT:
            REP2(asm volatile(S8("v_add_f32 %0, 0x3f000123, %0\n v_add_f32 %1, 0x3f000123, %1\n v_add_f32 %2, 0x3f000123, %2\n v_add_f32 %3, 0x3f000123, %3\n"
                               "v_add_f32 %4, 0x3f000123, %4\n v_add_f32 %5, 0x3f000123, %5\n v_add_f32 %6, 0x3f000123, %6\n v_add_f32 %7, 0x3f000123, %7\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
            break;
        case M_CNDMASK_AFTER_CMP:
            REP2(asm volatile(S8("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
            break;
        case M_SALU_MIX:     // 4 VALU + 4 SALU per block of 8 (only the VALU ones would be counted by SQ_INSTS_VALU)
            REP2(asm volatile(S8("v_fma_f32 %0, %0, %8, %9\n s_and_b64 %10, %10, exec\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %11, %11, 1\n"
                               "v_fma_f32 %2, %2, %8, %9\n s_and_b64 %10, %10, exec\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %11, %11, 1\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(smask), "s"(scnt) : "scc");)
            break;
        case M_ADD_VV:
            REP2(asm volatile(S8("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_MUL_VV:
            REP2(asm volatile(S8("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_FMAC:
            REP2(asm volatile(S8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_MOV:
            REP2(asm volatile(S8("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %0\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_ADD_E64:
            REP2(asm volatile(S8("v_add_f32_e64 %0, %8, %0\n v_add_f32_e64 %1, %8, %1\n v_add_f32_e64 %2, %8, %2\n v_add_f32_e64 %3, %8, %3\n v_add_f32_e64 %4, %8, %4\n v_add_f32_e64 %5, %8, %5\n v_add_f32_e64 %6, %8, %6\n v_add_f32_e64 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_FMA_ADD_ALT:
            REP2(asm volatile(S8("v_fma_f32 %0, %0, %8, %9\n v_add_f32_e32 %1, %8, %1\n v_fma_f32 %2, %2, %8, %9\n v_add_f32_e32 %3, %8, %3\n v_fma_f32 %4, %4, %8, %9\n v_add_f32_e32 %5, %8, %5\n v_fma_f32 %6, %6, %8, %9\n v_add_f32_e32 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_CMP_ADD_ALT:
            REP2(asm volatile(S8("v_cmp_lt_f32_e32 vcc, %0, %8\n v_add_f32_e32 %1, %8, %1\n v_cmp_lt_f32_e32 vcc, %2, %8\n v_add_f32_e32 %3, %8, %3\n v_cmp_lt_f32_e32 vcc, %4, %8\n v_add_f32_e32 %5, %8, %5\n v_cmp_lt_f32_e32 vcc, %6, %8\n v_add_f32_e32 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");)
            break;
        case M_ADDU32:
            REP2(asm volatile(S8("v_add_u32_e32 %0, %8, %0\n v_add_u32_e32 %1, %8, %1\n v_add_u32_e32 %2, %8, %2\n v_add_u32_e32 %3, %8, %3\n v_add_u32_e32 %4, %8, %4\n v_add_u32_e32 %5, %8, %5\n v_add_u32_e32 %6, %8, %6\n v_add_u32_e32 %7, %8, %7\n "): "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia), "v"(b));)
            break;
        case M_FMA_DISTINCT:
            REP2(asm volatile(S8("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %4\n v_fma_f32 %2, %3, %4, %5\n v_fma_f32 %3, %4, %5, %6\n v_fma_f32 %4, %5, %6, %7\n v_fma_f32 %5, %6, %7, %0\n v_fma_f32 %6, %7, %0, %1\n v_fma_f32 %7, %0, %1, %2\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_CVT_ADD_ALT:
            REP2(asm volatile(S8("v_cvt_f32_i32_e32 %0, %0\n v_add_f32_e32 %1, %8, %1\n v_cvt_f32_i32_e32 %2, %2\n v_add_f32_e32 %3, %8, %3\n v_cvt_f32_i32_e32 %4, %4\n v_add_f32_e32 %5, %8, %5\n v_cvt_f32_i32_e32 %6, %6\n v_add_f32_e32 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
            break;
        case M_CNDMASK_ADD_ALT:
            REP2(asm volatile(S8("v_cndmask_b32_e64 %0, %0, %8, %9\n v_add_f32_e32 %1, %8, %1\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_add_f32_e32 %3, %8, %3\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_add_f32_e32 %5, %8, %5\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_add_f32_e32 %7, %8, %7\n "): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(smask));)
            break;
        case M_DS_READ_U16:
            REP2(asm volatile(S8("ds_read_u16 %0, %8\n ds_read_u16 %1, %8\n ds_read_u16 %2, %8\n ds_read_u16 %3, %8\n ds_read_u16 %4, %8\n ds_read_u16 %5, %8\n ds_read_u16 %6, %8\n ds_read_u16 %7, %8\n  s_waitcnt lgkmcnt(0)\n"): "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(lin2));)
            break;
        case M_DS_READ_B32:
            REP2(asm volatile(S8("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n  s_waitcnt lgkmcnt(0)\n"): "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(lin4));)
            break;
        case M_DS_READ_B64:
            REP2(asm volatile(S8("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n  s_waitcnt lgkmcnt(0)\n"): "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(lin8));)
            break;
        case M_DS_READ_B128:
            REP2(asm volatile(S8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n  s_waitcnt lgkmcnt(0)\n"): "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(lin16));)
            break;
        case M_DS_READ_B32_RAND:
            REP2(asm volatile(S8("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n  s_waitcnt lgkmcnt(0)\n"): "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(rnd4));)
            break;
        case M_DS_READ_B128_RAND:
            REP2(asm volatile(S8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n  s_waitcnt lgkmcnt(0)\n"): "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(rnd16));)
            break;
        case M_DS_WRITE_B32:
            REP2(asm volatile(S8("ds_write_b32 %8, %0\n ds_write_b32 %8, %1\n ds_write_b32 %8, %2\n ds_write_b32 %8, %3\n ds_write_b32 %8, %4\n ds_write_b32 %8, %5\n ds_write_b32 %8, %6\n ds_write_b32 %8, %7\n  s_waitcnt lgkmcnt(0)\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(lin4) : "memory");)
            break;
        case M_DS_WRITE_B128:
            REP2(asm volatile(S8("ds_write_b128 %4, %0\n ds_write_b128 %4, %1\n ds_write_b128 %4, %2\n ds_write_b128 %4, %3\n ds_write_b128 %4, %0\n ds_write_b128 %4, %1\n ds_write_b128 %4, %2\n ds_write_b128 %4, %3\n  s_waitcnt lgkmcnt(0)\n"): "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(lin16) : "memory");)
            break;
        case M_DS_MIN_U64_16:
            REP2(asm volatile("s_mov_b32 exec_lo, 0x11111111\n s_mov_b32 exec_hi, 0x11111111\n" S8("ds_min_u64 %8, %0\n ds_min_u64 %8, %1\n ds_min_u64 %8, %2\n ds_min_u64 %8, %3\n ds_min_u64 %8, %4\n ds_min_u64 %8, %5\n ds_min_u64 %8, %6\n ds_min_u64 %8, %7\n  s_waitcnt lgkmcnt(0)\n") "s_mov_b64 exec, -1\n": "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(rnd8) : "memory");)
            break;
        case M_DS_ADD_RTN_1:
            REP2(asm volatile("s_mov_b64 exec, 1\n" S8("ds_add_rtn_u32 %0, %8, %9\n ds_add_rtn_u32 %1, %8, %9\n ds_add_rtn_u32 %2, %8, %9\n ds_add_rtn_u32 %3, %8, %9\n ds_add_rtn_u32 %4, %8, %9\n ds_add_rtn_u32 %5, %8, %9\n ds_add_rtn_u32 %6, %8, %9\n ds_add_rtn_u32 %7, %8, %9\n  s_waitcnt lgkmcnt(0)\n") "s_mov_b64 exec, -1\n": "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7) : "v"(lin4), "v"(ia) : "memory");)
            break;
        case M_DS_SWIZZLE:
            REP2(asm volatile(S8("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %1, %1 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %2, %2 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %3, %3 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %4, %4 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %5, %5 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %6, %6 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %7, %7 offset:swizzle(SWAP,1)\n  s_waitcnt lgkmcnt(0)\n"): "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(lin4));)
            break;
        case M_DS_READ_B32_BCAST:
            REP2(asm volatile(S8("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n  s_waitcnt lgkmcnt(0)\n"): "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(0));)
            break;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
              (float)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7) + (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + w0.x + w1.y + w2.z + w3.w;
    if (r == 12345.678f) lds[threadIdx.x] = r;      // (keeps the LDS allocation and every chain alive)
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *c = cyc + 3 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        c[0] = t1 - t0; c[1] = r0; c[2] = r1;
    }
}

int main(int argc, char **argv) {
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    ncu = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clock %d kHz (reported)\n", prop.name, ncu, prop.clockRate);
    const int iters = 400;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)ncu * 2048 * 4); hipMalloc(&cyc, (size_t)ncu * 32 * 3 * 8);
    const size_t lds_bytes = 96 * 1024;      // one workgroup per CU
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // (threads per workgroup, workgroups per CU, LDS per workgroup): W waves per SIMD; the LDS sizes keep one more workgroup off the CU
    struct Shape { int W, threads, per_cu, lds_kb; } shapes[] = {{1, 256, 1, 96}, {2, 512, 1, 96}, {4, 1024, 1, 96}, {6, 512, 3, 48}, {8, 512, 4, 39}};
    printf("%-40s %3s %14s %14s %14s %9s %10s\n", "instruction", "W", "cyc/instr/wave", "cyc/instr/SIMD", "G wave-instr/s", "clock GHz", "concurrent");
    for (int m = (argc > 1 ? atoi(argv[1]) : 0); m < M_COUNT; m++)
        for (const Shape &sh : shapes) {
            const int blocks = sh.per_cu * ncu, nw = blocks * (sh.threads / 64);
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {      // the first repetitions warm the clocks
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(sh.threads), (size_t)sh.lds_kb * 1024, 0, out, cyc, m, iters, 1.0001f, 0.5f);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            std::vector<unsigned long long> h((size_t)nw * 3);
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> c(nw), d(nw);
            unsigned long long rmin = ~0ull, rmax = 0;
            for (int i = 0; i < nw; i++) { c[i] = (double)h[3 * i]; d[i] = (double)(h[3 * i + 2] - h[3 * i + 1]); rmin = std::min(rmin, h[3 * i + 1]); rmax = std::max(rmax, h[3 * i + 2]); }
            std::sort(c.begin(), c.end()); std::sort(d.begin(), d.end());
            const double ninstr = (double)iters * 16 * mode_ops[m];                 // per wave
            const double medc = c[nw / 2], medd = d[nw / 2];                        // shader cycles / 100 MHz ticks of the median wave
            // rate: all waves' instructions over the span from the first wave's start to the last wave's end (device clock, no
            // launch overhead); "concurrent" = median wave duration / that span: 1.0 when every wave ran all the time
            const double span_s = (double)(rmax - rmin) / 100e6;
            printf("%-40s %3d %14.2f %14.2f %14.1f %9.3f %10.2f\n", mode_name[m], sh.W, medc / ninstr, medc / ninstr / sh.W, ninstr * nw / span_s / 1e9,
                   medc / (medd / 100e6) / 1e9, medd / (double)(rmax - rmin));
        }
    return 0;
}
