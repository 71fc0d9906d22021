// VALU issue-rate and IEEE conformance microbenchmark for gfx950.
// Decides how the closest-hit scan is written (plain vs packed f32 ops, SGPR operands) and
// checks that device division / sqrt are correctly rounded (needed for bit-exact hit t).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 65536;

#define REP8(X) X X X X X X X X

// mode 0: v_fma_f32, 8 independent chains
// mode 1: v_pk_fma_f32
// mode 2: v_mul_f32 + v_add_f32 alternating
// mode 3: v_pk_mul_f32 + v_pk_add_f32 alternating
// mode 4: v_fma_f32 with an SGPR operand
// mode 5: v_cmp_ge_f32 + v_fma
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed, float sc, unsigned long long* stamps)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __shared__ float lds[256];
    lds[threadIdx.x] = 1.0f - 1e-6f * (threadIdx.x & 3);
    __syncthreads();
    float a = seed + threadIdx.x * 1e-3f, b = 1.0f - 1e-6f;
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    float2v p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    float2v pa = {a, a}, pb = {b, b};
    unsigned long long acc = 0;
    if (MODE == 29 || MODE == 38) asm volatile("s_mov_b64 vcc, 0x55555555" ::: "vcc");
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b), "v"(a));
        } else if (MODE == 1) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                : "v"(pb), "v"(pa));
        } else if (MODE == 2) {
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b), "v"(a));
        } else if (MODE == 3) {
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                : "v"(pb), "v"(pa));
        } else if (MODE == 4) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "s"(sc), "v"(a));
        } else if (MODE == 5) {
            unsigned long long m0, m1, m2, m3;
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_cmp_ge_f32 %4, %0, %10\n v_fma_f32 %1, %1, %8, %9\n v_cmp_ge_f32 %5, %1, %10\n"
                "v_fma_f32 %2, %2, %8, %9\n v_cmp_ge_f32 %6, %2, %10\n v_fma_f32 %3, %3, %8, %9\n v_cmp_ge_f32 %7, %3, %10\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3)
                : "v"(b), "v"(a), "s"(sc));
            acc |= (m0 | m1) | (m2 | m3);
        }
        else if (MODE == 13) {  // VOPC e32 -> vcc only
            asm volatile(
                "v_cmp_ge_f32_e32 vcc, %0, %1\n v_cmp_ge_f32_e32 vcc, %1, %2\n v_cmp_ge_f32_e32 vcc, %2, %3\n v_cmp_ge_f32_e32 vcc, %3, %0\n"
                "v_cmp_ge_f32_e32 vcc, %0, %2\n v_cmp_ge_f32_e32 vcc, %1, %3\n v_cmp_ge_f32_e32 vcc, %2, %0\n v_cmp_ge_f32_e32 vcc, %3, %1\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "vcc");
        } else if (MODE == 14) {  // e64 compare with |abs| modifier, VGPR operands
            unsigned long long m0, m1, m2, m3;
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_cmp_ge_f32_e64 %4, |%0|, %1\n v_fma_f32 %1, %1, %8, %9\n v_cmp_ge_f32_e64 %5, |%1|, %2\n"
                "v_fma_f32 %2, %2, %8, %9\n v_cmp_ge_f32_e64 %6, |%2|, %3\n v_fma_f32 %3, %3, %8, %9\n v_cmp_ge_f32_e64 %7, |%3|, %0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3)
                : "v"(b), "v"(a));
            acc |= (m0 | m1) | (m2 | m3);
        } else if (MODE == 15) {  // v_max3_f32
            asm volatile(
                "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b), "v"(a));
        } else if (MODE == 16) {  // v_pk_mul_f32 alone
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                : "v"(pb));
        } else if (MODE == 17) {  // e64 compare, all-VGPR sources, SGPR-pair dest (what the kernel uses)
            unsigned long long m0, m1, m2, m3;
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_cmp_ge_f32_e64 %4, %0, %1\n v_fma_f32 %1, %1, %8, %9\n v_cmp_ge_f32_e64 %5, %1, %2\n"
                "v_fma_f32 %2, %2, %8, %9\n v_cmp_ge_f32_e64 %6, %2, %3\n v_fma_f32 %3, %3, %8, %9\n v_cmp_ge_f32_e64 %7, %3, %0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3)
                : "v"(b), "v"(a));
            acc |= (m0 | m1) | (m2 | m3);
        } else if (MODE == 18) {  // compare only, e64
            unsigned long long m0, m1, m2, m3;
            asm volatile(
                "v_cmp_ge_f32_e64 %4, %0, %1\n v_cmp_ge_f32_e64 %5, %1, %2\n v_cmp_ge_f32_e64 %6, %2, %3\n v_cmp_ge_f32_e64 %7, %3, %0\n"
                "v_cmp_ge_f32_e64 %4, %0, %2\n v_cmp_ge_f32_e64 %5, %1, %3\n v_cmp_ge_f32_e64 %6, %2, %0\n v_cmp_ge_f32_e64 %7, %3, %1\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3)
                : "v"(b), "v"(a));
            acc |= (m0 | m1) | (m2 | m3);
        }
        else if (MODE == 19) {  // VOPC e32 -> vcc, s_or into an accumulator after each
            unsigned long long m0 = 0;
            asm volatile(
                "v_cmp_ge_f32_e32 vcc, %0, %1\n s_or_b64 %4, %4, vcc\n v_cmp_ge_f32_e32 vcc, %1, %2\n s_or_b64 %4, %4, vcc\n"
                "v_cmp_ge_f32_e32 vcc, %2, %3\n s_or_b64 %4, %4, vcc\n v_cmp_ge_f32_e32 vcc, %3, %0\n s_or_b64 %4, %4, vcc\n"
                "v_cmp_ge_f32_e32 vcc, %0, %2\n s_or_b64 %4, %4, vcc\n v_cmp_ge_f32_e32 vcc, %1, %3\n s_or_b64 %4, %4, vcc\n"
                "v_cmp_ge_f32_e32 vcc, %2, %0\n s_or_b64 %4, %4, vcc\n v_cmp_ge_f32_e32 vcc, %3, %1\n s_or_b64 %4, %4, vcc\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(m0) : : "vcc");
            acc |= m0;
        } else if (MODE == 20) {  // integer compare e64
            unsigned long long m0, m1, m2, m3;
            asm volatile(
                "v_cmp_ge_u32_e64 %4, %0, %1\n v_cmp_ge_u32_e64 %5, %1, %2\n v_cmp_ge_u32_e64 %6, %2, %3\n v_cmp_ge_u32_e64 %7, %3, %0\n"
                "v_cmp_ge_u32_e64 %4, %0, %2\n v_cmp_ge_u32_e64 %5, %1, %3\n v_cmp_ge_u32_e64 %6, %2, %0\n v_cmp_ge_u32_e64 %7, %3, %1\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3));
            acc |= (m0 | m1) | (m2 | m3);
        } else if (MODE == 21) {  // v_max_f32 VOP2 with abs needs VOP3: plain VOP2 max
            asm volatile(
                "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        } else if (MODE == 22) {  // v_max_f32 e64 with |abs| on both
            asm volatile(
                "v_max_f32_e64 %0, |%0|, |%8|\n v_max_f32_e64 %1, |%1|, |%8|\n v_max_f32_e64 %2, |%2|, |%8|\n v_max_f32_e64 %3, |%3|, |%8|\n"
                "v_max_f32_e64 %4, |%4|, |%8|\n v_max_f32_e64 %5, |%5|, |%8|\n v_max_f32_e64 %6, |%6|, |%8|\n v_max_f32_e64 %7, |%7|, |%8|\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        } else if (MODE == 23) {  // v_and_b32 / v_or3_b32
            asm volatile(
                "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n"
                "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b), "v"(a));
        }

        else if (MODE == 24) {  // transcendental: v_rcp_f32
            asm volatile(
                "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 25) {  // transcendental: v_sqrt_f32 / v_rsq_f32
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_rsq_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                "v_sqrt_f32 %4, %4\n v_rsq_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 26) {  // INT32: v_add_u32 / v_lshlrev_b32 / v_and_b32
            asm volatile(
                "v_add_u32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_add_u32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                "v_add_u32 %4, %4, %8\n v_lshlrev_b32 %5, 1, %5\n v_add_u32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        } else if (MODE == 27) {  // INT32 multiply: v_mul_lo_u32 / v_mul_hi_u32
            asm volatile(
                "v_mul_lo_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                "v_mul_lo_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        } else if (MODE == 28) {  // CVT: v_cvt_f32_u32 / v_cvt_u32_f32
            asm volatile(
                "v_cvt_f32_u32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_u32_f32 %3, %3\n"
                "v_cvt_f32_u32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_u32_f32 %7, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 29) {  // v_cndmask_b32 on vcc
            asm volatile(
                "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b) : "vcc");
        } else if (MODE == 30) {  // v_mov_b32
            asm volatile(
                "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 31) {  // INT64-ish: v_lshlrev_b64 on pairs
            asm volatile(
                "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                "v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
        } else if (MODE == 32) {  // v_mbcnt_lo / v_mbcnt_hi
            asm volatile(
                "v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_hi_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_hi_u32_b32 %3, -1, %3\n"
                "v_mbcnt_lo_u32_b32 %4, -1, %4\n v_mbcnt_hi_u32_b32 %5, -1, %5\n v_mbcnt_lo_u32_b32 %6, -1, %6\n v_mbcnt_hi_u32_b32 %7, -1, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 33) {  // v_sub_f32 (ADD class) alone
            asm volatile(
                "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        } else if (MODE == 34) {  // v_mul_f32 (MUL class) alone
            asm volatile(
                "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        } else if (MODE == 35) {  // v_readlane_b32 into SGPRs (spilled-scalar reloads look like this)
            unsigned m0, m1, m2, m3;
            asm volatile(
                "v_readlane_b32 %4, %0, 1\n v_readlane_b32 %5, %1, 2\n v_readlane_b32 %6, %2, 3\n v_readlane_b32 %7, %3, 4\n"
                "v_readlane_b32 %4, %0, 5\n v_readlane_b32 %5, %1, 6\n v_readlane_b32 %6, %2, 7\n v_readlane_b32 %7, %3, 8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3));
            acc |= (m0 | m1) | (m2 | m3);
        }

        else if (MODE == 36) {  // v_cndmask_b32 e64 with an SGPR-pair mask
            unsigned long long msk = 0x5555555555555555ull;
            asm volatile(
                "v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b), "s"(msk));
        } else if (MODE == 37) {  // v_cmp_e32 -> vcc then v_cndmask on it (the compiler's usual select)
            asm volatile(
                "v_cmp_ge_f32_e32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_ge_f32_e32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                "v_cmp_ge_f32_e32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_ge_f32_e32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b) : "vcc");
        } else if (MODE == 38) {  // v_cndmask_b32 vcc, vcc defined once before the loop, different source / destination registers
            asm volatile(
                "v_cndmask_b32 %0, %1, %8, vcc\n v_cndmask_b32 %1, %2, %8, vcc\n v_cndmask_b32 %2, %3, %8, vcc\n v_cndmask_b32 %3, %4, %8, vcc\n"
                "v_cndmask_b32 %4, %5, %8, vcc\n v_cndmask_b32 %5, %6, %8, vcc\n v_cndmask_b32 %6, %7, %8, vcc\n v_cndmask_b32 %7, %0, %8, vcc\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b) : "vcc");
        } else if (MODE == 39) {  // v_min_f32 / v_max_f32 (not FMA / ADD / MUL class)
            asm volatile(
                "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(b));
        }
        else if (MODE == 6) {  // SGPR in src0 of v_fma
            asm volatile(
                "v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
                "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "s"(sc), "v"(a));
        } else if (MODE == 7) {  // SGPR in src2 of v_fma
            asm volatile(
                "v_fma_f32 %0, %0, %9, %8\n v_fma_f32 %1, %1, %9, %8\n v_fma_f32 %2, %2, %9, %8\n v_fma_f32 %3, %3, %9, %8\n"
                "v_fma_f32 %4, %4, %9, %8\n v_fma_f32 %5, %5, %9, %8\n v_fma_f32 %6, %6, %9, %8\n v_fma_f32 %7, %7, %9, %8\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "s"(sc), "v"(b));
        } else if (MODE == 8) {  // VOP2 v_fmac_f32 with SGPR src0: x += s*v
            asm volatile(
                "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "s"(sc), "v"(b));
        } else if (MODE == 9) {  // VOP2 v_fmac_f32 all VGPR
            asm volatile(
                "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(a), "v"(b));
        } else if (MODE == 10) {  // VOP2 v_sub_f32 / v_mul_f32 with SGPR src0
            asm volatile(
                "v_sub_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                "v_sub_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "s"(sc));
        } else if (MODE == 11) {  // 1 broadcast ds_read_b128 per 16 v_fma (LDS-fed operands)
            float4 s4 = *(const float4*)&lds[(i & 63) * 4];
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %10, %11\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %10, %11\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %10, %11\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %10, %11\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                : "v"(s4.x), "v"(s4.y), "v"(s4.z), "v"(s4.w));
        } else if (MODE == 12) {  // v_fma with v_cmp every 8 (the cull shape: 7 fma-class + 1 cmp)
            unsigned long long m0;
            asm volatile(
                "v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n"
                "v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n v_cmp_ge_f32 %8, %7, %11\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=s"(m0)
                : "v"(b), "v"(a), "s"(sc));
            acc |= m0;
        }

    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    float2v q = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    r += q.x + q.y + (float)(acc & 1);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    if (r == 12345.678f) out[0] = r;  // never true in practice; keeps everything live
}

template <int MODE>
double run_rate(const char* name, float* dout, int n_cu, double lane_ops_per_instr)
{
    static unsigned long long* dstamps = nullptr;
    if (!dstamps) CK(hipMalloc(&dstamps, 16));
    const int blocks = n_cu * 8;  // 8 blocks x 4 waves = 8 waves/SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<blocks, 256>>>(dout, 1.0f, 0.999f, dstamps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) rate_kernel<MODE><<<blocks, 256>>>(dout, 1.0f, 0.999f, dstamps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double s = ms * 1e-3 / reps;
    double wave_instr, per_simd_per_s;
    unsigned long long st[2];
    CK(hipMemcpy(st, dstamps, 16, hipMemcpyDeviceToHost));
    double ghz = (double)st[0] / (double)st[1] * 0.1;
    double n_instr = (MODE == 11) ? 16.0 : 8.0;
    wave_instr = (double)blocks * 4 * ITERS * n_instr;
    per_simd_per_s = wave_instr / (n_cu * 4.0) / s;
    double cyc = ghz * 1e9 / per_simd_per_s;
    double tflops = wave_instr * 64 * lane_ops_per_instr / s * 1e-12;
    printf("%-30s %8.3f ms  clk %.3f GHz  %6.3f cyc/wave-instr/SIMD  %7.2f T lane-flop/s\n", name, ms / reps, ghz, cyc, tflops);
    return cyc;
}

// ---- IEEE conformance: division, sqrt, reciprocal-of-sqrt-then-multiply as written in source ----
__global__ void conf_kernel(const float* x, const float* y, float* qdiv, float* qsqrt, float* qrcp, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    qdiv[i] = x[i] / y[i];
    qsqrt[i] = __builtin_sqrtf(fabsf(x[i]));
    qrcp[i] = 1.0f / __builtin_sqrtf(fabsf(y[i]));
}

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz  LDS/block %zu\n", prop.name, n_cu, prop.clockRate, prop.sharedMemPerBlock);
    float* dout;
    CK(hipMalloc(&dout, 64));
    run_rate<0>("v_fma_f32", dout, n_cu, 2);
    run_rate<1>("v_pk_fma_f32", dout, n_cu, 4);
    run_rate<2>("v_mul_f32/v_add_f32", dout, n_cu, 1);
    run_rate<3>("v_pk_mul_f32/v_pk_add_f32", dout, n_cu, 2);
    run_rate<4>("v_fma_f32 (sgpr operand)", dout, n_cu, 2);
    run_rate<5>("v_fma_f32+v_cmp_ge_f32(sgpr)", dout, n_cu, 1.5);
    run_rate<6>("v_fma_f32 sgpr src0", dout, n_cu, 2);
    run_rate<7>("v_fma_f32 sgpr src2", dout, n_cu, 2);
    run_rate<8>("v_fmac_f32 sgpr src0 (VOP2)", dout, n_cu, 2);
    run_rate<9>("v_fmac_f32 vgpr (VOP2)", dout, n_cu, 2);
    run_rate<10>("v_sub/v_mul sgpr src0 (VOP2)", dout, n_cu, 1);
    run_rate<11>("16 v_fma per ds_read_b128 bcast", dout, n_cu, 2);
    run_rate<12>("7 v_fma + 1 v_cmp(sgpr)", dout, n_cu, 15.0/8);
    run_rate<14>("v_fma + v_cmp_e64 |abs| vgpr", dout, n_cu, 1.5);
    run_rate<15>("v_max3_f32", dout, n_cu, 2);
    run_rate<16>("v_pk_mul_f32", dout, n_cu, 2);
    run_rate<17>("v_fma + v_cmp_e64 vgpr", dout, n_cu, 1.5);
    run_rate<18>("v_cmp_e64 vgpr only", dout, n_cu, 1);
    run_rate<13>("v_cmp_e32 vcc only", dout, n_cu, 1);
    run_rate<19>("v_cmp_e32 vcc + s_or_b64", dout, n_cu, 1);
    run_rate<20>("v_cmp_ge_u32_e64", dout, n_cu, 1);
    run_rate<21>("v_max_f32 VOP2", dout, n_cu, 1);
    run_rate<22>("v_max_f32_e64 |a|,|b|", dout, n_cu, 1);
    run_rate<23>("v_or3_b32 x4 + v_and_b32 x4", dout, n_cu, 1);
    // Issue cost per instruction class, in the classes rocprofv3's SQ_INSTS_VALU_* counters report (bench.py weights the
    // kernel's measured mix with these: roofline.frac).  Printed once more as one JSON line for tools/pmc_constants.py.
    const double c_fma = run_rate<0>("class FMA_F32  v_fma_f32", dout, n_cu, 2);
    const double c_add = run_rate<33>("class ADD_F32  v_sub_f32", dout, n_cu, 1);
    const double c_mul = run_rate<34>("class MUL_F32  v_mul_f32", dout, n_cu, 1);
    const double c_rcp = run_rate<24>("class TRANS_F32 v_rcp_f32", dout, n_cu, 1);
    const double c_sqrt = run_rate<25>("class TRANS_F32 v_sqrt/v_rsq", dout, n_cu, 1);
    const double c_int = run_rate<26>("class INT32 add/shift/and", dout, n_cu, 1);
    const double c_imul = run_rate<27>("class INT32 v_mul_lo/hi_u32", dout, n_cu, 1);
    const double c_cvt = run_rate<28>("class CVT v_cvt_f32_u32/u32_f32", dout, n_cu, 1);
    run_rate<29>("other: v_cndmask_b32 vcc (in place)", dout, n_cu, 1);
    run_rate<38>("other: v_cndmask_b32 vcc (rotating)", dout, n_cu, 1);
    run_rate<37>("other: v_cmp_e32 + v_cndmask vcc", dout, n_cu, 1);
    run_rate<39>("other: v_min_f32/v_max_f32", dout, n_cu, 1);
    const double c_cnd = run_rate<36>("other: v_cndmask_b32_e64 sgpr mask", dout, n_cu, 1);
    const double c_mov = run_rate<30>("other: v_mov_b32", dout, n_cu, 1);
    const double c_i64 = run_rate<31>("class INT64 v_lshlrev_b64", dout, n_cu, 1);
    const double c_mbc = run_rate<32>("other: v_mbcnt_lo/hi", dout, n_cu, 1);
    const double c_cmp = run_rate<18>("other: v_cmp_e64", dout, n_cu, 1);
    const double c_rdl = run_rate<35>("other: v_readlane_b32", dout, n_cu, 1);
    printf("VALU_CLASS_CYCLES {\"FMA_F32\": %.3f, \"ADD_F32\": %.3f, \"MUL_F32\": %.3f, \"TRANS_F32\": %.3f, \"INT32\": %.3f, \"INT32_MUL\": %.3f, "
           "\"CVT\": %.3f, \"INT64\": %.3f, \"OTHER\": %.3f, \"other_detail\": {\"cndmask\": %.3f, \"mov\": %.3f, \"mbcnt\": %.3f, \"cmp\": %.3f, \"readlane\": %.3f}}\n",
           c_fma, c_add, c_mul, 0.5 * (c_rcp + c_sqrt), c_int, c_imul, c_cvt, c_i64, 0.25 * (c_cnd + c_mov + c_mbc + c_cmp), c_cnd, c_mov, c_mbc, c_cmp, c_rdl);

    // conformance
    const int n = 1 << 22;
    std::vector<float> hx(n), hy(n), hd(n), hs(n), hr(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) {
        uint32_t a = lcg(s), b = lcg(s);
        float fx, fy;
        if (i & 1) {  // full-range bit patterns (finite)
            a = (a & 0x807fffffu) | (((a >> 23) % 254 + 1) << 23);
            b = (b & 0x807fffffu) | (((b >> 23) % 254 + 1) << 23);
            __builtin_memcpy(&fx, &a, 4);
            __builtin_memcpy(&fy, &b, 4);
        } else {  // typical magnitudes
            fx = ((int32_t)a) * (1.0f / 65536.0f / 128.0f);
            fy = ((int32_t)b) * (1.0f / 65536.0f / 4096.0f);
            if (fy == 0) fy = 1;
        }
        hx[i] = fx;
        hy[i] = fy;
    }
    float *dx, *dy, *dd, *ds, *dr;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&dd, n * 4)); CK(hipMalloc(&ds, n * 4)); CK(hipMalloc(&dr, n * 4));
    CK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hy.data(), n * 4, hipMemcpyHostToDevice));
    conf_kernel<<<n / 256, 256>>>(dx, dy, dd, ds, dr, n);
    CK(hipMemcpy(hd.data(), dd, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), dr, n * 4, hipMemcpyDeviceToHost));
    long bad_d = 0, bad_s = 0, bad_r = 0;
    for (int i = 0; i < n; ++i) {
        volatile float q = hx[i] / hy[i];
        volatile float sq = sqrtf(fabsf(hx[i]));
        volatile float sy = sqrtf(fabsf(hy[i]));
        volatile float rc = 1.0f / sy;
        float qq = q, ss = sq, rr = rc;
        if (memcmp(&qq, &hd[i], 4) != 0 && !(std::isnan(qq) && std::isnan(hd[i]))) { if (bad_d < 5) printf("div mismatch %a / %a: host %a dev %a\n", hx[i], hy[i], qq, hd[i]); ++bad_d; }
        if (memcmp(&ss, &hs[i], 4) != 0) { if (bad_s < 5) printf("sqrt mismatch %a: host %a dev %a\n", hx[i], ss, hs[i]); ++bad_s; }
        if (memcmp(&rr, &hr[i], 4) != 0) { if (bad_r < 5) printf("rcp-sqrt mismatch %a: host %a dev %a\n", hy[i], rr, hr[i]); ++bad_r; }
    }
    printf("conformance over %d samples: div mismatches %ld, sqrt mismatches %ld, 1/sqrt mismatches %ld\n", n, bad_d, bad_s, bad_r);
    return (bad_d || bad_s || bad_r) ? 2 : 0;
}
