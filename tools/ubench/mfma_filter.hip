// bf16x3 MFMA as an f32-accurate affine evaluator: numerics and issue rate on gfx950.
//
// Question 1 (numerics): h = A' + i B' + j C' with every f32 coefficient split exactly into three bf16 parts and the cell
// indices into base-256 digits (exact in bf16) is a K = 15 dot product whose 15 products are each exact in f32.  How far
// is v_mfma_f32_32x32x16_bf16's result from the exact sum, in units of 2^-24 * (|A'| + i|B'| + j|C'|)?
// Question 2 (rate): cycles per MFMA per SIMD for (a) bare MFMAs, (b) + 8 v_max3_f32 + 1 v_cmp per MFMA (the "does any
// of these 16 spheres pass" decision), (c) + one ds_read_b128 of the A operand per 4 MFMAs, at 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_filter tools/ubench/mfma_filter.hip
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

typedef short short8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

// ---------------- numerics ----------------
__global__ void mfma_once(const uint16_t* A, const uint16_t* B, float* D, int batches)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    for (int b = blockIdx.x; b < batches; b += gridDim.x) {
        short8 a, bb;
        for (int j = 0; j < 8; ++j) {
            a[j] = (short)A[((size_t)b * 32 + r) * 16 + 8 * h + j];
            bb[j] = (short)B[((size_t)b * 32 + r) * 16 + 8 * h + j];
        }
        float16v c = {};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, c, 0, 0, 0);
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * h;
            D[((size_t)b * 32 + row) * 32 + r] = c[v];
        }
    }
}

static uint16_t bf16_rn(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t b)
{
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void split3(float x, uint16_t p[3])
{
    p[0] = bf16_rn(x);
    float r1 = x - bf16_f(p[0]);
    p[1] = bf16_rn(r1);
    float r2 = r1 - bf16_f(p[1]);
    p[2] = bf16_rn(r2);
}

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static double urand()
{
    rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
    return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
}

static int numerics(int batches, int mode)
{
    std::vector<uint16_t> A((size_t)batches * 32 * 16, 0), B((size_t)batches * 32 * 16, 0);
    std::vector<float> coef((size_t)batches * 32 * 3);
    std::vector<int> ij((size_t)batches * 32 * 2);
    long inexact_split = 0;
    for (int b = 0; b < batches; ++b) {
        for (int n = 0; n < 32; ++n) {
            int i = (int)(urand() * 4096), j = (int)(urand() * 4096);
            ij[((size_t)b * 32 + n) * 2] = i;
            ij[((size_t)b * 32 + n) * 2 + 1] = j;
            uint16_t* col = &B[((size_t)b * 32 + n) * 16];
            const uint16_t one = bf16_rn(1.0f), ih = bf16_rn((float)(i >> 8)), il = bf16_rn((float)(i & 255)), jh = bf16_rn((float)(j >> 8)), jl = bf16_rn((float)(j & 255));
            const uint16_t v[16] = {one, one, one, ih, ih, ih, il, il, il, jh, jh, jh, jl, jl, jl, 0};
            memcpy(col, v, 32);
        }
        for (int m = 0; m < 32; ++m) {
            float a, bq, c;
            if (mode == 0) {  // typical magnitudes of the shared-origin table
                a = (float)((urand() * 2 - 1) * 300.0);
                bq = (float)((urand() * 2 - 1) * 0.3);
                c = (float)((urand() * 2 - 1) * 0.3);
            } else if (mode == 1) {  // heavy cancellation: A' ~ -(i B' + j C') for i, j ~ 2000
                bq = (float)((urand() * 2 - 1) * 0.3);
                c = (float)((urand() * 2 - 1) * 0.3);
                a = -(2000.0f * bq + 2000.0f * c) * (float)(1.0 + (urand() - 0.5) * 1e-3);
            } else {  // wide exponent spread
                a = (float)ldexp(urand() * 2 - 1, (int)(urand() * 40) - 20);
                bq = (float)ldexp(urand() * 2 - 1, (int)(urand() * 40) - 30);
                c = (float)ldexp(urand() * 2 - 1, (int)(urand() * 40) - 30);
            }
            coef[((size_t)b * 32 + m) * 3] = a;
            coef[((size_t)b * 32 + m) * 3 + 1] = bq;
            coef[((size_t)b * 32 + m) * 3 + 2] = c;
            uint16_t* row = &A[((size_t)b * 32 + m) * 16];
            uint16_t p[3];
            split3(a, p);
            row[0] = p[0]; row[1] = p[1]; row[2] = p[2];
            if ((double)bf16_f(p[0]) + (double)bf16_f(p[1]) + (double)bf16_f(p[2]) != (double)a) ++inexact_split;
            split3(bq * 256.0f, p);
            row[3] = p[0]; row[4] = p[1]; row[5] = p[2];
            split3(bq, p);
            row[6] = p[0]; row[7] = p[1]; row[8] = p[2];
            split3(c * 256.0f, p);
            row[9] = p[0]; row[10] = p[1]; row[11] = p[2];
            split3(c, p);
            row[12] = p[0]; row[13] = p[1]; row[14] = p[2];
            row[15] = 0;
        }
    }
    uint16_t *dA, *dB;
    float* dD;
    CK(hipMalloc(&dA, A.size() * 2));
    CK(hipMalloc(&dB, B.size() * 2));
    CK(hipMalloc(&dD, (size_t)batches * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    mfma_once<<<256, 64>>>(dA, dB, dD, batches);
    CK(hipDeviceSynchronize());
    std::vector<float> D((size_t)batches * 1024);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_vs_fma = 0;
    long n_ne_fma = 0;
    for (int b = 0; b < batches; ++b)
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                const float a = coef[((size_t)b * 32 + m) * 3], bq = coef[((size_t)b * 32 + m) * 3 + 1], c = coef[((size_t)b * 32 + m) * 3 + 2];
                const int i = ij[((size_t)b * 32 + n) * 2], j = ij[((size_t)b * 32 + n) * 2 + 1];
                // exact value of the K = 15 sum as the hardware is given it (parts may not add up to the coefficient when it underflows)
                double exact = 0, mag = 0;
                for (int k = 0; k < 16; ++k) {
                    const double t = (double)bf16_f(A[((size_t)b * 32 + m) * 16 + k]) * (double)bf16_f(B[((size_t)b * 32 + n) * 16 + k]);
                    exact += t;  // 15 terms of <= 24 bits spread over < 2^60: double holds the sum to ~2^-53 relative of the largest
                    mag += fabs(t);
                }
                const double mag_coef = fabs((double)a) + i * fabs((double)bq) + j * fabs((double)c);
                const double got = D[((size_t)b * 32 + m) * 32 + n];
                const double e = fabs(got - exact) / (ldexp(1.0, -24) * (mag_coef > 0 ? mag_coef : 1e-300));
                if (e > worst) worst = e;
                const float f = fmaf((float)i, bq, fmaf((float)j, c, a));
                if ((double)f != got) ++n_ne_fma;
                const double e2 = fabs(got - (double)f) / (ldexp(1.0, -24) * (mag_coef > 0 ? mag_coef : 1e-300));
                if (e2 > worst_vs_fma) worst_vs_fma = e2;
                (void)mag;
            }
    printf("numerics mode %d: %d x 1024 results; inexact 3-way splits %ld; max |mfma - exact| = %.3f u*M (u = 2^-24, M = |A'|+i|B'|+j|C'|); "
           "vs fmaf chain: %ld differ, max %.3f u*M\n", mode, batches, inexact_split, worst, n_ne_fma, worst_vs_fma);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dD));
    return 0;
}

// ---------------- rates ----------------
constexpr int ITERS = 4096;
// MODE 0: 4 independent MFMAs per iteration, nothing else
// MODE 1: + per MFMA 8 v_max3_f32 (|.| modifiers) over its 16 results and one v_cmp against a per-lane threshold
// MODE 2: MODE 1 + one 16-byte LDS read of the A fragment per 4 MFMAs (lane-linear addresses)
// MODE 3: general-ray shape: 4 MFMAs (h lo/hi K, S lo/hi K) -> 8 v_pk_fma (h*h - S) + 8 v_max3 + 1 cmp per 1024 tests
template <int MODE>
__global__ __launch_bounds__(512, 4) void rate_kernel(float* out, const uint16_t* tab, unsigned long long* stamps)
{
    extern __shared__ uint4 lds[];
    for (unsigned k = threadIdx.x; k < 2048; k += 512) lds[k] = reinterpret_cast<const uint4*>(tab)[k & 1023];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    short8 a = {1, 2, 3, 4, 5, 6, 7, (short)lane}, b0 = {(short)lane, 1, 1, 1, 2, 2, 2, 3}, b1 = b0 + (short)1, b2 = b0 + (short)2, b3 = b0 + (short)3;
    float thr = 1e30f + lane, thr2 = 2e30f + lane;
    float16v k0, k1, k2, k3;
    for (int v = 0; v < 16; ++v) { k0[v] = lane + v; k1[v] = lane - v; k2[v] = 2.f * lane + v; k3[v] = 3.f * lane - v; }
    unsigned long long hits = 0;
    float keep = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE >= 2) {
            const uint4 q = lds[((it & 31) * 64 + lane)];
            a = __builtin_bit_cast(short8, q);
        }
        float16v z = {};
        if (MODE == 4) {  // no matrix work at all
            float16v c0 = z, c1 = z, c2 = z, c3 = z;
            (void)c0; (void)c1; (void)c2; (void)c3;
        }
        float16v c0 = MODE == 4 ? z : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, z, 0, 0, 0);
        float16v c1 = MODE == 4 ? z : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, z, 0, 0, 0);
        float16v c2 = MODE == 4 ? z : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, z, 0, 0, 0);
        float16v c3 = MODE == 4 ? z : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b3, z, 0, 0, 0);
        if (MODE == 0) {
            keep += c0[0] + c1[5] + c2[9] + c3[15];
        } else if (MODE == 1 || MODE == 2) {
            auto any16 = [&](const float16v& c) {
                float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c[0]), __builtin_fabsf(c[1])), __builtin_fabsf(c[2]));
#pragma unroll
                for (int v = 3; v < 15; v += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(c[v])), __builtin_fabsf(c[v + 1]));
                m = __builtin_fmaxf(m, __builtin_fabsf(c[15]));
                return m >= thr;
            };
            const bool p = any16(c0) | any16(c1) | any16(c2) | any16(c3);
            hits |= __ballot(p);
        } else if (MODE == 4 || MODE == 5) {
            // MODE 4: the decision VALU alone, on registers no MFMA writes (c* are loop-carried copies); MODE 5: same VALU next to
            // 4 MFMAs whose results are only folded in once per iteration (independent instruction streams inside one wave)
            auto any16 = [&](const float16v& c) {
                float m0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c[0]), __builtin_fabsf(c[1])), __builtin_fabsf(c[2]));
                float m1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(c[3]), __builtin_fabsf(c[4])), __builtin_fabsf(c[5]));
#pragma unroll
                for (int v = 6; v < 14; v += 4) {
                    m0 = __builtin_fmaxf(__builtin_fmaxf(m0, __builtin_fabsf(c[v])), __builtin_fabsf(c[v + 1]));
                    m1 = __builtin_fmaxf(__builtin_fmaxf(m1, __builtin_fabsf(c[v + 2])), __builtin_fabsf(c[v + 3]));
                }
                return __builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(__builtin_fabsf(c[14]), __builtin_fabsf(c[15])));
            };
            asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
            const float q0 = any16(k0), q1 = any16(k1), q2 = any16(k2), q3 = any16(k3);
            hits |= __ballot(q0 >= thr) | __ballot(q1 >= thr2) | __ballot(q2 >= thr) | __ballot(q3 >= thr2);
            if (MODE == 5) keep += c0[0] + c1[5] + c2[9] + c3[15];
            k0[it & 15] += 1.0f;

            float m = -1e30f;
#pragma unroll
            for (int v = 0; v < 16; v += 2) {
                const float q0 = __builtin_fmaf(c0[v] + c1[v], c0[v], -c2[v]), q1 = __builtin_fmaf(c0[v + 1], c0[v + 1], -c3[v + 1]);
                m = __builtin_fmaxf(__builtin_fmaxf(m, q0), q1);
            }
            hits |= __ballot(m >= thr);
        }
        if (hits) {  // never taken (thr is huge); keeps the slow path's shape out of the loop
            keep += 1.f;
            hits = 0;
        }
        b0[1] = (short)it;  // every operand changes every iteration (nothing is loop-invariant)
        b1[2] = (short)(it + 1);
        b2[3] = (short)(it + 2);
        b3[4] = (short)(it + 3);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    if (keep + k0[3] + k1[1] + k2[2] + k3[7] == 12345.678f) out[0] = keep;
}

template <int MODE>
static void run_rate(const char* name, int n_cu, float* dout, const uint16_t* dtab, unsigned long long* dst)
{
    const int blocks = n_cu * 2;  // 2 x 8 waves per CU = 4 waves per SIMD (the pinhole kernel's residency)
    CK(hipFuncSetAttribute((const void*)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
    rate_kernel<MODE><<<blocks, 512, 32768>>>(dout, dtab, dst);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) rate_kernel<MODE><<<blocks, 512, 32768>>>(dout, dtab, dst);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long st[2];
    CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)st[0] / (double)st[1] * 0.1;
    const double s = ms * 1e-3 / reps;
    const double mfma_per_simd = 4.0 * ITERS * 4;  // 4 MFMAs per iteration, 4 waves per SIMD
    const double cyc = s * ghz * 1e9 / mfma_per_simd;
    printf("%-44s %8.3f ms  clk %.3f GHz  %6.2f cyc per MFMA per SIMD = %5.2f cyc per 256 tests\n", name, ms / reps, ghz, cyc, cyc / 4.0);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d\n", prop.name, prop.multiProcessorCount);
    numerics(256, 0);
    numerics(256, 1);
    numerics(256, 2);
    float* dout;
    uint16_t* dtab;
    unsigned long long* dst;
    CK(hipMalloc(&dout, 64));
    CK(hipMalloc(&dtab, 32768));
    CK(hipMemset(dtab, 0x3c, 32768));
    CK(hipMalloc(&dst, 16));
    run_rate<0>("4 MFMA 32x32x16 bf16", prop.multiProcessorCount, dout, dtab, dst);
    run_rate<1>("4 MFMA + 4 x (8 max3 + cmp)", prop.multiProcessorCount, dout, dtab, dst);
    run_rate<2>("4 MFMA + 4 x (8 max3 + cmp) + ds_read_b128", prop.multiProcessorCount, dout, dtab, dst);
    run_rate<3>("4 MFMA + 16 fma + 8 max3 + cmp (general ray)", prop.multiProcessorCount, dout, dtab, dst);
    run_rate<4>("no MFMA: 4 x (8 max3 + cmp), 2 chains", prop.multiProcessorCount, dout, dtab, dst);
    run_rate<5>("4 MFMA beside 4 x (8 max3 + cmp) on other regs", prop.multiProcessorCount, dout, dtab, dst);
    return 0;
}
