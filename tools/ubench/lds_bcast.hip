// Broadcast ds_read_b128 throughput on gfx950: every lane of a wave reads the same 16 bytes (the closest-hit scan's table
// read).  Reports CU cycles per wave-level read with 16 waves per CU issuing nothing else, and with packed FMAs in between.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/lds_bcast tools/ubench/lds_bcast.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float V4f __attribute__((ext_vector_type(4)));
typedef float V2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 16384;

// READS broadcast reads + FMAS v_pk_fma_f32 per iteration
template <int READS, int FMAS, int WIDTH>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps)
{
    __shared__ V4f lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = V4f{1.f, 0.5f, 0.25f, (float)i};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    V2f x0 = {1.f, 2.f}, x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0, m = {0.999f, 0.999f}, a = {0.001f, 0.001f};
    V4f acc = {0, 0, 0, 0};
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) V4f*)lds;
    for (int i = 0; i < ITERS; ++i) {
        const unsigned addr = base + ((i * READS) & 1023 & ~7) * 16;
        V4f r0v, r1v, r2v, r3v;
        if (WIDTH == 128) {
            if (READS == 4) asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)" : "=&v"(r0v), "=&v"(r1v), "=&v"(r2v), "=&v"(r3v) : "v"(addr));
            if (READS == 3) asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b128 %2, %3 offset:32\n s_waitcnt lgkmcnt(0)" : "=&v"(r0v), "=&v"(r1v), "=&v"(r2v) : "v"(addr));
            if (READS == 2) asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=&v"(r0v), "=&v"(r1v) : "v"(addr));
        }
        acc += r0v;
        if (FMAS >= 8)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(a));
        if (FMAS >= 16)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(a));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    V2f q = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (q.x + q.y + acc.x + acc.w == 12345.678f) out[0] = q.x;
}

// v_pk_fma_f32 with the broadcast operand in an SGPR pair instead of a VGPR pair (would a scalar-loaded table be as fast?)
template <int SGPR>
__global__ __launch_bounds__(512) void kp(float* out, unsigned long long* stamps, float sa, float sb)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    V2f x0 = {1.f, 2.f}, x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0, a = {0.001f, 0.001f};
    V2f mv = {sa + threadIdx.x * 1e-9f, sb};
    V2f ms = {sa, sb};
    for (int i = 0; i < ITERS; ++i) {
        if (SGPR)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[0,1,0]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[0,1,0]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[0,1,0]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(ms), "v"(a));
        else
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[0,1,0]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[0,1,0]\n"
                         "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[0,1,0]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[0,1,0]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(mv), "v"(a));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    V2f q = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (q.x + q.y == 12345.678f) out[0] = q.x;
}

template <int SGPR>
void runp(const char* name, int n_cu, float* dout, unsigned long long* dst)
{
    const int blocks = n_cu * 2;
    kp<SGPR><<<blocks, 512>>>(dout, dst, 0.999f, 0.998f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) kp<SGPR><<<blocks, 512>>>(dout, dst, 0.999f, 0.998f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long st[2];
    CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)st[0] / (double)st[1] * 0.1, s = ms * 1e-3 / 3;
    printf("%-44s %7.3f ms  clk %.2f GHz  %6.2f cycles per wave instruction per SIMD\n", name, ms / 3, ghz, s * ghz * 1e9 / (4.0 * ITERS * 8));
}

template <int READS, int FMAS, int WIDTH>
void run(const char* name, int n_cu, float* dout, unsigned long long* dst)
{
    const int blocks = n_cu * 2;  // 2 x 8 waves = 4 waves per SIMD, like the render kernel
    k<READS, FMAS, WIDTH><<<blocks, 512>>>(dout, dst);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k<READS, FMAS, WIDTH><<<blocks, 512>>>(dout, dst);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long st[2];
    CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)st[0] / (double)st[1] * 0.1, s = ms * 1e-3 / 3;
    const double cu_cycles = s * ghz * 1e9;
    const double reads_per_cu = 16.0 * ITERS * READS;
    printf("%-44s %7.3f ms  clk %.2f GHz  %6.2f CU-cycles per wave read  %6.1f cycles per iteration per SIMD\n", name, ms / 3, ghz,
           cu_cycles / reads_per_cu, cu_cycles / (4.0 * ITERS));
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    float* dout;
    unsigned long long* dst;
    CK(hipMalloc(&dout, 64));
    CK(hipMalloc(&dst, 16));
    run<4, 0, 128>("4 x ds_read_b128 bcast", n_cu, dout, dst);
    run<3, 0, 128>("3 x ds_read_b128 bcast", n_cu, dout, dst);
    run<2, 0, 128>("2 x ds_read_b128 bcast", n_cu, dout, dst);
    run<4, 8, 128>("4 x b128 + 8 v_pk_fma", n_cu, dout, dst);
    run<4, 16, 128>("4 x b128 + 16 v_pk_fma", n_cu, dout, dst);
    run<3, 8, 128>("3 x b128 + 8 v_pk_fma", n_cu, dout, dst);
    run<3, 16, 128>("3 x b128 + 16 v_pk_fma", n_cu, dout, dst);
    run<2, 16, 128>("2 x b128 + 16 v_pk_fma", n_cu, dout, dst);
    runp<0>("v_pk_fma_f32, broadcast operand in VGPRs", n_cu, dout, dst);
    runp<1>("v_pk_fma_f32, broadcast operand in SGPRs", n_cu, dout, dst);
    return 0;
}
