// inst_microbench.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the race kernel
// is made of, measured on gfx950 at 2 waves per SIMD (the race kernel's occupancy).  Not product code.
// Each kernel runs R iterations of 16 independent copies of one instruction.
//   hipcc -O3 --offload-arch=gfx950 -o tools/inst_microbench tools/inst_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, DECL, BODY, SINK)                                                          \
    __global__ void __launch_bounds__(256) NAME(double *out, int iters)                      \
    {                                                                                           \
        DECL                                                                                    \
        for (int it = 0; it < iters; ++it) { BODY }                                             \
        SINK                                                                                    \
    }

// operands
#define DECL_D double a[16], b = 1.0000001 + threadIdx.x * 1e-9, c = 0.5; for (int i = 0; i < 16; ++i) a[i] = 1.0 + i + threadIdx.x * 1e-3;
#define SINK_D double s = 0; for (int i = 0; i < 16; ++i) s += a[i]; out[blockIdx.x * 256 + threadIdx.x] = s;
#define DECL_U uint32_t u[16], v = threadIdx.x * 2654435761u + 12345u; for (int i = 0; i < 16; ++i) u[i] = v + i * 977u;
#define SINK_U uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= u[i]; out[blockIdx.x * 256 + threadIdx.x] = s;

#define X_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define X_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define X_MIN64(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define X_CMP64(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a[i]), "v"(b), "v"(u0), "v"(u1) : "vcc");
#define X_CMPONLY64(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define X_CMPU64(i) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define X_CMPU32(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(v) : "vcc");
#define X_CVT(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u0));
#define X_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(v) : "vcc");
#define X_ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(v));
#define X_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(v));
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(u[i]), "v"(v) : "vcc");
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(v));
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(v));
#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g));
#define X_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[i]));

KERNEL(k_add64, DECL_D, REP16(X_ADD64), SINK_D)
KERNEL(k_mul64, DECL_D, REP16(X_MUL64), SINK_D)
KERNEL(k_min64, DECL_D, REP16(X_MIN64), SINK_D)
KERNEL(k_cmp64, DECL_D, REP16(X_CMPONLY64), SINK_D)
KERNEL(k_cmpu64, DECL_D, REP16(X_CMPU64), SINK_D)
KERNEL(k_cmpu32, DECL_U, REP16(X_CMPU32), SINK_U)
KERNEL(k_cnd, DECL_U, REP16(X_CND), SINK_U)
KERNEL(k_add32, DECL_U, REP16(X_ADD32), SINK_U)
KERNEL(k_xor, DECL_U, REP16(X_XOR), SINK_U)
KERNEL(k_mulhi, DECL_U, REP16(X_MULHI), SINK_U)
KERNEL(k_mullo, DECL_U, REP16(X_MULLO), SINK_U)
KERNEL(k_bfe, DECL_U, REP16(X_BFE), SINK_U)
__global__ void __launch_bounds__(256) k_mad64(double *out, int iters)
{
    DECL_U
    uint64_t w[16];
    for (int i = 0; i < 16; ++i) w[i] = 0;
    for (int it = 0; it < iters; ++it) { REP16(X_MAD64) }
    uint64_t s = 0;
    for (int i = 0; i < 16; ++i) s ^= w[i];
    out[blockIdx.x * 256 + threadIdx.x] = (double)s;
}
__global__ void __launch_bounds__(256) k_cvt(double *out, int iters)
{
    DECL_D
    uint32_t u0 = threadIdx.x;
    for (int it = 0; it < iters; ++it) { REP16(X_CVT) }
    SINK_D
}
__global__ void __launch_bounds__(256) k_fma32(double *out, int iters)
{
    float f[16], g = 1.0001f;
    for (int i = 0; i < 16; ++i) f[i] = 1.0f + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) { REP16(X_FMA32) }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static int g_waves_per_simd = 2;

template <typename K>
static double run(K kern, const char *name, double *d)
{
    const int iters = 20000, blocks = 256 * g_waves_per_simd;       // k blocks x 4 waves per CU = k waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    // per SIMD: 2 waves x iters x 16 instructions in ms at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)g_waves_per_simd * iters * 16);
    printf("  \"%s\": %.2f,\n", name, cyc);
    return cyc;
}

int main(int argc, char **argv)
{
    if (argc > 1) g_waves_per_simd = atoi(argv[1]);
    double *d;
    (void)hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    printf("{\"unit\": \"cycles per wave64 instruction per SIMD (2.4 GHz assumed)\", \"waves_per_simd\": %d,\n", g_waves_per_simd);
    run(k_add32, "v_add_u32", d);
    run(k_xor, "v_xor_b32", d);
    run(k_bfe, "v_bfe_u32", d);
    run(k_cnd, "v_cndmask_b32", d);
    run(k_cmpu32, "v_cmp_gt_u32", d);
    run(k_fma32, "v_fma_f32", d);
    run(k_mulhi, "v_mul_hi_u32", d);
    run(k_mullo, "v_mul_lo_u32", d);
    run(k_mad64, "v_mad_u64_u32", d);
    run(k_add64, "v_add_f64", d);
    run(k_mul64, "v_mul_f64", d);
    run(k_min64, "v_min_f64", d);
    run(k_cmp64, "v_cmp_gt_f64", d);
    run(k_cmpu64, "v_cmp_gt_u64", d);
    run(k_cvt, "v_cvt_f64_u32", d);
    printf("  \"end\": 0}\n");
    return 0;
}
