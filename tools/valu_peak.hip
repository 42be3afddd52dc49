// valu_peak.hip -- what one MI355X SIMD sustains, per instruction class, in wave64 instructions per cycle.
//
// Calibrates the ceiling of the race kernel's binding resource (VALU issue).  Every kernel below is a loop of 128
// INDEPENDENT instructions of one class (16 accumulator chains x 8), forced by inline assembly; the grid puts W waves
// on every SIMD of the chip (W blocks of 256 threads per CU), and the rate is taken from WALL time (hipEvents), so no
// assumption about what a shader-clock tick is enters:
//     cycles per wave64 instruction per SIMD = 1024 SIMDs x 2.4e9 Hz x seconds / (waves x instructions per wave)
// (2.4 GHz nominal; the race kernel's profiles show 2.37-2.39 GHz under load, so the figures are <= 1.5 % high).
// Not product code.   hipcc --offload-arch=gfx950 -O3 -o tools/valu_peak tools/valu_peak.hip && tools/valu_peak > profiles/r3_valu_peak.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kChains = 16, kUnroll = 8;

// one class per kernel: BODY(j) is the instruction on chain j
#define DEFINE_KERNEL(NAME, DECL, BODY, SINK)                                                           \
    __global__ void __launch_bounds__(256) NAME(uint32_t iters, double *out)                            \
    {                                                                                                   \
        DECL;                                                                                           \
        for (uint32_t it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int u = 0; u < kUnroll; ++u) {                                       \
                _Pragma("unroll") for (int j = 0; j < kChains; ++j) { BODY; }                           \
            }                                                                                           \
        }                                                                                               \
        double s = 0;                                                                                   \
        _Pragma("unroll") for (int j = 0; j < kChains; ++j) s += SINK;                                  \
        if (s == 1.2345e300) out[threadIdx.x] = s;                                                      \
    }

#define F64_DECL double a[kChains]; const double b = 1.0000001 + threadIdx.x * 1e-9, c = 1e-7; \
    _Pragma("unroll") for (int j = 0; j < kChains; ++j) a[j] = 1.0 + j
#define U32_DECL uint32_t a[kChains]; uint32_t b = 0x9E3779B9u + threadIdx.x, c = 12345u; \
    _Pragma("unroll") for (int j = 0; j < kChains; ++j) a[j] = j * 77u + threadIdx.x
#define F32_DECL float a[kChains]; const float b = 1.0000001f, c = 1e-7f; \
    _Pragma("unroll") for (int j = 0; j < kChains; ++j) a[j] = 1.0f + j

DEFINE_KERNEL(k_fma_f64, F64_DECL, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)), a[j])
DEFINE_KERNEL(k_add_f64, F64_DECL, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c)), a[j])
DEFINE_KERNEL(k_mul_f64, F64_DECL, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b)), a[j])
DEFINE_KERNEL(k_min_f64, F64_DECL, asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b)), a[j])
DEFINE_KERNEL(k_cmp_f64, F64_DECL, asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc"), a[j])
DEFINE_KERNEL(k_cvt_f64_u32, F64_DECL; uint32_t q = threadIdx.x, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[j]) : "v"(q)), a[j])
DEFINE_KERNEL(k_cndmask_vop2, U32_DECL, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : ), (double)a[j])
DEFINE_KERNEL(k_cndmask_vop3, U32_DECL; unsigned long long m = 0x5555555555555555ull + blockIdx.x, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(m)), (double)a[j])
DEFINE_KERNEL(k_add_u32, U32_DECL, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b)), (double)a[j])
DEFINE_KERNEL(k_and_b32, U32_DECL, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b)), (double)a[j])
DEFINE_KERNEL(k_xor_b32, U32_DECL, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b)), (double)a[j])
DEFINE_KERNEL(k_bfe_u32, U32_DECL, asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(a[j])), (double)a[j])
DEFINE_KERNEL(k_lshl_add_u32, U32_DECL, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(c)), (double)a[j])
DEFINE_KERNEL(k_and_or_b32, U32_DECL, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)), (double)a[j])
DEFINE_KERNEL(k_mad_u64_u32, unsigned long long a[kChains]; uint32_t b = 0xD2511F53u; uint32_t c = threadIdx.x | 1u;
              _Pragma("unroll") for (int j = 0; j < kChains; ++j) a[j] = j + threadIdx.x,
              asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c) : "vcc"), (double)a[j])
DEFINE_KERNEL(k_fma_f32, F32_DECL, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)), (double)a[j])
DEFINE_KERNEL(k_mov_b32, U32_DECL, asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(b)), (double)a[j])

struct Case {
    const char *name;
    void (*fn)(uint32_t, double *);
};

int main()
{
    const Case cases[] = {
        {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_min_f64", k_min_f64},
        {"v_cmp_gt_f64 (vcc)", k_cmp_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cndmask_b32 (VOP2, vcc)", k_cndmask_vop2},
        {"v_cndmask_b32_e64 (SGPR mask)", k_cndmask_vop3}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
        {"v_xor_b32", k_xor_b32}, {"v_bfe_u32", k_bfe_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_and_or_b32", k_and_or_b32},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_fma_f32", k_fma_f32}, {"v_mov_b32", k_mov_b32},
    };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    CHECK(hipMalloc(&out, 256 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t iters = 20000;
    const double per_wave = (double)iters * kChains * kUnroll;
    printf("{\"unit\": \"cycles per wave64 instruction per SIMD at 2.4 GHz nominal, from wall time\", \"cus\": %d, \"iters\": %u, "
           "\"instructions_per_wave\": %.0f, \"waves_per_simd\": {", cus, iters, per_wave);
    const int ws[] = {1, 2, 3, 4, 8};
    for (size_t wi = 0; wi < sizeof(ws) / sizeof(ws[0]); ++wi) {
        const int W = ws[wi];
        printf("%s\"%d\": {", wi ? ", " : "", W);
        for (size_t ci = 0; ci < sizeof(cases) / sizeof(cases[0]); ++ci) {
            const dim3 grid(cus * W), block(256);                    // W blocks of 4 waves per CU: W waves per SIMD
            hipLaunchKernelGGL(cases[ci].fn, grid, block, 0, 0, 1000u, out);          // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(cases[ci].fn, grid, block, 0, 0, iters, out);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double waves = (double)cus * W * 4;
            const double simds = (double)cus * 4;
            const double cyc = simds * 2.4e9 * (best * 1e-3) / (waves * per_wave);
            printf("%s\"%s\": %.3f", ci ? ", " : "", cases[ci].name, cyc);
            fflush(stdout);
        }
        printf("}");
    }
    printf("}}\n");
    return 0;
}
