// valu_peak.hip -- what one MI355X SIMD sustains, per instruction class, in wave64 instructions per cycle.
//
// Calibrates the ceiling of the race kernel's binding resource (VALU issue).  Every kernel below is a loop of 128
// INDEPENDENT instructions of one class (16 accumulator chains x 8), forced by inline assembly; the grid puts W waves
// on every SIMD of the chip (W blocks of 256 threads per CU), and the rate is taken from WALL time (hipEvents), so no
// assumption about what a shader-clock tick is enters (every block also asks for floor(160 KB / W) of LDS it never touches,
// so that no CU can hold more than its W blocks and the waves are spread evenly):
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
    __global__ void __launch_bounds__(768) NAME(uint32_t iters, double *out)                            \
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
DEFINE_KERNEL(k_cmp_u32, U32_DECL, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc"), (double)a[j])
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

// VCC written right before it is read as a lane mask (the microbenchmark above never writes it)
DEFINE_KERNEL(k_cndmask_vop2_fresh, U32_DECL,
              if (j == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[15]), "v"(b) : "vcc");
              asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : ), (double)a[j])

// the race kernel's compare-exchange (race_isa.hip.h cmpx_time): 8 independent (time, payload) pairs per wave, per
// "instruction" of the table = one comparator of 5 instructions
#define CMPX_DECL double t[kChains]; uint32_t q[kChains]; \
    _Pragma("unroll") for (int j = 0; j < kChains; ++j) { t[j] = 1.0 + j * 0.37 + threadIdx.x * 1e-3; q[j] = j; }
#define CMPX_SINK (t[j] + (double)q[j])
DEFINE_KERNEL(k_cmpx_vcc, CMPX_DECL,
              if (j < 8) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_min_f64 %0, %0, %1\n\tv_max_f64 %1, %0, %1\n\t"
                                      "v_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc"
                                      : "+v"(t[j]), "+v"(t[j + 8]), "+v"(q[j]), "+v"(q[j + 8]) : : "vcc"), CMPX_SINK)
DEFINE_KERNEL(k_cmpx_sgpr, CMPX_DECL; unsigned long long m,
              if (j < 8) asm volatile("v_cmp_gt_f64 %4, %0, %1\n\tv_min_f64 %0, %0, %1\n\tv_max_f64 %1, %0, %1\n\t"
                                      "v_cndmask_b32_e64 %2, %2, %3, %4\n\tv_cndmask_b32_e64 %3, %3, %2, %4"
                                      : "+v"(t[j]), "+v"(t[j + 8]), "+v"(q[j]), "+v"(q[j + 8]), "=&s"(m)), CMPX_SINK)

// runs of R VOP2 selects on VCC between runs of R v_add_u32 (VCC never written): what a select costs by run length
#define CND_VCC asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : )
#define ADD_U32 asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b))
DEFINE_KERNEL(k_run1, U32_DECL, if (j % 2 < 1) CND_VCC; else ADD_U32, (double)a[j])
DEFINE_KERNEL(k_run2, U32_DECL, if (j % 4 < 2) CND_VCC; else ADD_U32, (double)a[j])
DEFINE_KERNEL(k_run4, U32_DECL, if (j % 8 < 4) CND_VCC; else ADD_U32, (double)a[j])
DEFINE_KERNEL(k_run8, U32_DECL, if (j < 8) CND_VCC; else ADD_U32, (double)a[j])
// compare, two selects on its result, one add -- with the mask in VCC and in an SGPR pair
DEFINE_KERNEL(k_cmp_sel2_vcc, U32_DECL,
              if (j % 4 == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc");
              else if (j % 4 < 3) CND_VCC; else ADD_U32, (double)a[j])
DEFINE_KERNEL(k_cmp_sel2_sgpr, U32_DECL; unsigned long long m = 0,
              if (j % 4 == 0) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(a[j]), "v"(b));
              else if (j % 4 < 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(m)); else ADD_U32,
              (double)a[j])

struct Case {
    const char *name;
    void (*fn)(uint32_t, double *);
};

int main()
{
    const Case cases[] = {
        {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_min_f64", k_min_f64},
        {"v_cmp_gt_f64 (vcc)", k_cmp_f64}, {"v_cmp_lt_u32 (vcc)", k_cmp_u32}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cndmask_b32 (VOP2, vcc)", k_cndmask_vop2},
        {"v_cndmask_b32_e64 (SGPR mask)", k_cndmask_vop3}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
        {"v_xor_b32", k_xor_b32}, {"v_bfe_u32", k_bfe_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_and_or_b32", k_and_or_b32},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_fma_f32", k_fma_f32}, {"v_mov_b32", k_mov_b32},
        {"v_cndmask_b32 (VOP2, vcc written per 16)", k_cndmask_vop2_fresh},
        {"selects on vcc in runs of 1 / v_add_u32 in runs of 1", k_run1}, {"selects on vcc in runs of 2 / adds in runs of 2", k_run2},
        {"selects on vcc in runs of 4 / adds in runs of 4", k_run4}, {"selects on vcc in runs of 8 / adds in runs of 8", k_run8},
        {"v_cmp_lt_u32 vcc, 2 selects on vcc, 1 add (mean)", k_cmp_sel2_vcc}, {"v_cmp_lt_u32 sgpr, 2 selects on it, 1 add (mean)", k_cmp_sel2_sgpr},
        {"compare-exchange, vcc (per half comparator)", k_cmpx_vcc}, {"compare-exchange, SGPR mask (per half comparator)", k_cmpx_sgpr},
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
    // {waves per SIMD, threads per block}: W blocks of 4 waves per CU, and the race kernel's own shape (one block of 12 waves)
    const int shapes[][2] = {{1, 256}, {2, 256}, {3, 256}, {4, 256}, {5, 256}, {6, 256}, {8, 256}, {3, 768}, {2, 512}};
    for (size_t wi = 0; wi < sizeof(shapes) / sizeof(shapes[0]); ++wi) {
        const int W = shapes[wi][0], threads = shapes[wi][1];
        const int blocks_per_cu = W * 256 / threads;
        const size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 1024;
        if (threads == 256) printf("%s\"%d\": {", wi ? ", " : "", W);
        else printf(", \"%d (one block of %d)\": {", W, threads);
        for (size_t ci = 0; ci < sizeof(cases) / sizeof(cases[0]); ++ci) {
            const dim3 grid(cus * blocks_per_cu), block(threads);
            CHECK(hipFuncSetAttribute((const void *)cases[ci].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(cases[ci].fn, grid, block, lds, 0, 1000u, out);        // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(cases[ci].fn, grid, block, lds, 0, iters, out);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double waves = (double)cus * W * 4;                  // = grid x threads / 64
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
