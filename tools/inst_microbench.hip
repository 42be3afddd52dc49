// inst_microbench.hip -- issue cost, in SHADER CYCLES per wave64 instruction per SIMD, of the instructions the
// race kernel is made of, on gfx950.  Not product code.
//
// Method (round 2; the round-1 harness converted wall time at an assumed 2.4 GHz and so folded the chip's
// load-dependent clock into every figure):
//   * cycles are read INSIDE the kernel with s_memtime (one tick = one shader cycle, MI355X_MICROARCH.md
//     "s_memtime tick vs SQ PMC units"), around a loop of R iterations x 32 INDEPENDENT copies of one
//     instruction (32 distinct destinations, constant sources: no dependent chain at any distance);
//   * W waves per SIMD are forced by LDS: blocks of 256 threads (one wave per SIMD), each declaring
//     160 KiB / W of dynamic LDS, grid = 256 CUs x W, so every CU holds exactly W blocks;
//   * cost per instruction per SIMD = median over waves of dt / (W x R x 32); the effective shader clock
//     (d s_memtime / d s_memrealtime x 100 MHz) is printed beside it.
// Sanity anchors from MI355X_MICROARCH.md: v_fma_f32 / v_add_u32 = 2 cycles at >= 2 waves per SIMD, 4 for a
// wave alone; f64 FMA = 4 (78.6 TFLOP/s fp64 vector peak = half the fp32 rate).
//   hipcc -O3 --offload-arch=gfx950 -o tools/inst_microbench tools/inst_microbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP32(X) X(24,25) X(26,27) X(28,29) X(30,31) X(32,33) X(34,35) X(36,37) X(38,39) X(40,41) X(42,43) X(44,45) X(46,47) X(48,49) X(50,51) X(52,53) X(54,55) X(24,25) X(26,27) X(28,29) X(30,31) X(32,33) X(34,35) X(36,37) X(38,39) X(40,41) X(42,43) X(44,45) X(46,47) X(48,49) X(50,51) X(52,53) X(54,55)

struct Stamp {
    unsigned long long cycles, realtime;
};

#define PROLOGUE                                                                              \
    extern __shared__ unsigned char lds_[];                                                   \
    double db = 1.0000001 + threadIdx.x * 1e-9, dc = 0.5 + threadIdx.x * 1e-7;                \
    uint32_t ub = threadIdx.x * 2654435761u + 12345u, uc = threadIdx.x + 7u;                  \
    float fb = 1.0001f + threadIdx.x, fc = 0.25f;                                             \
    const uint32_t la = threadIdx.x * 8;                                                      \
    uint32_t *lp = reinterpret_cast<uint32_t *>(lds_) + threadIdx.x * 2;                      \
    lp[0] = ub; lp[1] = uc;                                                                   \
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(ub), "v"(uc) : "vcc");                    \
    __syncthreads();                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();

#define EPILOGUE                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)");                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                           \
    sink[blockIdx.x * 256 + threadIdx.x] = db + dc + ub + uc + fb + fc;                       \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};

// One asm statement per loop body: the 32 copies are back to back, the compiler cannot put hazard s_nops or
// waits between them.  Operands: %0 ub  %1 uc (u32)  %2 db  %3 dc (f64)  %4 fb  %5 fc (f32)  %6 la (LDS byte address).
// Destinations are the fixed registers v24..v55 (16 pairs, each written twice per iteration; the kernels stay
// under 64 VGPRs so that 8 waves per SIMD can be resident).
#define CLOBBERS "vcc", "s20", "s21", "s22", "s23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"

#define KERNEL(NAME, X)                                                                       \
    __global__ void __launch_bounds__(256, 8) NAME(Stamp *out, double *sink, int iters)          \
    {                                                                                         \
        PROLOGUE                                                                              \
        for (int it = 0; it < iters; ++it) {                                                  \
            asm volatile(REP32(X) : : "v"(ub), "v"(uc), "v"(db), "v"(dc), "v"(fb), "v"(fc), "v"(la) : CLOBBERS); \
        }                                                                                     \
        EPILOGUE                                                                              \
    }

#define V1(a) "v" #a
#define V2(a, b) "v[" #a ":" #b "]"
#define I_ADD32(a, b) "v_add_u32 " V1(a) ", %0, %1\n"
#define I_XOR(a, b) "v_xor_b32 " V1(a) ", %0, %1\n"
#define I_AND(a, b) "v_and_b32 " V1(a) ", %0, %1\n"
#define I_MOV(a, b) "v_mov_b32 " V1(a) ", %0\n"
#define I_BFE(a, b) "v_bfe_u32 " V1(a) ", %0, 3, 5\n"
#define I_LSHLADD(a, b) "v_lshl_add_u32 " V1(a) ", %0, 3, %1\n"
#define I_AND_OR(a, b) "v_and_or_b32 " V1(a) ", %0, %1, %1\n"
#define I_CND_VCC(a, b) "v_cndmask_b32 " V1(a) ", %0, %1, vcc\n"
#define I_CND_SGPR(a, b) "v_cndmask_b32 " V1(a) ", %0, %1, s[20:21]\n"
#define I_CMP32(a, b) "v_cmp_gt_u32 vcc, %0, %1\n"
#define I_CMP32_SGPR(a, b) "v_cmp_gt_u32 s[20:21], %0, %1\n"
#define I_FFBH(a, b) "v_ffbh_u32 " V1(a) ", %0\n"
#define I_BCNT(a, b) "v_bcnt_u32_b32 " V1(a) ", %0, 0\n"
#define I_FMA32(a, b) "v_fma_f32 " V1(a) ", %4, %5, %5\n"
#define I_CVT_F32_U32(a, b) "v_cvt_f32_u32 " V1(a) ", %0\n"
#define I_MULHI(a, b) "v_mul_hi_u32 " V1(a) ", %0, %1\n"
#define I_MULLO(a, b) "v_mul_lo_u32 " V1(a) ", %0, %1\n"
#define I_MUL24(a, b) "v_mul_u32_u24 " V1(a) ", %0, %1\n"
#define I_MAD64(a, b) "v_mad_u64_u32 " V2(a, b) ", s[20:21], %0, %1, 0\n"
#define I_ADD64(a, b) "v_add_f64 " V2(a, b) ", %2, %3\n"
#define I_MUL64(a, b) "v_mul_f64 " V2(a, b) ", %2, %3\n"
#define I_FMA64(a, b) "v_fma_f64 " V2(a, b) ", %2, %3, %3\n"
#define I_MIN64(a, b) "v_min_f64 " V2(a, b) ", %2, %3\n"
#define I_CMP64(a, b) "v_cmp_gt_f64 vcc, %2, %3\n"
#define I_CMP64_SGPR(a, b) "v_cmp_gt_f64 s[20:21], %2, %3\n"
#define I_CMPU64(a, b) "v_cmp_gt_u64 vcc, %2, %3\n"
#define I_CVT_F64_U32(a, b) "v_cvt_f64_u32 " V2(a, b) ", %0\n"
#define I_CVT_F64_F32(a, b) "v_cvt_f64_f32 " V2(a, b) ", %4\n"
#define I_LSHR64(a, b) "v_lshrrev_b64 " V2(a, b) ", 3, %2\n"
#define I_PKMOV(a, b) "v_pk_mov_b32 " V2(a, b) ", %2, %3\n"
#define I_MOV64(a, b) "v_mov_b64 " V2(a, b) ", %2\n"
#define I_DSR32(a, b) "ds_read_b32 " V1(a) ", %6\n"
#define I_DSR64(a, b) "ds_read_b64 " V2(a, b) ", %6\n"
#define I_DSW32(a, b) "ds_write_b32 %6, %0\n"
#define I_DSW64(a, b) "ds_write_b64 %6, %2\n"
#define I_SNOP(a, b) "s_nop 0\n"
#define I_SMOV(a, b) "s_mov_b32 s20, s21\n"
#define I_SANDX(a, b) "s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]\n"
// the kernel's most common pattern: f64 compare -> two selects (counted as ONE unit of three instructions)
#define I_CMPSEL(a, b) "v_cmp_gt_f64 vcc, %2, %3\n v_cndmask_b32 " V1(a) ", %0, %1, vcc\n v_cndmask_b32 " V1(b) ", %1, %0, vcc\n"
// a compare-exchange on (f64 key, u32 payload) as the race kernel's network issues it (5 instructions)
#define I_CMPX(a, b) "v_cmp_gt_f64 vcc, %2, %3\n v_min_f64 " V2(a, b) ", %2, %3\n v_max_f64 " V2(a, b) ", %2, %3\n v_cndmask_b32 " V1(a) ", %0, %1, vcc\n v_cndmask_b32 " V1(b) ", %1, %0, vcc\n"

KERNEL(k_add32, I_ADD32) KERNEL(k_xor, I_XOR) KERNEL(k_and, I_AND) KERNEL(k_mov, I_MOV) KERNEL(k_bfe, I_BFE)
KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_andor, I_AND_OR) KERNEL(k_cndvcc, I_CND_VCC) KERNEL(k_cndsgpr, I_CND_SGPR)
KERNEL(k_cmp32, I_CMP32) KERNEL(k_cmp32s, I_CMP32_SGPR) KERNEL(k_ffbh, I_FFBH) KERNEL(k_bcnt, I_BCNT)
KERNEL(k_fma32, I_FMA32) KERNEL(k_cvtf32u32, I_CVT_F32_U32) KERNEL(k_mulhi, I_MULHI) KERNEL(k_mullo, I_MULLO)
KERNEL(k_mul24, I_MUL24) KERNEL(k_mad64, I_MAD64) KERNEL(k_add64, I_ADD64) KERNEL(k_mul64, I_MUL64)
KERNEL(k_fma64, I_FMA64) KERNEL(k_min64, I_MIN64) KERNEL(k_cmp64, I_CMP64) KERNEL(k_cmp64s, I_CMP64_SGPR) KERNEL(k_cmpu64, I_CMPU64)
KERNEL(k_cvtf64u32, I_CVT_F64_U32) KERNEL(k_cvtf64f32, I_CVT_F64_F32) KERNEL(k_lshr64, I_LSHR64)
KERNEL(k_pkmov, I_PKMOV) KERNEL(k_mov64, I_MOV64)
KERNEL(k_dsr32, I_DSR32) KERNEL(k_dsr64, I_DSR64) KERNEL(k_dsw32, I_DSW32) KERNEL(k_dsw64, I_DSW64)
KERNEL(k_snop, I_SNOP) KERNEL(k_smov, I_SMOV) KERNEL(k_saveexec, I_SANDX)
#define I_CND_VCC64(a, b) "v_cndmask_b32_e64 " V1(a) ", %0, %1, vcc\n"
#define I_CMPSEL_S(a, b) "v_cmp_gt_f64 s[20:21], %2, %3\n v_cndmask_b32 " V1(a) ", %0, %1, s[20:21]\n v_cndmask_b32 " V1(b) ", %1, %0, s[20:21]\n"
#define I_CMPX_S(a, b) "v_cmp_gt_f64 s[20:21], %2, %3\n v_min_f64 " V2(a, b) ", %2, %3\n v_max_f64 " V2(a, b) ", %2, %3\n v_cndmask_b32 " V1(a) ", %0, %1, s[20:21]\n v_cndmask_b32 " V1(b) ", %1, %0, s[20:21]\n"
// alternating mask registers, as a compiler that avoids VCC would emit
#define I_CMPX_S2(a, b) "v_cmp_gt_f64 s[20:21], %2, %3\n v_cmp_lt_f64 s[22:23], %2, %3\n v_cndmask_b32 " V1(a) ", %0, %1, s[20:21]\n v_cndmask_b32 " V1(b) ", %1, %0, s[22:23]\n"
#define I_MINMAX(a, b) "v_min_f64 " V2(a, b) ", %2, %3\n v_max_f64 " V2(a, b) ", %2, %3\n"
#define I_ADDC(a, b) "v_addc_co_u32 " V1(a) ", vcc, %0, %1, vcc\n"
#define I_ADDC_S(a, b) "v_addc_co_u32 " V1(a) ", s[22:23], %0, %1, s[20:21]\n"
#define I_MIN32(a, b) "v_min_u32 " V1(a) ", %0, %1\n"
#define I_MAX3(a, b) "v_max3_u32 " V1(a) ", %0, %1, %1\n"
#define I_MED3(a, b) "v_med3_u32 " V1(a) ", %0, %1, %1\n"
#define I_CMPF32(a, b) "v_cmp_gt_f32 vcc, %4, %5\n"
#define I_LDEXP64(a, b) "v_ldexp_f64 " V2(a, b) ", %2, %0\n"
#define I_PERM(a, b) "v_perm_b32 " V1(a) ", %0, %1, %1\n"
KERNEL(k_cmpsel, I_CMPSEL) KERNEL(k_cmpx, I_CMPX) KERNEL(k_cndvcc64, I_CND_VCC64) KERNEL(k_cmpsel_s, I_CMPSEL_S)
KERNEL(k_cmpx_s, I_CMPX_S) KERNEL(k_cmpx_s2, I_CMPX_S2) KERNEL(k_minmax, I_MINMAX) KERNEL(k_addc, I_ADDC) KERNEL(k_addc_s, I_ADDC_S)
KERNEL(k_min32, I_MIN32) KERNEL(k_max3, I_MAX3) KERNEL(k_med3, I_MED3) KERNEL(k_cmpf32, I_CMPF32) KERNEL(k_ldexp64, I_LDEXP64)
KERNEL(k_perm, I_PERM)

static int g_w = 2, g_cus = 256;
static size_t g_lds_total = 160 * 1024;

using Kern = void (*)(Stamp *, double *, int);

static void run(Kern k, const char *name, int per_copy, Stamp *d_out, double *d_sink)
{
    const int iters = 4000, blocks = g_cus * g_w;
    const size_t lds = (g_lds_total / g_w) & ~(size_t)255;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<Stamp> h((size_t)blocks * 4);
    for (int rep = 0; rep < 3; ++rep) {             // the last repetition is the one read (clock settled)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d_out, d_sink, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto &s : h) {
        cyc.push_back((double)s.cycles / ((double)g_w * iters * 32 * per_copy));
        clk.push_back((double)s.cycles / (double)s.realtime * 0.1);       // GHz: realtime ticks at 100 MHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    printf("  \"%s\": {\"cycles\": %.2f, \"p10\": %.2f, \"p90\": %.2f, \"clock_ghz\": %.2f},\n", name,
           cyc[cyc.size() / 2], cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10], clk[clk.size() / 2]);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_w = atoi(argv[1]);
    if (g_w < 1 || g_w > 8) g_w = 2;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    g_cus = prop.multiProcessorCount;
    g_lds_total = prop.sharedMemPerBlock;
    Stamp *d_out;
    double *d_sink;
    (void)hipMalloc(&d_out, sizeof(Stamp) * (size_t)g_cus * 8 * 4);
    (void)hipMalloc(&d_sink, sizeof(double) * (size_t)g_cus * 8 * 256);
    printf("{\"unit\": \"shader cycles (s_memtime) per wave64 instruction per SIMD\", \"waves_per_simd\": %d, \"cus\": %d,\n",
           g_w, g_cus);
#define RUN(k, name) run(k, name, 1, d_out, d_sink)
    RUN(k_add32, "v_add_u32"); RUN(k_xor, "v_xor_b32"); RUN(k_and, "v_and_b32"); RUN(k_mov, "v_mov_b32");
    RUN(k_bfe, "v_bfe_u32"); RUN(k_lshladd, "v_lshl_add_u32"); RUN(k_andor, "v_and_or_b32");
    RUN(k_cndvcc, "v_cndmask_b32(vcc)"); RUN(k_cndsgpr, "v_cndmask_b32(sgpr pair)");
    RUN(k_cmp32, "v_cmp_gt_u32(vcc)"); RUN(k_cmp32s, "v_cmp_gt_u32(sgpr pair)");
    RUN(k_ffbh, "v_ffbh_u32"); RUN(k_bcnt, "v_bcnt_u32_b32");
    RUN(k_fma32, "v_fma_f32"); RUN(k_cvtf32u32, "v_cvt_f32_u32");
    RUN(k_mulhi, "v_mul_hi_u32"); RUN(k_mullo, "v_mul_lo_u32"); RUN(k_mul24, "v_mul_u32_u24"); RUN(k_mad64, "v_mad_u64_u32");
    RUN(k_add64, "v_add_f64"); RUN(k_mul64, "v_mul_f64"); RUN(k_fma64, "v_fma_f64"); RUN(k_min64, "v_min_f64");
    RUN(k_cmp64, "v_cmp_gt_f64(vcc)"); RUN(k_cmp64s, "v_cmp_gt_f64(sgpr pair)"); RUN(k_cmpu64, "v_cmp_gt_u64");
    RUN(k_cvtf64u32, "v_cvt_f64_u32"); RUN(k_cvtf64f32, "v_cvt_f64_f32");
    RUN(k_lshr64, "v_lshrrev_b64"); RUN(k_pkmov, "v_pk_mov_b32"); RUN(k_mov64, "v_mov_b64");
    RUN(k_dsr32, "ds_read_b32"); RUN(k_dsr64, "ds_read_b64"); RUN(k_dsw32, "ds_write_b32"); RUN(k_dsw64, "ds_write_b64");
    RUN(k_snop, "s_nop"); RUN(k_smov, "s_mov_b32"); RUN(k_saveexec, "s_and_saveexec_b64+s_mov_b64 exec (pair)");
    run(k_cmpsel, "v_cmp_gt_f64+2x v_cndmask (per triple)", 1, d_out, d_sink);
    run(k_cmpx, "compare-exchange f64 key + u32 payload (per 5 instructions)", 1, d_out, d_sink);
    RUN(k_cndvcc64, "v_cndmask_b32_e64(vcc)");
    RUN(k_cmpsel_s, "v_cmp_gt_f64+2x v_cndmask via s[20:21] (per triple)");
    RUN(k_cmpx_s, "compare-exchange via s[20:21] (per 5 instructions)");
    RUN(k_cmpx_s2, "2 cmp to 2 sgpr pairs + 2 cndmask (per 4 instructions)");
    RUN(k_minmax, "v_min_f64+v_max_f64 (per pair)");
    RUN(k_addc, "v_addc_co_u32(vcc)"); RUN(k_addc_s, "v_addc_co_u32(sgpr pairs)");
    RUN(k_min32, "v_min_u32"); RUN(k_max3, "v_max3_u32"); RUN(k_med3, "v_med3_u32"); RUN(k_cmpf32, "v_cmp_gt_f32(vcc)");
    RUN(k_ldexp64, "v_ldexp_f64"); RUN(k_perm, "v_perm_b32");
    printf("  \"end\": 0}\n");
    return 0;
}
