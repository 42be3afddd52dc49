// cmpx_bench.hip -- what a (time, payload) compare-exchange costs on gfx950, by form.  Not product code.
//   A  the kernel's form (race_isa.hip.h cmpx_time): v_cmp_gt_f64 vcc; v_min_f64; v_max_f64; 2 x v_cndmask_b32 (payload)
//   B  v_cmpx_gt_f64 (EXEC := the lanes that swap); 3 x v_swap_b32 (time low, time high, payload); s_mov_b64 exec, saved
//   C  v_cmp_gt_f64 vcc; s_and_b64 exec, saved, vcc; 3 x v_swap_b32; s_mov_b64 exec, saved
// each as layers of 8 independent comparators (the sorting network) and as a chain of 15 dependent ones (a bubble pass).
// The loop bodies are written on fixed registers (v[10:41] = 16 times, v50..v65 = 16 payloads): inline assembly cannot
// name the halves of a 64-bit operand, which form B needs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/cmpx_bench tools/cmpx_bench.hip && tools/cmpx_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// the kernels: tools/cmpx_bench_gen.py > tools/cmpx_bodies.inc
#include "cmpx_bodies.inc"

struct Case { const char *name; void (*fn)(uint32_t, double *); int comparators; };

int main()
{
    const Case cases[] = {{"A layers (cmp, min, max, 2 cndmask + 3 moves)", k_A_layer, 16}, {"A0 layers (the 3 moves alone)", k_A0_layer, 16},
                          {"B layers (cmpx, 3 swaps, restore)", k_B_layer, 16}, {"C layers (cmp, s_and exec, 3 swaps, restore)", k_C_layer, 16},
                          {"D layers (cmpx, payload swap, restore, min, max + 2 moves)", k_D_layer, 16},
                          {"A chain", k_A_chain, 15}, {"A0 chain", k_A0_chain, 15}, {"B chain", k_B_chain, 15}, {"C chain", k_C_chain, 15},
                          {"D chain", k_D_chain, 15}};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    CHECK(hipMalloc(&out, 64 * sizeof(double) * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t iters = 40000;
    for (int W = 2; W <= 4; ++W) {
        for (const Case &c : cases) {
            // W blocks of 256 threads per CU (LDS reservation keeps it at W)
            const size_t lds = 160 * 1024 / W - 1024;
            CHECK(hipFuncSetAttribute((const void *)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(c.fn, dim3(cus * W), dim3(256), lds, 0, 1000u, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(cus * W), dim3(256), lds, 0, iters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            double h[8];
            CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            const double cyc = 1024.0 * 2.4e9 * ms * 1e-3 / ((double)cus * 4 * W * iters * c.comparators);
            printf("waves/SIMD %d  %-48s %.2f cycles per comparator per SIMD   (check %.3f %.3f)\n", W, c.name, cyc, h[0], h[5]);
        }
    }
    return 0;
}
