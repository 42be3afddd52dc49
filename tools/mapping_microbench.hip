// mapping_microbench.hip -- evidence for DESIGN.md section 3: what one ORDERING STEP of a 20-car field costs under
// the two candidate lane mappings on gfx950.  Not product code.
//
//   A  lane-per-car  : 3 races per wavefront (lanes 0..59), every lane holds one car; rank by counting over the 19
//                      other cars of the race, operands fetched with ds_bpermute (__shfl); then the sorted order is
//                      formed with one ds_permute push.  This is the "wavefront shuffles and ballots" mapping.
//   B  lane-per-race : 64 races per wavefront, the 20 cars of a race in VGPRs; 97-comparator merge-exchange network
//                      on (time, tie-break word) -- network_sort of csrc/race_kernel_reg.hip.h, the product's step.
//
// Both sort by (time, tie word) = Python's stable sort key.  Each iteration perturbs the times (as a lap does) and
// re-sorts.  Output: ordering steps per second (one step = one race's field sorted once) for each mapping.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I monte_carlo_gp_amd/csrc -o /tmp/mapping_microbench tools/mapping_microbench.hip
#include "race_kernel_reg.hip.h"

#include <cstdio>
#include <vector>

constexpr int N = 20;

__global__ void __launch_bounds__(256) lane_per_car(double *out, int iters)
{
    const int lane = threadIdx.x & 63;
    const int race = lane / N;              // 0..2 (lanes 60..63 idle)
    const int car = lane - race * N;
    const bool live = lane < 3 * N;
    const int base = race * N;
    double t = 100.0 + 0.37 * car + 1e-3 * (blockIdx.x % 97) + 1e-4 * (threadIdx.x >> 6);
    uint32_t key = (uint32_t)car << 27;
    uint32_t h = (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        t += 90.0 + (double)(h >> 8) * (1.0 / 16777216.0);          // "lap time"
        // rank by counting: 19 rotations, operands through the LDS crossbar (ds_bpermute)
        int rank = 0;
#pragma unroll
        for (int r = 1; r < N; ++r) {
            int src = car + r;
            src = base + (src >= N ? src - N : src);
            const double tj = __shfl(t, src, 64);
            const uint32_t kj = __shfl(key, src, 64);
            rank += (tj < t) || (tj == t && kj < key);
        }
        // sorted order: push (t, key) to lane base + rank (ds_permute), as the next phase needs "car ahead"
        const int dst = live ? base + rank : lane;
        const int lo = __builtin_amdgcn_ds_permute(dst << 2, __double2loint(t));
        const int hi = __builtin_amdgcn_ds_permute(dst << 2, __double2hiint(t));
        const uint32_t k2 = (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)key);
        t = __hiloint2double(hi, lo);
        key = k2;
    }
    if (live) out[(size_t)blockIdx.x * 256 + threadIdx.x] = t + key;
}

__global__ void __launch_bounds__(256, 2) lane_per_race(double *out, int iters)
{
    double cum[N];
    uint32_t pk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cum[i] = 100.0 + 0.37 * i + 1e-3 * (blockIdx.x % 97) + 1e-4 * threadIdx.x;
        pk[i] = (uint32_t)i << 27;
    }
    uint32_t h = (uint32_t)(blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            h = h * 1664525u + 1013904223u;
            cum[i] += 90.0 + (double)(h >> 8) * (1.0 / 16777216.0);
        }
        mcgp::network_sort<N>(cum, pk);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += cum[i] * (i + 1) + pk[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int blocks = 256 * 8, iters = 2000;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float msA = 0, msB = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(lane_per_car, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&msA, a, b);
        hipEventRecord(a);
        hipLaunchKernelGGL(lane_per_race, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&msB, a, b);
    }
    const double waves = (double)blocks * 4;
    const double stepsA = waves * 3 * iters / (msA * 1e-3);      // 3 races per wave
    const double stepsB = waves * 64 * iters / (msB * 1e-3);     // 64 races per wave
    printf("{\"field\": %d, \"lane_per_car\": {\"ms\": %.3f, \"ordering_steps_per_s\": %.4g}, "
           "\"lane_per_race\": {\"ms\": %.3f, \"ordering_steps_per_s\": %.4g}, \"ratio_race_over_car\": %.2f}\n",
           N, msA, stepsA, msB, stepsB, stepsB / stepsA);
    return 0;
}
