// emu_kernel.cpp -- DEBUGGING build of the register kernel's source for the host (not product code, not a
// fallback: nothing in monte_carlo_gp_amd/ can reach it).  Compiles csrc/race_kernel_reg.hip.h with g++ through
// the stand-in <hip/hip_runtime.h> of this directory and runs the kernel's three phases (load tables, simulate,
// flush) for the threads of one block one after another, so a kernel edit can be compared with the oracle on
// the CPU before a GPU run.  tests/test_kernel_host_build.py.
//   g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -Itools/emu -o tools/emu/libmcgp_emu.so tools/emu/emu_kernel.cpp
#include <cstdint>
// statistics hook: counts[what] += value, calls[what] += 1
extern "C" unsigned long long emu_stat_sum[16], emu_stat_calls[16];
#if !defined(EMU_PART) || EMU_PART == 0
unsigned long long emu_stat_sum[16], emu_stat_calls[16];
#endif
#define MCGP_STAT(what, value) (emu_stat_sum[(what)] += (unsigned long long)(value), emu_stat_calls[(what)] += 1)
// per-(simulation, lap, pass) trace of the number of overtake candidates (tools only): emu_trace_buf[sim][lap][pass] = n + 1
extern "C" unsigned char *emu_trace_buf;
extern "C" unsigned long long emu_trace_sims, emu_trace_laps;
#if !defined(EMU_PART) || EMU_PART == 0
unsigned char *emu_trace_buf = nullptr;
unsigned long long emu_trace_sims = 0, emu_trace_laps = 0;
#endif
#define MCGP_TRACE_PASS(sim, lap, pass, n_cand)                                                              \
    do {                                                                                                     \
        if (emu_trace_buf && (unsigned long long)(sim) < emu_trace_sims && (unsigned long long)(lap) < emu_trace_laps) \
            emu_trace_buf[((unsigned long long)(sim) * emu_trace_laps + (lap)) * 3 + (pass)] = (unsigned char)((n_cand) + 1); \
    } while (0)
#define MCGP_COOPERATIVE_EVENTS 0        // threads run one at a time here: every lane handles its own event
#include "race_isa_host.h"

#include "../../monte_carlo_gp_amd/csrc/params_build.h"
#include "../../monte_carlo_gp_amd/csrc/race_kernel_reg.hip.h"
#include "../../monte_carlo_gp_amd/csrc/normal53_table.h"

#include <vector>

#if !defined(EMU_PART) || EMU_PART == 0
emu_dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};
namespace mcgp {
alignas(16) unsigned char smem[1 << 20];
}
#endif

#if !defined(EMU_PART) || EMU_PART == 0
// the kernel's inverse-normal transform (race_common.hip.h) on the host, for the known-answer test of its tail cells
extern "C" float emu_normal_from_u32(uint32_t w)
{
    return mcgp::normal_from_u32(w, reinterpret_cast<const float4 *>(mcgp_normal_table_bits));
}
// ... its binary64 transform of the reference-width build ...
extern "C" double emu_normal53(uint32_t w, uint32_t companion)
{
    return mcgp::normal53(w, companion, reinterpret_cast<const double *>(mcgp_normal53_table_bits));
}
// ... and the once-per-race retirement draw with the thresholds the parameter block carries (params_build.h)
extern "C" uint32_t emu_draw_retirement_lap(uint32_t w, double p, int total_laps)
{
    return mcgp::draw_retirement_lap(w, mcgp::threshold(p), total_laps);
}
extern "C" unsigned long long emu_threshold(double p) { return mcgp::threshold(p); }
extern "C" unsigned long long emu_threshold53(double p) { return mcgp::threshold53(p); }
extern "C" unsigned long long emu_survival64(double p) { return mcgp::survival64(p); }
#endif

// The field sizes are spread over EMU_PARTS translation units (compiled in parallel by tests/kernel_host_build.py):
// unit EMU_PART holds the sizes n with n % EMU_PARTS == EMU_PART and exports emu_run_part<EMU_PART>; unit 0 also
// holds the dispatcher emu_run.
#ifndef EMU_PARTS
#define EMU_PARTS 1
#define EMU_PART 0
#endif
#ifndef EMU_ALL_SIZES
#define EMU_ALL_SIZES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) \
    X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#endif

template <int N>
static int run_size(const mcgp::KParams &kp, uint64_t n_sims, uint64_t sim_offset, uint64_t seed, unsigned long long *hist,
                    uint8_t *orders, const uint8_t *fixed_grid)
{
    if constexpr (N % EMU_PARTS == EMU_PART) {
        // the reference-width build (mcgp_config.deviates = MCGP_DEVIATES_53), in its own geometry (WideGeo: blocks of 8 waves,
        // rows of the binary64 inverse-normal table in LDS)
        if (kp.wide) {
            using G = mcgp::WideGeo<N>;
            constexpr uint32_t B = G::B;
            const double *tab = reinterpret_cast<const double *>(mcgp_normal53_table_bits);
            const uint32_t n_chunks = (uint32_t)((n_sims + 63) / 64);
            for (uint32_t t = 0; t < B; ++t) mcgp::reg_load_tables<N, G>(&kp, mcgp::smem, t, tab);
            std::vector<uint32_t> retire_ws((size_t)(N + 1) * B);
            for (uint32_t t = 0; t < B; ++t)
                mcgp::reg_simulate<N, true, G>(&kp, mcgp::smem, t, nullptr, n_sims, sim_offset, (uint32_t)seed,
                                               (uint32_t)(seed >> 32), orders, fixed_grid, n_chunks, retire_ws.data(), B, 0u, tab);
            for (uint32_t t = 0; t < B; ++t) mcgp::reg_flush_hist<N, G>(mcgp::smem, t, hist);
            return 0;
        }
        // one block at a time; inside a block the three phases of the kernel run for every "thread" in turn
        constexpr uint32_t B = mcgp::RegGeo<N>::B;
        const uint32_t n_chunks = (uint32_t)((n_sims + 63) / 64);
        for (uint32_t t = 0; t < B; ++t) mcgp::reg_load_tables<N>(&kp, mcgp::smem, t);
        std::vector<uint32_t> retire_ws((size_t)(N + 1) * B);        // the lanes' retirement lists (device memory on the GPU)
        for (uint32_t t = 0; t < B; ++t)
            mcgp::reg_simulate<N>(&kp, mcgp::smem, t, nullptr, n_sims, sim_offset, (uint32_t)seed, (uint32_t)(seed >> 32),
                                  orders, fixed_grid, n_chunks, retire_ws.data(), B, 0u);
        for (uint32_t t = 0; t < B; ++t) mcgp::reg_flush_hist<N>(mcgp::smem, t, hist);
        return 0;
    } else {
        return -2;
    }
}

#define EMU_CAT2(a, b) a##b
#define EMU_CAT(a, b) EMU_CAT2(a, b)
extern "C" int EMU_CAT(emu_run_part, EMU_PART)(const mcgp::KParams *kp, uint32_t n, uint64_t n_sims, uint64_t sim_offset,
                                               uint64_t seed, unsigned long long *hist, uint8_t *orders,
                                               const uint8_t *fixed_grid)
{
    switch (n) {
#define X(N_) case N_: return run_size<N_>(*kp, n_sims, sim_offset, seed, hist, orders, fixed_grid);
        EMU_ALL_SIZES(X)
#undef X
        default: return -1;
    }
}

#if EMU_PART == 0
extern "C" int emu_run_part1(const mcgp::KParams *, uint32_t, uint64_t, uint64_t, uint64_t, unsigned long long *, uint8_t *, const uint8_t *);
extern "C" int emu_run_part2(const mcgp::KParams *, uint32_t, uint64_t, uint64_t, uint64_t, unsigned long long *, uint8_t *, const uint8_t *);
extern "C" int emu_run_part3(const mcgp::KParams *, uint32_t, uint64_t, uint64_t, uint64_t, unsigned long long *, uint8_t *, const uint8_t *);

extern "C" int emu_run(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs, uint32_t n,
                       uint64_t n_sims, uint64_t sim_offset, uint64_t seed, unsigned long long *hist,
                       uint8_t *orders, const uint8_t *fixed_grid, const char **err)
{
    static mcgp::KParams kp;
    static const char *none = "";
    *err = none;
    const int rc = mcgp::build_params(cfg, drv, grid_probs, n, &kp, err);
    if (rc != MCGP_OK) return rc;
    if (!mcgp::reg_kernel_serves(kp)) { *err = "served by the generic kernel (reg_kernel_serves)"; return -100; }
    threadIdx = {0, 0, 0};
    blockIdx = {0, 0, 0};
    blockDim = {1, 1, 1};
    gridDim = {1, 1, 1};
    int r = -1;
    switch (EMU_PARTS > 1 ? n % EMU_PARTS : 0) {
        case 0: r = emu_run_part0(&kp, n, n_sims, sim_offset, seed, hist, orders, fixed_grid); break;
#if EMU_PARTS == 4
        case 1: r = emu_run_part1(&kp, n, n_sims, sim_offset, seed, hist, orders, fixed_grid); break;
        case 2: r = emu_run_part2(&kp, n, n_sims, sim_offset, seed, hist, orders, fixed_grid); break;
        case 3: r = emu_run_part3(&kp, n, n_sims, sim_offset, seed, hist, orders, fixed_grid); break;
#endif
    }
    if (r != 0) *err = "no register instantiation for this field size";
    return r;
}
#endif
