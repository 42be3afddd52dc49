// mcgp_hip.hip -- C-ABI (include/mcgp.h) over the gfx950 race kernel.
//
// Host side of the drop-in boundary: validates arguments, folds the reference's
// per-race constants into the kernel's parameter block, owns one cached context
// per HIP device (parameter buffer, scratch histogram, events) and launches
// race_kernel.  No CPU compute path exists here: without a HIP device every
// compute entry point returns MCGP_E_NO_DEVICE.
#include "../../include/mcgp.h"
#include "params_build.h"
#include "race_kernel.hip.h"
#include "race_kernel_reg.hip.h"

#define MCGP_FE_FN __host__ __device__ static inline
#include "frontend_exp.h"
#include "elo_update.h"
#include "normal53_table.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? MCGP_E_NOMEM : MCGP_E_HIP,                    \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

// Field sizes with a register-resident instantiation: every n the ABI admits (1..32).  The generic LDS kernel
// (race_kernel.hip.h) is kept as an independently written second implementation, selected with
// MCGP_FORCE_GENERIC=1: tests run both and require identical results.
#ifdef MCGP_ONLY_N20      // diagnostic builds (tools/ablate.sh)
#define MCGP_REG_SIZES(X) X(20)
#elif defined(MCGP_ONLY_N)
#define MCGP_REG_SIZES(X) X(MCGP_ONLY_N)
#else
#define MCGP_REG_SIZES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) \
    X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#endif

// The reference-width build (mcgp_config.deviates = MCGP_DEVIATES_53) is compiled for the same field sizes (reg_inst.hip).
#define MCGP_WIDE_SIZES(X) MCGP_REG_SIZES(X)

// Grid-probability front end (reference src/elo.py:124-141, src/predictor.py:321-407): one thread per driver row.
// in = [rating | teammate_delta | form_score | circuit_affinity], n doubles each.  Thread 0 computes the n pole
// probabilities (a sequential softmax + renormalisation in the reference's order), then every thread builds its
// row.  out may point straight into a parameter block's grid_probs slot.
__global__ void __launch_bounds__(mcgp::kMaxCars)
grid_probs_kernel(const double *__restrict__ in, const int32_t *__restrict__ penalty, int n, double *__restrict__ out)
{
    __shared__ double pole[mcgp::kMaxCars];
    if (threadIdx.x == 0) mcgp_fe_pole_probs(in, in + n, n, pole);
    __syncthreads();
    const int d = threadIdx.x;
    if (d < n) {
        double row[mcgp::kMaxCars], tmp[mcgp::kMaxCars];
        mcgp_fe_grid_row(pole[d], in[2 * n + d], in[3 * n + d], penalty[d], n, row, tmp);
        for (int s = 0; s < n; ++s) out[(size_t)d * n + s] = row[s];
    }
}

// A season of Elo updates (reference src/elo.py:45-122) in one block of 32 x 32 threads with the ratings in LDS.
// Per event: thread (a, b) computes the term of entry a against entry b (csrc/elo_update.h; the one expensive part,
// a 10^x and a division, all m (m - 1) of them at once); a barrier; thread (a, 0) adds entry a's terms up in list
// order -- the reference's inner loop, so the sum has its bits --; a barrier; the deltas are applied.
__global__ void __launch_bounds__(mcgp::kMaxCars * mcgp::kMaxCars)
elo_season_kernel(int n, int n_events, const int32_t *__restrict__ kind, const double *__restrict__ k,
                  const uint32_t *__restrict__ count, const uint8_t *__restrict__ who, const double *__restrict__ value,
                  double *__restrict__ ratings, double *__restrict__ after)
{
    constexpr int M = mcgp::kMaxCars;
    __shared__ double r[2 * M];                           // [kind][driver], row stride n
    __shared__ double term[M][M + 1];
    __shared__ double v[M];
    __shared__ uint8_t w[M];
    const int t = threadIdx.x, a = t / M, b = t % M;
    if (t < 2 * n) r[t] = ratings[t];
    __syncthreads();
    for (int e = 0; e < n_events; ++e) {
        const int m = (int)count[e];
        double *row = r + (kind[e] ? n : 0);
        if (t < m) {
            w[t] = who[(size_t)e * n + t];
            v[t] = value[(size_t)e * n + t];
        }
        __syncthreads();
        if (m >= 2 && a < m && b < m && a != b) term[a][b] = mcgp_elo_term(row[w[a]], v[a], row[w[b]], v[b], k[e], m);
        __syncthreads();
        if (m >= 2 && b == 0 && a < m) {
            double delta = 0.0;
            for (int j = 0; j < m; ++j)
                if (j != a) delta = delta + term[a][j];
            row[w[a]] = row[w[a]] + delta;                // (the terms were all computed from the ratings before the event)
        }
        __syncthreads();
        if (after && t < 2 * n) after[(size_t)e * 2 * n + t] = r[t];
    }
    if (t < 2 * n) ratings[t] = r[t];
}

constexpr int kParamSlots = 4;
constexpr int kSlotReaders = 8;        // streams with a launch in flight on one parameter block
constexpr int kStreamTimers = 8;       // streams whose most recent call keeps its own timing events
constexpr int kBatchTimer = -2;        // DeviceCtx::last_timer after mcgp_run_batch: the call's own pair of events

struct DeviceCtx {
    std::mutex mu;
    bool ready = false;
    int cu_count = 0;
    size_t lds_per_block = 0;
    // Parameter blocks: a small ring so that a launch never rewrites a block an
    // earlier, still running launch reads; an unchanged problem re-uses its block
    // with no upload at all (bench.py's steady state).
    struct Slot {
        mcgp::KParams *dev = nullptr;
        mcgp::KParams *host = nullptr;        // pinned copy of what `dev` holds
        // one completion event per STREAM that has launched on this block: eviction waits for every one of
        // them (a single event re-recorded by the latest stream would forget a reader still running elsewhere)
        struct Reader {
            hipStream_t stream = nullptr;
            hipEvent_t done = nullptr;        // recorded after that stream's last launch reading `dev`
            bool live = false;
            uint64_t seq = 0;                 // order of the last launch through this entry
        } reader[kSlotReaders];
        hipEvent_t uploaded = nullptr;        // recorded after the upload of `dev`; other streams wait on it
        hipStream_t upload_stream = nullptr;
        bool used = false;
        bool shareable = false;               // false: the device copy differs from `host` (matrix written by the front end)
    } slot[kParamSlots];
    int next_slot = 0;
    unsigned long long *d_hist = nullptr;     // scratch for the host-buffer entry points
    uint8_t *d_grid = nullptr;                // fixed grid for mcgp_simulate_race
    uint8_t *d_order1 = nullptr;              // one finishing order (mcgp_simulate_race)
    double *d_fe_in = nullptr;                // front end: 4 x 32 doubles in, 32 penalties, n x n matrix out
    int32_t *d_fe_pen = nullptr;
    double *d_fe_out = nullptr;
    double *d_norm53 = nullptr;               // binary64 inverse-normal table of the reference-width build (50 KB)
    uint8_t *d_orders = nullptr;              // staging for mcgp_run(orders_out), grown on demand, kept
    size_t d_orders_bytes = 0;
    // timing events per stream (most recent call on that stream), so that calls on different streams of one
    // device do not re-record each other's events
    struct Timer {
        hipStream_t stream = nullptr;
        hipEvent_t start = nullptr, stop = nullptr;
        uint32_t *d_ticket = nullptr;       // the work counter of this stream's launches (race_kernel_reg.hip.h, phase 2)
        uint32_t *d_retire = nullptr;       // the lanes' retirement lists of this stream's launches (grow-only)
        size_t retire_bytes = 0;
        bool used = false;
        uint64_t seq = 0;
    } timer[kStreamTimers];
    uint64_t timer_seq = 0;
    int last_timer = -1;
    unsigned char *d_elo = nullptr;         // scratch of mcgp_elo_season (grow-only)
    size_t elo_bytes = 0;
    unsigned char *d_batch = nullptr;       // mcgp_run_batch: parameter blocks, items, histograms, ticket (grow-only)
    size_t batch_bytes = 0;
    uint32_t *d_batch_retire = nullptr;     // ... and the lanes' retirement lists
    size_t batch_retire_bytes = 0;
    hipEvent_t batch_start = nullptr, batch_stop = nullptr;     // ... and the timing events of the last batch call
    uint32_t last_grid = 0, last_block = 0, last_lds = 0;
    char last_kernel[48] = "";
};

constexpr int kMaxDevices = 64;
DeviceCtx g_ctx[kMaxDevices];

int device_count_nothrow()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Validates the device index and returns its context; nothing is initialised yet.
int find_ctx(int device, DeviceCtx **out)
{
    const int nd = device_count_nothrow();
    if (nd <= 0) return fail(MCGP_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= nd || device >= kMaxDevices)
        return fail(MCGP_E_NO_DEVICE, "device index out of range");
    *out = &g_ctx[device];
    return MCGP_OK;
}

void release_ctx(DeviceCtx &c)
{
    for (auto &sl : c.slot) {
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.host) (void)hipHostFree(sl.host);
        for (auto &rd : sl.reader)
            if (rd.done) (void)hipEventDestroy(rd.done);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        sl = DeviceCtx::Slot{};
    }
    if (c.d_hist) (void)hipFree(c.d_hist);
    if (c.d_grid) (void)hipFree(c.d_grid);
    if (c.d_order1) (void)hipFree(c.d_order1);
    if (c.d_fe_in) (void)hipFree(c.d_fe_in);
    if (c.d_fe_pen) (void)hipFree(c.d_fe_pen);
    if (c.d_fe_out) (void)hipFree(c.d_fe_out);
    c.d_fe_in = c.d_fe_out = nullptr;
    c.d_fe_pen = nullptr;
    if (c.d_orders) (void)hipFree(c.d_orders);
    if (c.d_norm53) (void)hipFree(c.d_norm53);
    c.d_norm53 = nullptr;
    if (c.d_elo) (void)hipFree(c.d_elo);
    c.d_elo = nullptr;
    c.elo_bytes = 0;
    if (c.d_batch) (void)hipFree(c.d_batch);
    if (c.d_batch_retire) (void)hipFree(c.d_batch_retire);
    if (c.batch_start) (void)hipEventDestroy(c.batch_start);
    if (c.batch_stop) (void)hipEventDestroy(c.batch_stop);
    c.batch_start = c.batch_stop = nullptr;
    c.d_batch = nullptr;
    c.d_batch_retire = nullptr;
    c.batch_bytes = c.batch_retire_bytes = 0;
    for (auto &t : c.timer) {
        if (t.start) (void)hipEventDestroy(t.start);
        if (t.stop) (void)hipEventDestroy(t.stop);
        if (t.d_ticket) (void)hipFree(t.d_ticket);
        if (t.d_retire) (void)hipFree(t.d_retire);
        t = DeviceCtx::Timer{};
    }
    c.last_timer = -1;
    c.d_hist = nullptr;
    c.d_grid = c.d_order1 = c.d_orders = nullptr;
    c.d_orders_bytes = 0;
}

int init_ctx_body(int device, DeviceCtx &c)
{
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c.cu_count = prop.multiProcessorCount;
    c.lds_per_block = prop.sharedMemPerBlock;       // 160 KiB on gfx950
    if (const char *e = std::getenv("MCGP_LDS_PER_BLOCK")) {      // tests: a device / runtime that offers less LDS per block
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        if (v >= 16384 && v < c.lds_per_block) c.lds_per_block = (size_t)v;
    }
    for (auto &sl : c.slot) {
        HIP_TRY(hipMalloc(&sl.dev, sizeof(mcgp::KParams)));
        HIP_TRY(hipHostMalloc(&sl.host, sizeof(mcgp::KParams)));
        HIP_TRY(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
    }
    HIP_TRY(hipMalloc(&c.d_hist, sizeof(unsigned long long) * MCGP_MAX_CARS * MCGP_MAX_CARS));
    HIP_TRY(hipMalloc(&c.d_grid, MCGP_MAX_CARS));
    HIP_TRY(hipMalloc(&c.d_order1, MCGP_MAX_CARS));
    HIP_TRY(hipMalloc(&c.d_fe_in, sizeof(double) * 4 * MCGP_MAX_CARS));
    HIP_TRY(hipMalloc(&c.d_fe_pen, sizeof(int32_t) * MCGP_MAX_CARS));
    HIP_TRY(hipMalloc(&c.d_fe_out, sizeof(double) * MCGP_MAX_CARS * MCGP_MAX_CARS));
    HIP_TRY(hipMalloc(&c.d_norm53, sizeof(mcgp_normal53_table_bits)));
    HIP_TRY(hipMemcpy(c.d_norm53, mcgp_normal53_table_bits, sizeof(mcgp_normal53_table_bits), hipMemcpyHostToDevice));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&mcgp::race_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_per_block));
    // (the register kernels raise their dynamic-LDS limit when they are selected: launch())
    return MCGP_OK;
}

// First use of a device: the CALLER HOLDS c.mu, so two threads making their first call on one
// device cannot both initialise it (nor one launch into buffers the other is still creating).
// A HIP failure part-way releases what was allocated; the next call starts from scratch.
int ensure_ctx_locked(int device, DeviceCtx &c)
{
    if (c.ready) return MCGP_OK;
    const int rc = init_ctx_body(device, c);
    if (rc != MCGP_OK) {
        release_ctx(c);
        return rc;
    }
    c.ready = true;
    return MCGP_OK;
}

int build_params(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs, uint32_t n,
                 mcgp::KParams *kp)
{
    const char *err = "";
    const int rc = mcgp::build_params(cfg, drv, grid_probs, n, kp, &err);
    return rc == MCGP_OK ? rc : fail(rc, err);
}

using KernelFn = void (*)(const mcgp::KParams *, uint64_t, uint64_t, uint32_t, uint32_t, unsigned long long *,
                          uint8_t *, const uint8_t *, uint32_t, uint32_t *, uint32_t *);

}  // namespace

// The register-resident instantiations are compiled one per translation unit (reg_inst.hip,
// -DMCGP_INST_N=<n>) so that the build runs in parallel; here they are only declared.
namespace mcgp {
#define X(N_) extern template __global__ void race_kernel_reg<N_>(const KParams *, uint64_t, uint64_t, uint32_t, uint32_t, \
                                                                  unsigned long long *, uint8_t *, const uint8_t *, uint32_t, \
                                                                  uint32_t *, uint32_t *);
MCGP_REG_SIZES(X)
#undef X
#define X(N_) extern template __global__ void race_kernel_reg<N_, kSmallBlockWaves>(const KParams *, uint64_t, uint64_t, uint32_t, \
                                                                                    uint32_t, unsigned long long *, uint8_t *,   \
                                                                                    const uint8_t *, uint32_t, uint32_t *, uint32_t *);
MCGP_REG_SIZES(X)
#undef X
#define X(N_) extern template __global__ void race_kernel_reg_wide<N_>(const KParams *, uint64_t, uint64_t, uint32_t, uint32_t, \
                                                                       unsigned long long *, uint8_t *, const uint8_t *, uint32_t, \
                                                                       uint32_t *, uint32_t *, const double *);
MCGP_WIDE_SIZES(X)
#undef X
#define X(N_) extern template __global__ void race_kernel_reg_batch<N_>(const KParams *, const BatchItem *, uint32_t, uint64_t, \
                                                                        unsigned long long *, uint32_t, uint32_t *, uint32_t *);
MCGP_REG_SIZES(X)
#undef X
}  // namespace mcgp

namespace {

using BatchKernelFn = void (*)(const mcgp::KParams *, const mcgp::BatchItem *, uint32_t, uint64_t, unsigned long long *,
                               uint32_t, uint32_t *, uint32_t *);
using WideKernelFn = void (*)(const mcgp::KParams *, uint64_t, uint64_t, uint32_t, uint32_t, unsigned long long *,
                              uint8_t *, const uint8_t *, uint32_t, uint32_t *, uint32_t *, const double *);
WideKernelFn select_wide_kernel(uint32_t n)
{
    switch (n) {
#define X(N_) case N_: return &mcgp::race_kernel_reg_wide<N_>;
        MCGP_WIDE_SIZES(X)
#undef X
        default: return nullptr;
    }
}

BatchKernelFn select_batch_kernel(uint32_t n)
{
    switch (n) {
#define X(N_) case N_: return &mcgp::race_kernel_reg_batch<N_>;
        MCGP_REG_SIZES(X)
#undef X
        default: return nullptr;
    }
}

// the register kernel of a field size in its small block shape (kSmallBlockWaves)
KernelFn select_small_kernel(uint32_t n)
{
    switch (n) {
#define X(N_) case N_: return &mcgp::race_kernel_reg<N_, mcgp::kSmallBlockWaves>;
        MCGP_REG_SIZES(X)
#undef X
        default: return nullptr;
    }
}

KernelFn select_kernel(const mcgp::KParams &kp, bool *is_reg)
{
    *is_reg = false;
    const uint32_t n = (uint32_t)kp.n;
    const char *force = std::getenv("MCGP_FORCE_GENERIC");
    if (force && force[0] == '1') return &mcgp::race_kernel;
    if (!mcgp::reg_kernel_serves(kp)) return &mcgp::race_kernel;      // lap times near zero, values near the ends of binary64
    switch (n) {
#define X(N_) case N_: *is_reg = true; return &mcgp::race_kernel_reg<N_>;
        MCGP_REG_SIZES(X)
#undef X
        default: return &mcgp::race_kernel;
    }
}

// Launch geometry: persistent blocks striding over batches of `block` simulations.  The register kernel's
// block size and LDS footprint are compile-time functions of the field size (RegGeo<N>, shared with the
// kernel); the number of blocks per CU follows from LDS and the kernel's register allocation.
void launch_geometry(const DeviceCtx &c, uint32_t n, bool is_reg, KernelFn kernel, uint64_t n_sims, uint32_t *grid,
                     uint32_t *block, uint32_t *lds, int reg_waves = 0 /* 0: the register kernel's default block shape */,
                     bool wide = false /* the reference-width build: its block also holds table rows (WideGeo) */, int total_laps = 0)
{
    // waves per CU the kernel's register allocation admits: 4 SIMDs x floor(512 / VGPRs, granule 8), at most 8 each
    int reg_cap = 8;
    {
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel)) == hipSuccess && attr.numRegs > 0) {
            const int alloc = ((attr.numRegs + 7) / 8) * 8;
            int per_simd = 512 / alloc;
            if (per_simd > 8) per_simd = 8;
            if (per_simd < 1) per_simd = 1;
            reg_cap = 4 * per_simd;
        }
    }
    int waves = 1, blocks_per_cu = 1;
    size_t bytes = 0;
    if (is_reg) {
        waves = reg_waves ? reg_waves : mcgp::reg_block_waves((int)n);
        bytes = wide ? mcgp::wide_launch_lds_bytes((int)n, waves, total_laps)
                     : mcgp::shared_lds_bytes_reg((int)n) + (size_t)waves * 64 * mcgp::per_thread_lds_bytes_reg((int)n);
    } else {
        const size_t per_wave = 64 * mcgp::per_thread_lds_bytes((int)n);
        waves = (int)((c.lds_per_block - mcgp::kSharedTableBytes) / per_wave);
        if (waves > 8) waves = 8;
        if (waves < 1) waves = 1;
        if (const char *e = std::getenv("MCGP_WAVES_PER_BLOCK")) {          // tuning / diagnostics (generic kernel only)
            const int w = std::atoi(e);
            if (w >= 1 && w <= waves) waves = w;
        }
        uint32_t threads = (uint32_t)waves * 64u;
        if (n_sims < threads) threads = (uint32_t)(((n_sims + 63) / 64) * 64);
        if (threads == 0) threads = 64;
        waves = (int)(threads / 64);
        bytes = mcgp::kSharedTableBytes + (size_t)threads * mcgp::per_thread_lds_bytes((int)n);
    }
    blocks_per_cu = (int)(c.lds_per_block / bytes);
    if (blocks_per_cu * waves > reg_cap) blocks_per_cu = reg_cap / waves;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (const char *e = std::getenv("MCGP_MAX_BLOCKS_PER_CU")) {
        const int m = std::atoi(e);
        if (m >= 1 && m < blocks_per_cu) blocks_per_cu = m;
    }
    const uint32_t threads = (uint32_t)waves * 64u;
    const uint64_t n_batches = (n_sims + threads - 1) / threads;
    uint64_t g = (uint64_t)c.cu_count * (uint64_t)blocks_per_cu;
    if (g > n_batches) g = n_batches;
    if (g < 1) g = 1;
    *grid = (uint32_t)g;
    *block = threads;
    *lds = (uint32_t)bytes;
}

// Most simulations one kernel launch may take.  The per-block LDS histogram counts in uint32 and a cell
// can receive at most one count per simulation the block runs, so keeping the whole launch below 2^32
// simulations rules out a silent wrap; longer runs are split into several launches on the same stream
// (results do not depend on the split: draws are addressed by global simulation id).
uint64_t max_sims_per_launch()
{
    uint64_t cap = 0xFFFFFE00ull;                                       // < 2^32, a multiple of 512
    if (const char *e = std::getenv("MCGP_MAX_SIMS_PER_LAUNCH")) {      // tests exercise the split with a small cap
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        if (v >= 1 && v < cap) cap = v;
    }
    return cap;
}

// Front-end inputs of one call, already on the device (c.d_fe_in / c.d_fe_pen), or null.
struct FrontEnd {
    bool on = false;
};

// The timing events and work counter of `stream` (its most recent call); the least recently used entry is recycled.
int claim_timer(DeviceCtx &c, hipStream_t stream, int *out)
{
    int ti = -1;
    for (int i = 0; i < kStreamTimers; ++i)
        if (c.timer[i].used && c.timer[i].stream == stream) { ti = i; break; }
    if (ti < 0) {
        ti = 0;
        for (int i = 0; i < kStreamTimers; ++i) {
            if (!c.timer[i].used) { ti = i; break; }
            if (c.timer[i].seq < c.timer[ti].seq) ti = i;
        }
        if (!c.timer[ti].d_ticket) {
            // events and counter are created into locals and committed together: a failure part-way leaves the entry
            // empty (not an entry with events and a NULL counter for the next launch to hand to the kernel)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            uint32_t *tk = nullptr;
            hipError_t err = hipEventCreate(&e0);
            if (err == hipSuccess) err = hipEventCreate(&e1);
            if (err == hipSuccess) err = hipMalloc(&tk, sizeof(uint32_t));
            if (err != hipSuccess) {
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                return fail(err == hipErrorOutOfMemory ? MCGP_E_NOMEM : MCGP_E_HIP,
                            std::string("stream timer / work counter: ") + hipGetErrorString(err));
            }
            c.timer[ti].start = e0;
            c.timer[ti].stop = e1;
            c.timer[ti].d_ticket = tk;
        } else if (c.timer[ti].used) {
            // recycled from another stream: its last launch may still be claiming work from the entry's counter
            HIP_TRY(hipEventSynchronize(c.timer[ti].stop));
        }
        c.timer[ti].stream = stream;
        c.timer[ti].used = true;
    }
    *out = ti;
    return MCGP_OK;
}

int launch(DeviceCtx &c, const mcgp::KParams &kp, uint64_t n_sims, uint64_t sim_offset, uint64_t seed,
           hipStream_t stream, unsigned long long *d_hist, uint8_t *d_orders, const uint8_t *d_fixed_grid,
           const FrontEnd &fe = FrontEnd())
{
    if (n_sims == 0) return MCGP_OK;
    bool is_reg = false;
    KernelFn kernel = select_kernel(kp, &is_reg);
    // the reference-width build (mcgp_config.deviates = MCGP_DEVIATES_53): the register kernel's geometry, its own code
    WideKernelFn wide = nullptr;
    if (kp.wide) {
        wide = select_wide_kernel((uint32_t)kp.n);
        if (!wide || !mcgp::reg_kernel_serves(kp))
            return fail(MCGP_E_BAD_ARG, "deviates = MCGP_DEVIATES_53 serves the problems the register kernel takes "
                                        "(reg_kernel_serves: lap times clear of zero, overtake_delta >= 0)");
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)c.lds_per_block));
        is_reg = true;
        kernel = nullptr;
    }
    DeviceCtx::Slot *sl = nullptr;
    if (!fe.on)           // a block whose matrix is written by the device front end is never shared
        for (auto &cand : c.slot)
            if (cand.used && cand.shareable && std::memcmp(cand.host, &kp, sizeof(kp)) == 0) { sl = &cand; break; }
    if (!sl) {
        sl = &c.slot[c.next_slot];
        c.next_slot = (c.next_slot + 1) % kParamSlots;
        if (sl->used)                                           // every stream that read it has finished
            for (auto &rd : sl->reader)
                if (rd.live) {
                    HIP_TRY(hipEventSynchronize(rd.done));
                    rd.live = false;
                }
        std::memcpy(sl->host, &kp, sizeof(kp));
        sl->used = true;
        sl->shareable = !fe.on;
        HIP_TRY(hipMemcpyAsync(sl->dev, sl->host, sizeof(kp), hipMemcpyHostToDevice, stream));
        if (fe.on) {
            // the n x n matrix goes from the front-end kernel straight into the block's grid_probs slot, on
            // the launch stream, between the upload and the race kernel (the slot is marked not shareable:
            // its pinned host copy, still being read by the asynchronous upload, no longer describes it)
            hipLaunchKernelGGL(grid_probs_kernel, dim3(1), dim3(mcgp::kMaxCars), 0, stream, c.d_fe_in, c.d_fe_pen,
                               (int)kp.n, sl->dev->grid_probs);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(sl->uploaded, stream));
        sl->upload_stream = stream;
    } else if (sl->upload_stream != stream) {
        // cached block uploaded on another stream: order this stream behind that upload
        HIP_TRY(hipStreamWaitEvent(stream, sl->uploaded, 0));
    }
    int ti = -1;
    {
        const int rc = claim_timer(c, stream, &ti);
        if (rc != MCGP_OK) return rc;
    }
    c.timer[ti].seq = ++c.timer_seq;
    if (!c.timer[ti].d_ticket) return fail(MCGP_E_HIP, "no work counter for this stream");
    HIP_TRY(hipEventRecord(c.timer[ti].start, stream));
    const uint64_t cap = max_sims_per_launch();
    uint32_t grid = 0, block = 0, lds = 0;
    // The register kernel's default block fills the CU's LDS to the last half kilobyte at 20 cars (RegGeo).  A device or a
    // runtime that offers less per block gets the same kernel in blocks of kSmallBlockWaves waves -- less than half the
    // footprint, about 15 % slower --: chosen up front when the device reports less LDS than the block needs, and once more,
    // with a fresh launch, when the launch itself is refused for its resources.  The shape in use shows in
    // mcgp_last_launch_info (block_threads) and mcgp_last_kernel_name.
    bool small_shape = false;
    auto resource_error = [](hipError_t e) {
        return e == hipErrorOutOfMemory || e == hipErrorInvalidValue || e == hipErrorLaunchOutOfResources ||
               e == hipErrorInvalidConfiguration;
    };
    for (uint64_t done = 0; done < n_sims; done += cap) {
        const uint64_t m = (n_sims - done) < cap ? (n_sims - done) : cap;
        for (;;) {
            KernelFn k = kernel;
            if (small_shape) {
                k = select_small_kernel((uint32_t)kp.n);
                if (!k) return fail(MCGP_E_HIP, "no small-block instantiation for this field size");
            }
            launch_geometry(c, (uint32_t)kp.n, is_reg, wide ? reinterpret_cast<KernelFn>(wide) : k, m, &grid, &block, &lds,
                            wide ? mcgp::wide_block_waves(kp.n) : small_shape ? mcgp::kSmallBlockWaves : 0, wide != nullptr,
                            kp.total_laps);
            const bool may_shrink = is_reg && !wide && !small_shape;
            if (may_shrink && lds > c.lds_per_block) {              // the device offers less than the default block needs
                small_shape = true;
                continue;
            }
            if (lds > c.lds_per_block)
                return fail(MCGP_E_HIP, "the kernel's smallest block needs " + std::to_string(lds) + " bytes of LDS, the device offers " +
                                            std::to_string(c.lds_per_block) + " per block");
            if (is_reg && !wide) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    if (may_shrink && resource_error(e)) { small_shape = true; continue; }
                    return fail(MCGP_E_HIP, std::string("hipFuncSetAttribute(dynamic LDS): ") + hipGetErrorString(e));
                }
            }
            // units of work: the register kernel's waves claim chunks of 64 simulations from the stream's counter, a block
            // of the generic kernel takes batches of `block` by its index (both < 2^32 because m < 2^32)
            const uint64_t unit = is_reg ? 64u : block;
            const uint64_t n_batches = (m + unit - 1) / unit;
            if (is_reg) {
                HIP_TRY(hipMemsetAsync(c.timer[ti].d_ticket, 0, sizeof(uint32_t), stream));
                // the lanes' retirement lists: scratch of the launch, (n + 1) words per lane, kept per stream and grown on
                // demand (a launch of this stream that still uses the old buffer has been enqueued before the free, which
                // the runtime orders behind it)
                const size_t want = mcgp::reg_retire_ws_bytes(kp.n, (size_t)grid * block);
                if (want > c.timer[ti].retire_bytes) {
                    if (c.timer[ti].d_retire) {
                        HIP_TRY(hipStreamSynchronize(stream));
                        (void)hipFree(c.timer[ti].d_retire);
                    }
                    c.timer[ti].d_retire = nullptr;
                    c.timer[ti].retire_bytes = 0;
                    HIP_TRY(hipMalloc(&c.timer[ti].d_retire, want));
                    c.timer[ti].retire_bytes = want;
                }
            }
            if (wide)
                hipLaunchKernelGGL(wide, dim3(grid), dim3(block), lds, stream, sl->dev, m, sim_offset + done,
                                   (uint32_t)seed, (uint32_t)(seed >> 32), d_hist,
                                   d_orders ? d_orders + (size_t)done * (size_t)kp.n : nullptr, d_fixed_grid,
                                   (uint32_t)n_batches, c.timer[ti].d_ticket, c.timer[ti].d_retire, c.d_norm53);
            else
                hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, stream, sl->dev, m, sim_offset + done,
                                   (uint32_t)seed, (uint32_t)(seed >> 32), d_hist,
                                   d_orders ? d_orders + (size_t)done * (size_t)kp.n : nullptr, d_fixed_grid,
                                   (uint32_t)n_batches, c.timer[ti].d_ticket, c.timer[ti].d_retire);
            const hipError_t e = hipGetLastError();
            if (e == hipSuccess) break;
            if (may_shrink && resource_error(e)) {                  // refused for its resources: once more, small blocks
                small_shape = true;
                continue;
            }
            return fail(e == hipErrorOutOfMemory ? MCGP_E_NOMEM : MCGP_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        }
    }
    HIP_TRY(hipEventRecord(c.timer[ti].stop, stream));
    c.last_timer = ti;
    {
        // completion event of (this block, this stream)
        DeviceCtx::Slot::Reader *rd = nullptr, *spare = nullptr;
        for (auto &r : sl->reader) {
            if (r.live && r.stream == stream) { rd = &r; break; }
            // an entry whose launch has completed is free again (it used to stay "live" until the block was evicted,
            // so that the ninth stream on a long-lived block blocked the host on entry 0 -- the most recent launch)
            if (r.live && hipEventQuery(r.done) == hipSuccess) r.live = false;
            if (!r.live && !spare) spare = &r;
        }
        if (!rd) {
            if (!spare) {                               // more streams IN FLIGHT than entries: wait for the oldest launch
                spare = &sl->reader[0];
                for (auto &r : sl->reader)
                    if (r.seq < spare->seq) spare = &r;
                HIP_TRY(hipEventSynchronize(spare->done));
            }
            rd = spare;
            if (!rd->done) HIP_TRY(hipEventCreateWithFlags(&rd->done, hipEventDisableTiming));
            rd->stream = stream;
            rd->live = true;
        }
        rd->seq = c.timer_seq;
        HIP_TRY(hipEventRecord(rd->done, stream));
    }
    c.last_grid = grid;
    c.last_block = block;
    c.last_lds = lds;
    if (wide) std::snprintf(c.last_kernel, sizeof(c.last_kernel), "mcgp::race_kernel_reg_wide<%d>", kp.n);
    else if (is_reg && small_shape) std::snprintf(c.last_kernel, sizeof(c.last_kernel), "mcgp::race_kernel_reg<%d, %d>", kp.n, mcgp::kSmallBlockWaves);
    else if (is_reg) std::snprintf(c.last_kernel, sizeof(c.last_kernel), "mcgp::race_kernel_reg<%d>", kp.n);
    else std::snprintf(c.last_kernel, sizeof(c.last_kernel), "mcgp::race_kernel");
    return MCGP_OK;
}

}  // namespace

extern "C" {

int32_t mcgp_abi_version(void) { return MCGP_ABI_VERSION; }

// Identity of the sources this binary was compiled from (csrc/source_hash.py, passed in by the Makefile).  The same
// value sits in the file as the text after "MCGP_BUILD_HASH=", so that a loader can read it without mapping the library.
#ifndef MCGP_SOURCE_HASH
#define MCGP_SOURCE_HASH "unknown"
#endif
extern const char mcgp_build_hash_marker[];
__attribute__((used)) const char mcgp_build_hash_marker[] = "MCGP_BUILD_HASH=" MCGP_SOURCE_HASH;
const char *mcgp_build_hash(void) { return mcgp_build_hash_marker + sizeof("MCGP_BUILD_HASH=") - 1; }

int32_t mcgp_device_count(void) { return device_count_nothrow(); }

const char *mcgp_last_error(void) { return g_err.c_str(); }

int32_t mcgp_run_device(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs,
                        uint32_t n, uint64_t n_sims, uint64_t sim_offset, uint64_t seed, int32_t device,
                        void *stream, uint64_t *d_hist, uint8_t *d_orders)
{
    if (!grid_probs || !d_hist) return fail(MCGP_E_BAD_ARG, "grid_probs / d_hist is NULL");
    mcgp::KParams *kp = new (std::nothrow) mcgp::KParams;
    if (!kp) return fail(MCGP_E_NOMEM, "host allocation failed");
    int rc = build_params(cfg, drv, grid_probs, n, kp);
    DeviceCtx *c = nullptr;
    if (rc == MCGP_OK) rc = find_ctx(device, &c);
    if (rc == MCGP_OK) {
        std::lock_guard<std::mutex> lock(c->mu);
        rc = ensure_ctx_locked(device, *c);
        if (rc == MCGP_OK) {
            hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) rc = fail(MCGP_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
            else rc = launch(*c, *kp, n_sims, sim_offset, seed, (hipStream_t)stream,
                             reinterpret_cast<unsigned long long *>(d_hist), d_orders, nullptr);
        }
    }
    delete kp;
    return rc;
}

int32_t mcgp_run(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs, uint32_t n,
                 uint64_t n_sims, uint64_t sim_offset, uint64_t seed, int32_t device, uint64_t *hist_out,
                 uint8_t *orders_out)
{
    if (!grid_probs || !hist_out) return fail(MCGP_E_BAD_ARG, "grid_probs / hist_out is NULL");
    mcgp::KParams *kp = new (std::nothrow) mcgp::KParams;
    if (!kp) return fail(MCGP_E_NOMEM, "host allocation failed");
    int rc = build_params(cfg, drv, grid_probs, n, kp);
    DeviceCtx *c = nullptr;
    if (rc == MCGP_OK) rc = find_ctx(device, &c);
    if (rc != MCGP_OK) { delete kp; return rc; }
    std::lock_guard<std::mutex> lock(c->mu);
    auto body = [&]() -> int {
        int r = ensure_ctx_locked(device, *c);
        if (r != MCGP_OK) return r;
        HIP_TRY(hipSetDevice(device));
        const size_t hist_bytes = sizeof(unsigned long long) * n * n;
        HIP_TRY(hipMemsetAsync(c->d_hist, 0, hist_bytes, nullptr));
        // per-simulation orders are staged in chunks so the device buffer stays bounded; the staging
        // buffer is kept in the context (grown on demand), not allocated per call
        const uint64_t chunk = orders_out ? (uint64_t)(1u << 22) : n_sims;
        if (orders_out && n_sims) {
            const size_t want = (size_t)(chunk < n_sims ? chunk : n_sims) * n;
            if (want > c->d_orders_bytes) {
                if (c->d_orders) (void)hipFree(c->d_orders);
                c->d_orders = nullptr;
                c->d_orders_bytes = 0;
                HIP_TRY(hipMalloc(&c->d_orders, want));
                c->d_orders_bytes = want;
            }
        }
        uint8_t *d_orders = orders_out ? c->d_orders : nullptr;
        for (uint64_t done = 0; done < n_sims && r == MCGP_OK; done += chunk) {
            const uint64_t m = (n_sims - done) < chunk ? (n_sims - done) : chunk;
            r = launch(*c, *kp, m, sim_offset + done, seed, nullptr, c->d_hist, d_orders, nullptr);
            if (r == MCGP_OK && d_orders) {
                hipError_t e = hipMemcpy(orders_out + (size_t)done * n, d_orders, (size_t)m * n, hipMemcpyDeviceToHost);
                if (e != hipSuccess) r = fail(MCGP_E_HIP, std::string("hipMemcpy(orders): ") + hipGetErrorString(e));
            }
        }
        if (r != MCGP_OK) return r;
        unsigned long long h[MCGP_MAX_CARS * MCGP_MAX_CARS];
        HIP_TRY(hipMemcpy(h, c->d_hist, hist_bytes, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n * n; ++i) hist_out[i] += h[i];
        return MCGP_OK;
    };
    rc = body();
    delete kp;
    return rc;
}

int32_t mcgp_simulate_race(const mcgp_config *cfg, const mcgp_drivers *drv, const uint8_t *grid, uint32_t n,
                           uint64_t sim_id, uint64_t seed, int32_t device, uint8_t *order_out)
{
    if (!grid || !order_out) return fail(MCGP_E_BAD_ARG, "grid / order_out is NULL");
    if (n >= 1 && n <= MCGP_MAX_CARS) {
        uint32_t seen = 0;
        for (uint32_t p = 0; p < n; ++p) {
            if (grid[p] >= n || (seen >> grid[p] & 1u)) return fail(MCGP_E_BAD_ARG, "grid is not a permutation of 0..n-1");
            seen |= 1u << grid[p];
        }
    }
    mcgp::KParams *kp = new (std::nothrow) mcgp::KParams;
    if (!kp) return fail(MCGP_E_NOMEM, "host allocation failed");
    int rc = build_params(cfg, drv, nullptr, n, kp);
    DeviceCtx *c = nullptr;
    if (rc == MCGP_OK) rc = find_ctx(device, &c);
    if (rc != MCGP_OK) { delete kp; return rc; }
    std::lock_guard<std::mutex> lock(c->mu);
    auto body = [&]() -> int {
        int r = ensure_ctx_locked(device, *c);
        if (r != MCGP_OK) return r;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpy(c->d_grid, grid, n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemsetAsync(c->d_hist, 0, sizeof(unsigned long long) * n * n, nullptr));
        r = launch(*c, *kp, 1, sim_id, seed, nullptr, c->d_hist, c->d_order1, c->d_grid);
        if (r == MCGP_OK) {
            hipError_t e = hipMemcpy(order_out, c->d_order1, n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) r = fail(MCGP_E_HIP, std::string("hipMemcpy(order): ") + hipGetErrorString(e));
        }
        return r;
    };
    rc = body();
    delete kp;
    return rc;
}

// Uploads the front-end inputs of one call into the context's buffers (NULL stream: ordered before what follows).
static int upload_front_end(DeviceCtx &c, const double *rating, const double *teammate_delta, const double *form_score,
                            const double *circuit_affinity, const int32_t *penalty, uint32_t n)
{
    if (!rating || !teammate_delta || !form_score || !circuit_affinity || !penalty)
        return fail(MCGP_E_BAD_ARG, "a front-end input array is NULL");
    if (n < 1 || n > MCGP_MAX_CARS) return fail(MCGP_E_BAD_ARG, "n must be in [1, 32]");
    double in[4 * MCGP_MAX_CARS];
    for (uint32_t d = 0; d < n; ++d) {
        // a non-finite input makes the softmax (inf - inf) or the row adjustments NaN: the race kernel would then
        // sample uniform grids silently, where mcgp_run rejects such a matrix
        if (!std::isfinite(rating[d]) || !std::isfinite(teammate_delta[d]) || !std::isfinite(form_score[d]) ||
            !std::isfinite(circuit_affinity[d]))
            return fail(MCGP_E_BAD_ARG, "a front-end input (rating / teammate_delta / form_score / circuit_affinity) "
                                        "is not finite");
        in[d] = rating[d];
        in[n + d] = teammate_delta[d];
        in[2 * n + d] = form_score[d];
        in[3 * n + d] = circuit_affinity[d];
    }
    HIP_TRY(hipMemcpy(c.d_fe_in, in, sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c.d_fe_pen, penalty, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    return MCGP_OK;
}

int32_t mcgp_grid_probs(const double *quali_rating, const double *teammate_delta, const double *form_score,
                        const double *circuit_affinity, const int32_t *penalty, uint32_t n, int32_t device,
                        double *grid_probs_out)
{
    if (!grid_probs_out) return fail(MCGP_E_BAD_ARG, "grid_probs_out is NULL");
    DeviceCtx *c = nullptr;
    int rc = find_ctx(device, &c);
    if (rc != MCGP_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    auto body = [&]() -> int {
        int r = ensure_ctx_locked(device, *c);
        if (r != MCGP_OK) return r;
        HIP_TRY(hipSetDevice(device));
        r = upload_front_end(*c, quali_rating, teammate_delta, form_score, circuit_affinity, penalty, n);
        if (r != MCGP_OK) return r;
        hipLaunchKernelGGL(grid_probs_kernel, dim3(1), dim3(mcgp::kMaxCars), 0, nullptr, c->d_fe_in, c->d_fe_pen, (int)n,
                           c->d_fe_out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(grid_probs_out, c->d_fe_out, sizeof(double) * n * n, hipMemcpyDeviceToHost));
        return MCGP_OK;
    };
    return body();
}

int32_t mcgp_elo_season(uint32_t n_drivers, uint32_t n_events, const int32_t *kind, const double *k,
                        const uint32_t *count, const uint8_t *who, const double *value, double *ratings,
                        double *after_out, int32_t device)
{
    const uint32_t n = n_drivers;
    if (n < 1 || n > MCGP_MAX_CARS) return fail(MCGP_E_BAD_ARG, "n_drivers must be in [1, 32]");
    if (!ratings) return fail(MCGP_E_BAD_ARG, "ratings is NULL");
    if (n_events > 0 && (!kind || !k || !count || !who || !value)) return fail(MCGP_E_BAD_ARG, "an event array is NULL");
    for (uint32_t i = 0; i < 2 * n; ++i)
        if (!std::isfinite(ratings[i])) return fail(MCGP_E_BAD_ARG, "non-finite rating");
    for (uint32_t e = 0; e < n_events; ++e) {
        if (kind[e] != 0 && kind[e] != 1) return fail(MCGP_E_BAD_ARG, "event kind must be 0 (qualifying) or 1 (race)");
        if (!std::isfinite(k[e])) return fail(MCGP_E_BAD_ARG, "non-finite K factor");
        if (count[e] > n) return fail(MCGP_E_BAD_ARG, "more entries than drivers in an event");
        uint32_t seen = 0;
        for (uint32_t j = 0; j < count[e]; ++j) {
            const uint32_t d = who[(size_t)e * n + j];
            if (d >= n) return fail(MCGP_E_BAD_ARG, "driver index out of range in an event");
            if (seen & (1u << d)) return fail(MCGP_E_BAD_ARG, "a driver is listed twice in one event");
            seen |= 1u << d;
            if (!std::isfinite(value[(size_t)e * n + j])) return fail(MCGP_E_BAD_ARG, "non-finite lap time / position");
        }
    }
    DeviceCtx *c = nullptr;
    int rc = find_ctx(device, &c);
    if (rc != MCGP_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    // one buffer for everything the kernel reads and writes, packed on the host: one upload, one launch, one
    // download (the call is latency-bound: a season is a few hundred events for one wavefront)
    const size_t ne = n_events, row = n;
    const size_t o_rat = 0, o_after = o_rat + 2 * row * 8, o_k = o_after + (after_out ? ne * 2 * row * 8 : 0);
    const size_t o_val = o_k + ne * 8, o_kind = o_val + ne * row * 8, o_cnt = o_kind + ne * 4, o_who = o_cnt + ne * 4;
    const size_t bytes = o_who + ne * row;
    std::vector<unsigned char> host(bytes - o_k + 2 * row * 8);        // [ratings | inputs]: the after block is output only
    const size_t in0 = 2 * row * 8;                                    // inputs start here in `host`, at o_k on the device
    std::memcpy(host.data(), ratings, 2 * row * 8);
    if (ne) {
        std::memcpy(host.data() + in0 + (o_k - o_k), k, ne * 8);
        std::memcpy(host.data() + in0 + (o_val - o_k), value, ne * row * 8);
        std::memcpy(host.data() + in0 + (o_kind - o_k), kind, ne * 4);
        std::memcpy(host.data() + in0 + (o_cnt - o_k), count, ne * 4);
        std::memcpy(host.data() + in0 + (o_who - o_k), who, ne * row);
    }
    auto body = [&]() -> int {
        int r = ensure_ctx_locked(device, *c);
        if (r != MCGP_OK) return r;
        HIP_TRY(hipSetDevice(device));
        if (c->elo_bytes < bytes) {                                    // grow-only scratch of the context
            if (c->d_elo) (void)hipFree(c->d_elo);
            c->d_elo = nullptr;
            c->elo_bytes = 0;
            HIP_TRY(hipMalloc(&c->d_elo, bytes));
            c->elo_bytes = bytes;
        }
        unsigned char *d = c->d_elo;
        HIP_TRY(hipMemcpy(d + o_rat, host.data(), 2 * row * 8, hipMemcpyHostToDevice));
        if (ne) HIP_TRY(hipMemcpy(d + o_k, host.data() + in0, bytes - o_k, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(elo_season_kernel, dim3(1), dim3(mcgp::kMaxCars * mcgp::kMaxCars), 0, nullptr, (int)n, (int)n_events,
                           reinterpret_cast<const int32_t *>(d + o_kind), reinterpret_cast<const double *>(d + o_k),
                           reinterpret_cast<const uint32_t *>(d + o_cnt), d + o_who,
                           reinterpret_cast<const double *>(d + o_val), reinterpret_cast<double *>(d + o_rat),
                           after_out ? reinterpret_cast<double *>(d + o_after) : nullptr);
        HIP_TRY(hipGetLastError());
        if (after_out && ne) {
            // ratings and the snapshots are adjacent: one download
            std::vector<double> back(2 * row + ne * 2 * row);
            HIP_TRY(hipMemcpy(back.data(), d + o_rat, back.size() * 8, hipMemcpyDeviceToHost));
            std::memcpy(ratings, back.data(), 2 * row * 8);
            std::memcpy(after_out, back.data() + 2 * row, ne * 2 * row * 8);
        } else {
            HIP_TRY(hipMemcpy(ratings, d + o_rat, 2 * row * 8, hipMemcpyDeviceToHost));
        }
        return MCGP_OK;
    };
    return body();
}

int32_t mcgp_run_from_ratings(const mcgp_config *cfg, const mcgp_drivers *drv, const double *quali_rating,
                              const double *teammate_delta, const double *form_score,
                              const double *circuit_affinity, const int32_t *penalty, uint32_t n, uint64_t n_sims,
                              uint64_t sim_offset, uint64_t seed, int32_t device, uint64_t *hist_out,
                              double *grid_probs_out)
{
    if (!hist_out) return fail(MCGP_E_BAD_ARG, "hist_out is NULL");
    mcgp::KParams *kp = new (std::nothrow) mcgp::KParams;
    if (!kp) return fail(MCGP_E_NOMEM, "host allocation failed");
    int rc = build_params(cfg, drv, nullptr, n, kp);               // grid_probs slot left zero: the device fills it
    DeviceCtx *c = nullptr;
    if (rc == MCGP_OK) rc = find_ctx(device, &c);
    if (rc != MCGP_OK) { delete kp; return rc; }
    std::lock_guard<std::mutex> lock(c->mu);
    auto body = [&]() -> int {
        int r = ensure_ctx_locked(device, *c);
        if (r != MCGP_OK) return r;
        HIP_TRY(hipSetDevice(device));
        r = upload_front_end(*c, quali_rating, teammate_delta, form_score, circuit_affinity, penalty, n);
        if (r != MCGP_OK) return r;
        const size_t hist_bytes = sizeof(unsigned long long) * n * n;
        HIP_TRY(hipMemsetAsync(c->d_hist, 0, hist_bytes, nullptr));
        FrontEnd fe;
        fe.on = true;
        r = launch(*c, *kp, n_sims, sim_offset, seed, nullptr, c->d_hist, nullptr, nullptr, fe);
        if (r != MCGP_OK) return r;
        unsigned long long h[MCGP_MAX_CARS * MCGP_MAX_CARS];
        HIP_TRY(hipMemcpy(h, c->d_hist, hist_bytes, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n * n; ++i) hist_out[i] += h[i];
        if (grid_probs_out) {
            // the matrix the race kernel sampled from, recomputed by the same kernel from the same inputs into the
            // context's scratch matrix (the parameter block itself is not read back)
            hipLaunchKernelGGL(grid_probs_kernel, dim3(1), dim3(mcgp::kMaxCars), 0, nullptr, c->d_fe_in, c->d_fe_pen, (int)n,
                               c->d_fe_out);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(grid_probs_out, c->d_fe_out, sizeof(double) * n * n, hipMemcpyDeviceToHost));
        }
        return MCGP_OK;
    };
    rc = body();
    delete kp;
    return rc;
}

int32_t mcgp_run_batch(uint32_t n_problems, const mcgp_config *cfgs, const mcgp_drivers *drvs,
                       const double *const *grid_probs, uint32_t n, uint64_t n_sims, const uint64_t *sim_offsets,
                       const uint64_t *seeds, int32_t device, uint64_t *hist_out)
{
    if (!cfgs || !drvs || !grid_probs || !seeds || !hist_out) return fail(MCGP_E_BAD_ARG, "a batch array is NULL");
    if (n_problems < 1 || n_problems > 4096) return fail(MCGP_E_BAD_ARG, "n_problems must be in [1, 4096]");
    if (n_sims >= 0xFFFFFE00ull) return fail(MCGP_E_BAD_ARG, "n_sims per problem must be below 2^32 - 512 in a batch");
    if (n_sims == 0) return MCGP_OK;
    // Every problem gets what mcgp_run would give it (reference src/validation.py:179-185: a sweep is a loop over
    // independent predictions, none of which can make another one fail).  The problems the batch kernel takes -- the
    // register kernel's domain at the default deviate width -- share ONE launch; the others (deviates = 53, a problem only
    // the generic kernel serves, everything under MCGP_FORCE_GENERIC=1) run one after the other through the single-problem
    // path, inside this call, into the same output.
    std::vector<mcgp::KParams> kps;                  // the problems of the shared launch, compacted
    std::vector<mcgp::BatchItem> items;
    std::vector<uint32_t> shared_index, solo_index;  // original indices
    std::vector<mcgp::KParams> solo_kps;
    const char *force = std::getenv("MCGP_FORCE_GENERIC");
    const bool force_generic = force && force[0] == '1';
    const BatchKernelFn kernel = select_batch_kernel(n);
    {
        mcgp::KParams kp;
        for (uint32_t p = 0; p < n_problems; ++p) {
            if (!grid_probs[p]) return fail(MCGP_E_BAD_ARG, "a grid_probs pointer of the batch is NULL");
            const int rc = build_params(&cfgs[p], &drvs[p], grid_probs[p], n, &kp);
            if (rc != MCGP_OK) return rc;
            if (kp.wide && !mcgp::reg_kernel_serves(kp))
                return fail(MCGP_E_BAD_ARG, "deviates = MCGP_DEVIATES_53 serves the problems the register kernel takes "
                                            "(reg_kernel_serves: lap times clear of zero, overtake_delta >= 0)");
            if (!kernel || force_generic || kp.wide || !mcgp::reg_kernel_serves(kp)) {
                solo_index.push_back(p);
                solo_kps.push_back(kp);
            } else {
                shared_index.push_back(p);
                kps.push_back(kp);
                items.push_back(mcgp::BatchItem{sim_offsets ? sim_offsets[p] : 0ull, seeds[p]});
            }
        }
    }
    const uint32_t n_shared = (uint32_t)kps.size();
    DeviceCtx *c = nullptr;
    int rc = find_ctx(device, &c);
    if (rc != MCGP_OK) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    auto body = [&]() -> int {
        int r = ensure_ctx_locked(device, *c);
        if (r != MCGP_OK) return r;
        HIP_TRY(hipSetDevice(device));
        // the call's own pair of timing events (mcgp_last_kernel_ms after a batch call = everything it ran on the device)
        if (!c->batch_start) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            hipError_t err = hipEventCreate(&e0);
            if (err == hipSuccess) err = hipEventCreate(&e1);
            if (err != hipSuccess) {
                if (e0) (void)hipEventDestroy(e0);
                return fail(MCGP_E_HIP, std::string("batch timing events: ") + hipGetErrorString(err));
            }
            c->batch_start = e0;
            c->batch_stop = e1;
        }
        HIP_TRY(hipEventRecord(c->batch_start, nullptr));
        const size_t cell_bytes = sizeof(unsigned long long) * n * n;
        if (n_shared) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)c->lds_per_block));
            // geometry: the register kernel's block, one more LDS word that names the block's next problem
            const int waves = mcgp::reg_block_waves((int)n);
            const uint32_t block = (uint32_t)waves * 64u;
            const size_t lds = mcgp::shared_lds_bytes_reg((int)n) + (size_t)block * mcgp::per_thread_lds_bytes_reg((int)n) +
                               mcgp::kBatchLdsExtra;
            int reg_cap = 4 * mcgp::reg_min_waves((int)n);
            int blocks_per_cu = (int)(c->lds_per_block / lds);
            if (blocks_per_cu * waves > reg_cap) blocks_per_cu = reg_cap / waves;
            if (blocks_per_cu < 1) blocks_per_cu = 1;
            const uint32_t n_chunks = (uint32_t)((n_sims + 63) / 64);
            // no more blocks than the batch has wave-chunks to fill them with
            const uint64_t n_tasks = ((uint64_t)n_shared * n_chunks + (uint64_t)waves - 1) / (uint64_t)waves;
            uint64_t grid = (uint64_t)c->cu_count * (uint64_t)blocks_per_cu;
            if (grid > n_tasks) grid = n_tasks;
            // one device buffer: [parameter blocks | items | histograms | one ticket counter per problem]
            const size_t o_items = sizeof(mcgp::KParams) * n_shared;
            const size_t o_hist = o_items + sizeof(mcgp::BatchItem) * n_shared;
            const size_t hist_bytes = cell_bytes * n_shared;
            const size_t o_ticket = o_hist + hist_bytes;
            const size_t ticket_bytes = (sizeof(uint32_t) * n_shared + 15) / 16 * 16;
            const size_t bytes = o_ticket + ticket_bytes;
            if (bytes > c->batch_bytes) {
                if (c->d_batch) (void)hipFree(c->d_batch);
                c->d_batch = nullptr;
                c->batch_bytes = 0;
                HIP_TRY(hipMalloc(&c->d_batch, bytes));
                c->batch_bytes = bytes;
            }
            const size_t ws = mcgp::reg_retire_ws_bytes((int)n, (size_t)grid * block);
            if (ws > c->batch_retire_bytes) {
                if (c->d_batch_retire) (void)hipFree(c->d_batch_retire);
                c->d_batch_retire = nullptr;
                c->batch_retire_bytes = 0;
                HIP_TRY(hipMalloc(&c->d_batch_retire, ws));
                c->batch_retire_bytes = ws;
            }
            unsigned char *d = c->d_batch;
            HIP_TRY(hipMemcpyAsync(d, kps.data(), o_items, hipMemcpyHostToDevice, nullptr));
            HIP_TRY(hipMemcpyAsync(d + o_items, items.data(), sizeof(mcgp::BatchItem) * n_shared, hipMemcpyHostToDevice, nullptr));
            HIP_TRY(hipMemsetAsync(d + o_hist, 0, hist_bytes + ticket_bytes, nullptr));
            hipLaunchKernelGGL(kernel, dim3((uint32_t)grid), dim3(block), lds, nullptr,
                               reinterpret_cast<const mcgp::KParams *>(d), reinterpret_cast<const mcgp::BatchItem *>(d + o_items),
                               n_shared, n_sims, reinterpret_cast<unsigned long long *>(d + o_hist), n_chunks,
                               reinterpret_cast<uint32_t *>(d + o_ticket), c->d_batch_retire);
            HIP_TRY(hipGetLastError());
            std::vector<unsigned long long> h((size_t)n * n * n_shared);
            HIP_TRY(hipMemcpy(h.data(), d + o_hist, hist_bytes, hipMemcpyDeviceToHost));
            for (uint32_t j = 0; j < n_shared; ++j) {
                uint64_t *out = hist_out + (size_t)shared_index[j] * n * n;
                for (uint32_t i = 0; i < n * n; ++i) out[i] += h[(size_t)j * n * n + i];
            }
            c->last_grid = (uint32_t)grid;
            c->last_block = block;
            c->last_lds = (uint32_t)lds;
            std::snprintf(c->last_kernel, sizeof(c->last_kernel), "mcgp::race_kernel_reg_batch<%u>", n);
        }
        // the problems that run by themselves: exactly mcgp_run's path, one after the other (null stream)
        for (size_t j = 0; j < solo_index.size(); ++j) {
            const uint32_t p = solo_index[j];
            HIP_TRY(hipMemsetAsync(c->d_hist, 0, cell_bytes, nullptr));
            r = launch(*c, solo_kps[j], n_sims, sim_offsets ? sim_offsets[p] : 0ull, seeds[p], nullptr, c->d_hist, nullptr, nullptr);
            if (r != MCGP_OK) return r;
            unsigned long long h[MCGP_MAX_CARS * MCGP_MAX_CARS];
            HIP_TRY(hipMemcpy(h, c->d_hist, cell_bytes, hipMemcpyDeviceToHost));
            uint64_t *out = hist_out + (size_t)p * n * n;
            for (uint32_t i = 0; i < n * n; ++i) out[i] += h[i];
        }
        HIP_TRY(hipEventRecord(c->batch_stop, nullptr));
        c->last_timer = kBatchTimer;
        return MCGP_OK;
    };
    return body();
}

int32_t mcgp_last_kernel_ms(int32_t device, float *ms_out)
{
    if (!ms_out) return fail(MCGP_E_BAD_ARG, "ms_out is NULL");
    if (device < 0 || device >= kMaxDevices) return fail(MCGP_E_BAD_ARG, "device index out of range");
    DeviceCtx &c = g_ctx[device];
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.ready || (c.last_timer < 0 && c.last_timer != kBatchTimer))
        return fail(MCGP_E_BAD_ARG, "no kernel launched on this device yet");
    HIP_TRY(hipSetDevice(device));
    if (c.last_timer == kBatchTimer) {                  // mcgp_run_batch: everything the call ran on the device
        HIP_TRY(hipEventSynchronize(c.batch_stop));
        HIP_TRY(hipEventElapsedTime(ms_out, c.batch_start, c.batch_stop));
        return MCGP_OK;
    }
    HIP_TRY(hipEventSynchronize(c.timer[c.last_timer].stop));
    HIP_TRY(hipEventElapsedTime(ms_out, c.timer[c.last_timer].start, c.timer[c.last_timer].stop));
    return MCGP_OK;
}

int32_t mcgp_stream_kernel_ms(int32_t device, void *stream, float *ms_out)
{
    if (!ms_out) return fail(MCGP_E_BAD_ARG, "ms_out is NULL");
    if (device < 0 || device >= kMaxDevices) return fail(MCGP_E_BAD_ARG, "device index out of range");
    DeviceCtx &c = g_ctx[device];
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.ready) return fail(MCGP_E_BAD_ARG, "no kernel launched on this device yet");
    for (auto &t : c.timer)
        if (t.used && t.stream == (hipStream_t)stream) {
            HIP_TRY(hipSetDevice(device));
            HIP_TRY(hipEventSynchronize(t.stop));
            HIP_TRY(hipEventElapsedTime(ms_out, t.start, t.stop));
            return MCGP_OK;
        }
    return fail(MCGP_E_BAD_ARG, "no timed launch on this stream (or its entry was recycled by 8 newer streams)");
}

const char *mcgp_last_kernel_name(int32_t device)
{
    if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready) return "";
    return g_ctx[device].last_kernel;
}

int32_t mcgp_last_launch_info(int32_t device, uint32_t *grid_blocks, uint32_t *block_threads, uint32_t *lds_bytes)
{
    if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready ||
        (g_ctx[device].last_timer < 0 && g_ctx[device].last_timer != kBatchTimer))
        return fail(MCGP_E_BAD_ARG, "no kernel launched on this device yet");
    if (grid_blocks) *grid_blocks = g_ctx[device].last_grid;
    if (block_threads) *block_threads = g_ctx[device].last_block;
    if (lds_bytes) *lds_bytes = g_ctx[device].last_lds;
    return MCGP_OK;
}

}  // extern "C"
