/* mcgp.h -- C ABI of the MI355X (gfx950) Monte Carlo race-simulation engine.
 *
 * libmcgp_hip.so is the drop-in for the reference's hot path
 *     RaceSimulator(config).run_monte_carlo(...)      reference src/simulation.py:56-100
 * (sole call site: reference src/predictor.py:264,283-291).  The reference has no
 * FFI of its own (it is pure Python); these entry points are what a ctypes
 * binding of that method binds -- see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain pointers and sizes, no C++ / torch types; caller owns every buffer and
 *     the library keeps no pointer past return;
 *   - every function returns 0 on success or a negative MCGP_E_* code and never
 *     throws or aborts; mcgp_last_error() gives the thread-local message;
 *   - blocking calls, re-entrant for distinct devices (one cached context per
 *     device, guarded by a mutex);
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with MCGP_E_NO_DEVICE;
 *   - no RNG state: every random draw is a pure function of
 *     (seed, sim_offset + i, lap, purpose, index) (Philox4x32-10), so any split
 *     of [0, N) over calls / devices / ranks sums to the same histogram.
 */
#ifndef MCGP_H
#define MCGP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: mcgp_build_hash, mcgp_run_batch; the retirement draw of laps >= 2 moved to one word per driver and race, which
 * changes the results for a given seed (the oracle's Philox back-end moved with it) */
#define MCGP_ABI_VERSION 2
#define MCGP_MAX_CARS 32
#define MCGP_MAX_LAPS 1000

enum {
    MCGP_OK = 0,
    MCGP_E_BAD_ARG = -1,    /* NULL pointer, n out of [1, 32], laps out of [1, 1000], bad enum */
    MCGP_E_NO_DEVICE = -2,  /* no HIP device / device index out of range */
    MCGP_E_HIP = -3,        /* a HIP runtime call failed (message has the HIP error string) */
    MCGP_E_NOMEM = -4       /* device or host allocation failed */
};

/* tyre compounds (reference src/config.py:45-51) and track condition
 * (run_monte_carlo argument `track_condition`, reference src/simulation.py:68) */
enum { MCGP_SOFT = 0, MCGP_MEDIUM = 1, MCGP_HARD = 2, MCGP_INTERMEDIATE = 3, MCGP_WET = 4 };
enum { MCGP_DRY = 0, MCGP_DAMP = 1, MCGP_WET_TRACK = 2 };
enum { MCGP_DEVIATES_32 = 0, MCGP_DEVIATES_53 = 1 };

/* RaceConfig -- replaces the dataclass at reference src/simulation.py:37-52, with
 * the dict-valued fields resolved to dense tables by the caller (the host shim):
 *   tire_compounds[c].get('pace_delta', 0) / .get('deg_rate', 0.05) / .get('optimal_laps', 30)
 *   (reference :317-325, :454-455).
 * pop_*: outcome of `available.pop()` on the two-element set at reference :486,488,
 * which depends on PYTHONHASHSEED in the reference; explicit here. */
typedef struct mcgp_config {
    int32_t total_laps;
    int32_t track_condition;
    double pit_loss;
    double overtake_delta;
    double sc_probability;
    double vsc_probability;
    double red_flag_probability;
    double drs_delta;
    double dirty_air_threshold;
    double dirty_air_penalty;
    double comp_pace_delta[5];
    double comp_deg_rate[5];
    int32_t comp_optimal_laps[5];
    int32_t pop_soft_hard;      /* compound id taken from {SOFT, HARD}   */
    int32_t pop_medium_hard;    /* compound id taken from {MEDIUM, HARD} */
    /* Width of the random deviates.  0 (default): 32-bit uniforms w / 2^32 and normals from a binary32 cubic table --
     * the product's fast path.  MCGP_DEVIATES_53: the reference's width -- 53-bit uniforms (random.random(),
     * np.random.choice: genrand_res53) and binary64 normals (reference src/simulation.py:137,194,302,330,524) --, every
     * draw keeping the 32-bit mode's word as its leading bits and taking 21 more from a companion Philox block; a
     * priced option (half as many Philox blocks again, a binary64 inverse normal from a table in LDS), built for every
     * field size; MCGP_E_BAD_ARG for a problem only the generic kernel takes. */
    int32_t deviates;
} mcgp_config;

/* Per-driver inputs as structure-of-arrays of length n; index = position of the driver
 * in grid_probs' key order.  Replaces the five dict arguments of run_monte_carlo
 * (reference :62-66) with their .get() defaults resolved:
 *   base_pace     base_pace.get(d, 90.0)                           :202,294,514
 *   tire_deg      tire_deg.get(d, 0.05)                            :203,295,514
 *   tire_deg_pit  tire_deg.get(d, 0.0)                             :458
 *   variance      driver_variance.get(d, 0.15)                     :204,296
 *   team_dnf      config.dnf_rates.get(driver_teams.get(d,'Unknown'), 0.002)   :286
 *   lap_dnf       driver_dnf_rates.get(d, team_dnf[d])             :190-193 */
typedef struct mcgp_drivers {
    const double *base_pace;
    const double *tire_deg;
    const double *tire_deg_pit;
    const double *variance;
    const double *team_dnf;
    const double *lap_dnf;
} mcgp_drivers;

int32_t mcgp_abi_version(void);
/* Identity of the binary: the hash of the sources it was compiled from (csrc/source_hash.py: sha256 over the .hip and .h
 * files of csrc/, the Makefile and this header, 16 hex digits; "unknown" for a build outside the Makefile).  The host
 * binding computes the same hash from the tree and refuses a library that carries another one; bench.py quotes
 * profiled counters only when they were taken from a binary with this hash.  The file also holds the text
 * "MCGP_BUILD_HASH=<hash>", readable without loading the library. */
const char *mcgp_build_hash(void);
int32_t mcgp_device_count(void);          /* number of HIP devices, 0 if none */
const char *mcgp_last_error(void);        /* thread-local, never NULL */

/* run_monte_carlo (reference :59-100) on `device`.
 *   grid_probs  n x n row-major [driver][grid slot]     (grid_probs dict, reference :62)
 *   hist_out    n x n row-major [driver][position-1] counts; ACCUMULATED into
 *               (caller zeroes); divide by the total simulation count for the
 *               probabilities of reference :97-100
 *   orders_out  optional, [n_sims][n]: driver index classified p-th in simulation
 *               sim_offset+i; NULL to skip (histogram-only mode writes no per-sim bytes)
 * Host buffers in, host buffers out. */
int32_t mcgp_run(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs,
                 uint32_t n, uint64_t n_sims, uint64_t sim_offset, uint64_t seed,
                 int32_t device, uint64_t *hist_out, uint8_t *orders_out);

/* Same computation with DEVICE-resident outputs, asynchronous on `stream`
 * (a hipStream_t, NULL = the device's default stream): d_hist (n*n uint64, device,
 * accumulated) and optional d_orders (n_sims*n uint8, device).  Used by the
 * multi-GPU path (RCCL all-reduce of d_hist) and by bench.py.  The small
 * parameter block is uploaded on the same stream before the launch; a block cached
 * from an earlier call on ANOTHER stream is re-used only behind an event wait on
 * that upload, and is overwritten (4 blocks are cached) only after the launches of
 * EVERY stream that used it have completed.  Runs of 2^32 simulations or more are split into several launches
 * on the stream (the per-block histogram counts in 32 bits).  Every launch is preceded, on the same stream, by a
 * 4-byte memset of the stream's work counter (the kernel's waves claim their simulations from it; the library
 * keeps a counter per stream, 8 per device, and recycles the least recently used one behind its last launch).
 * The call itself is NOT capturable into a hipGraph: on a stream's first use it creates events and allocates the
 * counter and the kernel's per-lane scratch, a recycled counter or an evicted parameter block is waited for on the
 * host, and a captured graph would keep a counter the library may later hand to another stream.  A step is one
 * memset + one kernel of ~75 ms at bench size, so there is no launch overhead for a graph to remove.  Results do
 * not depend on which wave ran which simulation. */
int32_t mcgp_run_device(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs,
                        uint32_t n, uint64_t n_sims, uint64_t sim_offset, uint64_t seed,
                        int32_t device, void *stream, uint64_t *d_hist, uint8_t *d_orders);

/* Several problems in ONE launch: the reference predicts a race from 10 000 simulations (reference
 * src/predictor.py:284) and a backtest runs the races of a season one after the other (src/validation.py:179-185); at
 * that size a launch lasts as long as one race of one lane and most of the device idles.  n_problems races of the same
 * field size n, n_sims simulations each: cfgs[p], drvs[p], grid_probs[p] (n x n) as for mcgp_run, simulation ids
 * sim_offsets[p] .. sim_offsets[p] + n_sims - 1 (NULL: 0) under seeds[p]; hist_out = [n_problems][n][n], ACCUMULATED
 * into.  Every problem gets exactly what mcgp_run would give it, error or histogram (a sweep of the reference is a loop
 * over independent predictions): the problems the shared launch takes go into it; a problem at deviates =
 * MCGP_DEVIATES_53, or one only the second kernel serves (lap times near zero, a negative overtake_delta), or every
 * problem under MCGP_FORCE_GENERIC=1, runs by itself inside the same call.  Host buffers in and out, blocking.
 * mcgp_last_kernel_ms afterwards = the device time of everything the call ran. */
int32_t mcgp_run_batch(uint32_t n_problems, const mcgp_config *cfgs, const mcgp_drivers *drvs,
                       const double *const *grid_probs, uint32_t n, uint64_t n_sims, const uint64_t *sim_offsets,
                       const uint64_t *seeds, int32_t device, uint64_t *hist_out);

/* simulate_race (reference :147-242): one race from a FIXED starting grid
 * (grid[p] = driver index on slot p), simulation id sim_id.  order_out[p] = driver
 * index classified p-th.  Bit-identical to what mcgp_run computes for a simulation
 * whose sampled grid equals `grid`. */
int32_t mcgp_simulate_race(const mcgp_config *cfg, const mcgp_drivers *drv, const uint8_t *grid,
                           uint32_t n, uint64_t sim_id, uint64_t seed, int32_t device,
                           uint8_t *order_out);

/* Grid-probability front end on the device ("next" row f3): the n x n matrix [driver][grid slot] that
 * F1Predictor._predict_quali + _adjust_for_penalties build from the Elo quali ratings
 * (reference src/elo.py:124-141 softmax; src/predictor.py:321-375 teammate / form / circuit adjustments and the
 * Gaussian bump around (1 - p) n; :377-407 penalty shift).  Inputs are arrays of length n in driver order with the
 * reference's .get() defaults resolved by the caller (rating: ratings.get(d).get('quali', 1500); features: 0;
 * penalty: grid positions, strings already mapped through PENALTY_TYPES).  exp() is the front end's own
 * (csrc/frontend_exp.h): results agree with the reference's numpy matrices to ~1e-15 relative and are
 * bit-identical to the CPU oracle's restatement of the same text.
 *   mcgp_grid_probs        host arrays in, host matrix out (n x n doubles)
 *   mcgp_run_from_ratings  run_monte_carlo with the matrix produced ON THE DEVICE, written by the front-end
 *                          kernel straight into the race kernel's parameter block (no host round trip);
 *                          hist_out as in mcgp_run, grid_probs_out optional (NULL to skip). */
int32_t mcgp_grid_probs(const double *quali_rating, const double *teammate_delta, const double *form_score,
                        const double *circuit_affinity, const int32_t *penalty, uint32_t n, int32_t device,
                        double *grid_probs_out);
int32_t mcgp_run_from_ratings(const mcgp_config *cfg, const mcgp_drivers *drv, const double *quali_rating,
                              const double *teammate_delta, const double *form_score,
                              const double *circuit_affinity, const int32_t *penalty, uint32_t n, uint64_t n_sims,
                              uint64_t sim_offset, uint64_t seed, int32_t device, uint64_t *hist_out,
                              double *grid_probs_out);

/* Elo updates of a whole season on the device ("next" row f4): the events of F1EloSystem.update_quali_ratings
 * (reference src/elo.py:45-83) and update_race_ratings (:85-122), applied in order in ONE kernel launch with the
 * ratings resident in LDS between events.  Per event e: kind[e] = 0 updates the qualifying ratings, 1 the race
 * ratings; k[e] = the K factor set_recency_weight (:13-38) leaves for it; count[e] = m entries of the result list,
 * who[e * n_drivers + j] = driver index of entry j, value[e * n_drivers + j] = its best lap time (qualifying) or
 * finishing position (race): lower wins, equal values tie.  Every entry's delta is the sum over the other entries,
 * in list order, of k (actual - expected) / (m - 1) with the ratings BEFORE the event; all deltas are applied
 * afterwards; m < 2 changes nothing (:54-56, :93-94).  ratings = [2][n_drivers] (qualifying row, race row), in and
 * out, with drivers not seen yet at the caller's initial rating (the reference creates them at that value on first
 * appearance, :59-61).  after_out (optional, NULL to skip) = [n_events][2][n_drivers], the ratings after each
 * event.  `10 ** exponent` is the library's own (csrc/elo_update.h): bit-identical to the CPU oracle's restatement
 * of the same text and within a few ulp of the reference's ratings.  MCGP_E_BAD_ARG: n_drivers out of [1, 32], an
 * entry count above n_drivers, a driver index >= n_drivers or listed twice in one event, a kind other than 0 / 1,
 * a non-finite k, value or rating. */
int32_t mcgp_elo_season(uint32_t n_drivers, uint32_t n_events, const int32_t *kind, const double *k,
                        const uint32_t *count, const uint8_t *who, const double *value, double *ratings,
                        double *after_out, int32_t device);

/* Measurement hooks (bench.py): duration in ms of the race kernel(s) of a call, from
 * hipEvents the library records on the call's launch stream around its launches
 * (the query synchronises on the stop event).  Every stream keeps its own pair of
 * events, so calls on different streams of one device do not disturb each other:
 *   mcgp_stream_kernel_ms  the most recent call launched on `stream` (NULL = the
 *                          default stream, which is where mcgp_run and
 *                          mcgp_simulate_race launch); the library remembers the 8
 *                          most recently used streams per device;
 *   mcgp_last_kernel_ms    the most recent call on `device` by any thread --
 *                          meaningful when one thread drives the device.
 * mcgp_last_launch_info / mcgp_last_kernel_name describe that same most recent call -- including the block SHAPE the
 * launch ended up with: the register kernel's default block fills a CU's LDS almost completely (163 264 of 163 840 B at
 * 20 cars); on a device or under a runtime that offers less per block the launch falls back, by itself, to the same
 * kernel in blocks of 4 waves (block_threads = 256, name "mcgp::race_kernel_reg<n, 4>"), same results, about 15 % slower.
 * If not even that block fits, the call fails with MCGP_E_HIP and a message that names both sizes. */
int32_t mcgp_last_kernel_ms(int32_t device, float *ms_out);
int32_t mcgp_stream_kernel_ms(int32_t device, void *stream, float *ms_out);
int32_t mcgp_last_launch_info(int32_t device, uint32_t *grid_blocks, uint32_t *block_threads,
                              uint32_t *lds_bytes);
const char *mcgp_last_kernel_name(int32_t device);   /* "" before the first launch */

#ifdef __cplusplus
}
#endif
#endif
