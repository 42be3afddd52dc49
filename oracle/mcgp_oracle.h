/* mcgp_oracle.h -- CPU ORACLE for the Monte Carlo race-simulation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (reference: src/simulation.py:59-560).  It exists to CHECK the HIP
 * path and to be timed as the CPU baseline; nothing in the shipped package may
 * import, link or execute it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it.
 *
 * Two random back-ends drive the same race logic:
 *   MCGP_ORACLE_RNG_MT      two Mersenne-Twister streams replaying CPython's
 *                           `random` and numpy's legacy `np.random` bit for bit,
 *                           in the reference's call order.  With it the oracle
 *                           must reproduce the reference's integer histograms,
 *                           finishing orders and per-lap state EXACTLY
 *                           (pinned by tests/golden, see tests/test_oracle_golden.py).
 *   MCGP_ORACLE_RNG_PHILOX  counter-based Philox4x32-10, every draw addressed by
 *                           (seed, simulation id, lap, purpose, index); the HIP
 *                           kernel must match this back-end EXACTLY.
 *   MCGP_ORACLE_RNG_PHILOX53 the same draws refined to the reference's deviate precision: each
 *                           uniform keeps the 32-bit word as its leading bits and takes 21 more from a
 *                           companion Philox block; each normal is the binary64 inverse normal CDF at
 *                           that 53-bit point.  Run beside PHILOX on the same (seed, simulation id) it
 *                           measures what the product's 32-bit deviates change (tools/deviate_bias.py).
 */
#ifndef MCGP_ORACLE_H
#define MCGP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCGP_ORACLE_MAX_CARS 32
#define MCGP_ORACLE_RNG_MT 0
#define MCGP_ORACLE_RNG_PHILOX 1
#define MCGP_ORACLE_RNG_PHILOX53 2   /* PHILOX refined to 53-bit uniforms / binary64 normals (common random numbers) */

enum { ORC_SOFT = 0, ORC_MEDIUM = 1, ORC_HARD = 2, ORC_INTERMEDIATE = 3, ORC_WET = 4 };
enum { ORC_DRY = 0, ORC_DAMP = 1, ORC_WET_TRACK = 2 };

/* RaceConfig (reference src/simulation.py:37-52) with every dict resolved to a dense table. */
typedef struct {
    int32_t total_laps;
    int32_t track_condition;       /* ORC_DRY / ORC_DAMP / ORC_WET_TRACK (run_monte_carlo argument) */
    double pit_loss;
    double overtake_delta;
    double sc_probability;
    double vsc_probability;
    double red_flag_probability;
    double drs_delta;
    double dirty_air_threshold;
    double dirty_air_penalty;
    double comp_pace_delta[5];     /* tire_compounds[c].get('pace_delta', 0)   */
    double comp_deg_rate[5];       /* tire_compounds[c].get('deg_rate', 0.05)  */
    int32_t comp_optimal_laps[5];  /* tire_compounds[c].get('optimal_laps', 30) */
    int32_t pop_soft_hard;         /* result of {'SOFT','HARD'}.pop()   (Q13, hash-seed dependent) */
    int32_t pop_medium_hard;       /* result of {'MEDIUM','HARD'}.pop() */
} orc_config;

/* Per-driver inputs, index = position of the driver in grid_probs' key order.
 * The caller resolves the reference's .get(...) defaults:
 *   base_pace     base_pace.get(d, 90.0)                    :202,294,514
 *   tire_deg      tire_deg.get(d, 0.05)                     :203,295,514
 *   tire_deg_pit  tire_deg.get(d, 0.0)                      :458
 *   variance      driver_variance.get(d, 0.15)              :204,296
 *   team_dnf      config.dnf_rates.get(team(d), 0.002)      :286  (x4.0 on lap 1)
 *   lap_dnf       driver_dnf_rates.get(d, team_dnf)         :190-193 */
typedef struct {
    const double *base_pace, *tire_deg, *tire_deg_pit, *variance, *team_dnf, *lap_dnf;
} orc_drivers;

/* Optional per-lap observer: state after every _update_positions call of the
 * first n_trace simulations, cars in DRIVER-index order.  Arrays are
 * [n_trace][total_laps][n]; any pointer may be NULL. */
typedef struct {
    int64_t n_trace;
    double *cum, *tbl, *last;
    int16_t *age, *dnf_lap;
    uint8_t *comp, *used, *dnf, *drs;
} orc_trace;

/* Opaque MT stream pair (CPython `random` + numpy legacy RandomState). */
typedef struct orc_mt_state orc_mt_state;
orc_mt_state *orc_mt_new(void);
void orc_mt_free(orc_mt_state *);
void orc_mt_seed(orc_mt_state *, uint32_t seed);          /* random.seed(s); np.random.seed(s)  :77-78 */
double orc_mt_py_random(orc_mt_state *);                   /* random.random() */
double orc_mt_np_sample(orc_mt_state *);                   /* np.random.random_sample() */
double orc_mt_np_normal(orc_mt_state *, double loc, double scale);  /* np.random.normal(loc, scale) */
int orc_mt_np_choice(orc_mt_state *, const double *p, int n);      /* np.random.choice(n, p=p) */

/* Philox helpers exposed for unit tests. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_normal_from_u32(uint32_t w);
/* Phi^-1((q53 + 0.5) / 2^53) for q53 < 2^52 (lower tail), Newton from z_start; exposed for unit tests. */
double orc_phi_inverse_tail(uint64_t q53, float z_start);
/* the same value in the form the device computes it: degree-7 table, explicit fma (normal53_table.h) */
double orc_normal53_tail(uint64_t q53);
/* test hook: the lap (2 .. total_laps, 0 = none) on which a car with per-lap retirement probability p retires, from the
 * draw's word (and, wide != 0, the 21 refinement bits of its companion): the once-per-race law of the counter-based
 * back-ends (reference src/simulation.py:190-197 drawn as one geometric variate) */
int orc_retirement_lap(uint32_t w, uint32_t extra21, int wide, double p, int total_laps);

/* Run n_sims simulations.
 *   grid_probs  n x n row-major [driver][slot]
 *   hist        n x n row-major [driver][position-1], ACCUMULATED (caller zeroes)
 *   orders      optional [n_sims][n]: driver index at finishing position p
 *   grids       optional [n_sims][n]: driver index at grid slot p
 * rng = MT:     mt must be non-NULL; draws continue from its current state
 *               (seed it first for `seed=...`, leave it for `seed=None`, Q20);
 *               seed / sim_offset are ignored.
 * rng = PHILOX: draws are a pure function of (seed, sim_offset + i); mt unused.
 * Returns 0, or a negative value on a bad argument. */
int orc_run(const orc_config *cfg, const orc_drivers *drv, const double *grid_probs, int32_t n,
            int64_t n_sims, uint64_t sim_offset, uint64_t seed, int32_t rng, orc_mt_state *mt,
            uint64_t *hist, uint8_t *orders, uint8_t *grids, const orc_trace *trace);

/* Simulate one race from a FIXED grid (reference simulate_race :147-242).  MT only draws
 * continue from mt; Philox uses (seed, sim_id).  order_out[p] = driver index at position p. */
int orc_simulate_race(const orc_config *cfg, const orc_drivers *drv, const uint8_t *grid, int32_t n,
                      uint64_t sim_id, uint64_t seed, int32_t rng, orc_mt_state *mt, uint8_t *order_out);

/* _sample_grid only (reference :102-145), MT back-end; for the G5 fixtures. */
int orc_sample_grid_mt(const double *grid_probs, int32_t n, orc_mt_state *mt, uint8_t *grid_out);

/* Grid-probability front end, reference src/elo.py:124-141 + src/predictor.py:321-407, with the front end's own
 * exp (frontend_exp.h).  out: n x n row-major [driver][grid slot]. */
double orc_fe_exp(double x);
int orc_grid_probs(const double *rating, const double *teammate_delta, const double *form_score,
                   const double *circuit_affinity, const int32_t *penalty, int32_t n, double *out);

/* A season of Elo updates, reference src/elo.py:40-122, with the library's own 10^x (elo_update.h).  Arguments as
 * mcgp_elo_season (include/mcgp.h); -1 on a malformed argument. */
double orc_elo_pow10(double x);
int orc_elo_season(int32_t n, int32_t n_events, const int32_t *kind, const double *k, const uint32_t *count,
                   const uint8_t *who, const double *value, double *ratings, double *after_out);

#ifdef __cplusplus
}
#endif
#endif
