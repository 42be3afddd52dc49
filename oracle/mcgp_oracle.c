/* mcgp_oracle.c -- CPU ORACLE (plain C) for the Monte Carlo race simulation.
 *
 * TEST INFRASTRUCTURE ONLY -- see mcgp_oracle.h.  Not part of the product path.
 *
 * Restates reference src/simulation.py function by function; every function
 * cites the reference lines it follows.  The arithmetic is IEEE binary64 in
 * the reference's own evaluation order (compile with -ffp-contract=off).
 * Parity status: PINNED -- the MT back-end reproduces the reference's own
 * outputs (tests/golden/<case>.npz, produced by tests/golden/make_goldens.py from
 * the reference under PYTHONHASHSEED=0) bit for bit.
 */
#include "mcgp_oracle.h"
#include "normal_table.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Mersenne Twister MT19937 (Matsumoto & Nishimura), as embedded in CPython's */
/* Modules/_randommodule.c and numpy/random/src/mt19937 (both third-party to  */
/* the reference; call sites src/simulation.py:77-78,137,168-174,194,287,302, */
/* 330,392,524).  Pinned by tests/golden/mt_streams.npz (G4).                 */
/* ------------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int idx;
} mt19937;

struct orc_mt_state {
    mt19937 py;      /* stdlib `random`            (stream U of SURVEY 3.3) */
    mt19937 np;      /* legacy global `np.random`  (stream G)               */
    int has_gauss;   /* numpy legacy gauss cache; survives choice() calls   */
    double gauss;
};

static void mt_init_genrand(mt19937 *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = MT_N;
}

static void mt_init_by_array(mt19937 *s, const uint32_t *key, int len)
{
    mt_init_genrand(s, 19650218u);
    int i = 1, j = 0;
    int k = MT_N > len ? MT_N : len;
    for (; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { s->mt[0] = s->mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { s->mt[0] = s->mt[MT_N - 1]; i = 1; }
    }
    s->mt[0] = 0x80000000u;
    s->idx = MT_N;
}

static uint32_t mt_next(mt19937 *s)
{
    if (s->idx >= MT_N) {
        uint32_t *mt = s->mt;
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < MT_N - 1; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* genrand_res53: random.random() and numpy's legacy next_double. */
static double mt_res53(mt19937 *s)
{
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

orc_mt_state *orc_mt_new(void)
{
    orc_mt_state *s = (orc_mt_state *)calloc(1, sizeof(*s));
    if (s) orc_mt_seed(s, 0);
    return s;
}

void orc_mt_free(orc_mt_state *s) { free(s); }

/* random.seed(int) -> init_by_array over the 32-bit words of |seed|;
 * np.random.seed(int) -> init_genrand(seed) and gauss cache cleared.   (:77-78) */
void orc_mt_seed(orc_mt_state *s, uint32_t seed)
{
    uint32_t key[1] = { seed };
    mt_init_by_array(&s->py, key, 1);
    mt_init_genrand(&s->np, seed);
    s->has_gauss = 0;
    s->gauss = 0.0;
}

double orc_mt_py_random(orc_mt_state *s) { return mt_res53(&s->py); }
double orc_mt_np_sample(orc_mt_state *s) { return mt_res53(&s->np); }

/* numpy legacy_gauss: polar Box-Muller with a one-deep cache. */
static double np_legacy_gauss(orc_mt_state *s)
{
    if (s->has_gauss) {
        const double t = s->gauss;
        s->has_gauss = 0;
        s->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_res53(&s->np) - 1.0;
        x2 = 2.0 * mt_res53(&s->np) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}

/* np.random.normal(loc, scale) = loc + scale * legacy_gauss. */
double orc_mt_np_normal(orc_mt_state *s, double loc, double scale)
{
    return loc + scale * np_legacy_gauss(s);
}

/* np.random.choice(n, p=p), size=None: cdf = p.cumsum(); cdf /= cdf[-1];
 * idx = cdf.searchsorted(random_sample(), side='right').                */
static int choice_from_uniform(const double *p, int n, double u)
{
    double cdf[MCGP_ORACLE_MAX_CARS];
    double acc = p[0];
    cdf[0] = acc;
    for (int i = 1; i < n; i++) { acc = acc + p[i]; cdf[i] = acc; }
    const double last = cdf[n - 1];
    int idx = 0;
    for (int i = 0; i < n; i++) {
        cdf[i] = cdf[i] / last;
        if (cdf[i] <= u) idx = i + 1;   /* cdf is non-decreasing: count of entries <= u */
    }
    return idx;
}

int orc_mt_np_choice(orc_mt_state *s, const double *p, int n)
{
    return choice_from_uniform(p, n, mt_res53(&s->np));
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11) and the u32 -> deviate transforms of   */
/* the counter-based back-end.                                                */
/* ------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double bits_to_double(unsigned long long b)
{
    double d;
    memcpy(&d, &b, 8);
    return d;
}
static inline float bits_to_float(uint32_t b)
{
    float f;
    memcpy(&f, &b, 4);
    return f;
}

/* One 32-bit word -> N(0,1) by the piecewise-cubic inverse CDF of
 * tools/gen_normal_table.py.  Only integer ops and binary32 fma: the HIP
 * kernel evaluates the same expression and gets the same bits. */
float orc_normal_from_u32(uint32_t w)
{
    /* float(m + 16): exponent 4 .. 31 = the octave, its four leading mantissa bits = the cell, the other 19 = the offset
     * inside the cell (tools/gen_normal_table.py: no special case for the smallest m); the rows are stored from
     * exponent 4 on, i.e. from row (131 mod 32) * 16 = 48 of the bit field */
    const float fm = (float)((w & 0x7fffffffu) + 16u);                      /* round to nearest even */
    uint32_t b;
    memcpy(&b, &fm, 4);
    const uint32_t row = ((b >> 19) & 0x1ffu) - 48u;
    const float t = (float)(b & 0x7ffffu);                                  /* cell coordinate x 2^19 */
    const unsigned int *cf = &mcgp_normal_table_bits[4 * row];
    float z = __builtin_fmaf(bits_to_float(cf[3]), t, bits_to_float(cf[2]));
    z = __builtin_fmaf(z, t, bits_to_float(cf[1]));
    z = __builtin_fmaf(z, t, bits_to_float(cf[0]));
    return (w >> 31) ? -z : z;
}

/* ---- 53-bit refinement of the counter-based deviates (MCGP_ORACLE_RNG_PHILOX53) ----
 * The reference draws 53-bit uniforms (genrand_res53) and exact normals (reference :137,194,302,330,524);
 * the product draws 32-bit words.  This back-end measures what that substitution does, by COMMON RANDOM
 * NUMBERS: every draw keeps the word w the 32-bit back-end reads (same Philox address) as its leading 32
 * bits and takes 21 more from a companion block (counter word 3 | 0x8000, same word position):
 *     uniform  u53 = (w 2^21 + e) / 2^53                  in [w / 2^32, (w + 1) / 2^32)
 *     normal   sign = w >> 31, m = w & 0x7fffffff, tail probability p = (m 2^21 + e + 0.5) / 2^53
 *              (the 32-bit draw's (m + 0.5) / 2^32 refined), z = -+ Phi^-1(p) in binary64
 *              (normal53_tail below: the degree-7 table form the device computes, |err| <= 3e-15; the erfc iteration
 *              that follows is kept as an independent check of that table)
 * so a race simulated with both back-ends differs only by the refinement itself.  The library's `deviates = 53` mode
 * (include/mcgp.h) computes exactly this back-end on the GPU. */
static double phi_inverse_tail(uint64_t q53, float z_start)
{
    /* p = (q53 + 0.5) / 2^53 in (0, 0.5): lower-tail probability; returns z = Phi^-1(p) < 0.
     * Newton on F(z) = erfc(-z / sqrt 2) / 2 (erfc of a positive argument: full relative accuracy in the
     * tail), started from the product's own piecewise cubic (|err| <= 4.8e-7): two steps reach binary64. */
    const double p = ((double)q53 + 0.5) * 0x1p-53;
    double z = (double)z_start;
    if (z > -1e-300) z = -1e-300;
    /* two steps suffice from the table value; only the cells of the 16 smallest m, which span decades of
     * p, start far away and take more */
    for (int it = 0; it < 100; it++) {
        const double F = 0.5 * erfc(-z * 0.70710678118654752440);
        const double pdf = 0.39894228040143267794 * exp(-0.5 * z * z);
        const double d = (F - p) / pdf;
        const double step = d / (1.0 + 0.5 * z * d);  /* Halley correction: F'' / F' = -z */
        z = z - step;
        if (fabs(step) <= 1e-16 * fabs(z)) break;
    }
    return z;
}

double orc_phi_inverse_tail(uint64_t q53, float z_start) { return phi_inverse_tail(q53, z_start); }

/* The binary64 deviate of the reference-width back-end, in the form the DEVICE computes it (csrc/race_common.hip.h:
 * normal53_tail): a piecewise polynomial of degree 7 on log-spaced cells of the 52-bit tail index q, explicit fma
 * only -- the same bits on every machine (tools/gen_normal53_table.py; |err| <= 3e-15 against Phi^-1, checked against
 * the erfc iteration above by tests/test_oracle_golden.py).  Returns z0 = Phi^-1((q + 0.5) / 2^53) < 0. */
#include "normal53_table.h"
static double normal53_tail(uint64_t q)
{
    uint32_t row;
    double t;
    if (q < 16u) {
        row = (uint32_t)q;
        t = 0.0;
    } else {
        const int hb = 63 - __builtin_clzll(q);
        const int sh = hb - 4;
        const uint32_t k = (uint32_t)(q >> sh) & 15u;
        const uint64_t r = q & (((uint64_t)1 << sh) - 1u);
        t = ((double)r + 0.5) * ldexp(1.0, -sh);               /* exact: r < 2^47, a power of two */
        row = 16u + 16u * (uint32_t)sh + k;
    }
    const unsigned long long *c = &mcgp_normal53_table_bits[(size_t)MCGP_NORMAL53_COEFFS * row];
    double z = bits_to_double(c[MCGP_NORMAL53_COEFFS - 1]);
    for (int d = MCGP_NORMAL53_COEFFS - 2; d >= 0; d--) z = fma(z, t, bits_to_double(c[d]));
    return z;
}
double orc_normal53_tail(uint64_t q) { return normal53_tail(q); }

/* ------------------------------------------------------------------------- */
/* Random source seen by the race logic.  The logic asks for draws in the      */
/* reference's order; the MT back-end answers from its streams, the Philox     */
/* back-end from the draw's address.                                           */
/*   counter = { sim_lo, sim_hi, lap, purpose << 16 | index }, key = seed      */
/*   GRID  (lap 0)  index = slot >> 2, word = slot & 3                         */
/*   EVENT (lap)    words: red flag, safety car, VSC, VSC tyre draw            */
/*   CAR   (lap 1)  index = driver; words: DNF, lap noise, start delta          */
/*   RETIRE (lap 0) index = driver >> 2, word driver & 3: the lap (>= 2) on     */
/*                   which the driver retires, drawn ONCE per race: with        */
/*                   t = ceil(p 2^32), q = 2^32 - t the car survives lap k iff  */
/*                   w < S_k, S_2 = q, S_{k+1} = floor(S_k q / 2^32) (the lap   */
/*                   of the first success of the reference's per-lap draw,      */
/*                   :190-197, is geometric; philox_retirement_lap below)       */
/*   CAR   (lap>=2) index = place >> 2 (one block serves four cars), word       */
/*                   place & 3 = lap noise, where `place` is                    */
/*                   the car's place in the FIELD ORDER the lap starts with:    */
/*                   all cars, retired ones included, in the time order of the  */
/*                   end of the previous lap (stable, i.e. grid order on equal  */
/*                   times) -- re-sorted only if a VSC has just rounded two     */
/*                   running cars onto the same time (field_order_* below)      */
/*   OVT   (lap)    the k-th overtake ATTEMPT of pass p (k counted along the   */
/*                   pass's sorted order) reads word k & 3 of index 8p + k / 4  */
/* ------------------------------------------------------------------------- */
enum { PURPOSE_GRID = 0, PURPOSE_EVENT = 1, PURPOSE_CAR = 2, PURPOSE_OVT = 3, PURPOSE_RETIRE = 4 };

typedef struct {
    int mode;
    orc_mt_state *mt;
    uint32_t key[2];
    uint64_t sim;
} rng_t;

static uint32_t philox_word(const rng_t *r, uint32_t lap, uint32_t purpose, uint32_t index, int word)
{
    const uint32_t ctr[4] = { (uint32_t)r->sim, (uint32_t)(r->sim >> 32), lap, (purpose << 16) | index };
    uint32_t out[4];
    orc_philox4x32_10(ctr, r->key, out);
    return out[word];
}

static inline double u32_to_unit(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }

/* the 21 refinement bits of a draw: top bits of the same word of the companion block */
static uint32_t philox_extra21(const rng_t *r, uint32_t lap, uint32_t purpose, uint32_t index, int word)
{
    return philox_word(r, lap, purpose, 0x8000u | index, word) >> 11;
}
/* uniform of the draw at (lap, purpose, index, word) under the back-end in use (Philox family only) */
static double philox_uniform(const rng_t *r, uint32_t lap, uint32_t purpose, uint32_t index, int word)
{
    const uint32_t w = philox_word(r, lap, purpose, index, word);
    if (r->mode != MCGP_ORACLE_RNG_PHILOX53) return u32_to_unit(w);
    const uint64_t q = ((uint64_t)w << 21) | philox_extra21(r, lap, purpose, index, word);
    return (double)q * 0x1p-53;
}
/* standard normal of the draw at that address */
static double philox_normal(const rng_t *r, uint32_t lap, uint32_t purpose, uint32_t index, int word)
{
    const uint32_t w = philox_word(r, lap, purpose, index, word);
    const float z32 = orc_normal_from_u32(w);
    if (r->mode != MCGP_ORACLE_RNG_PHILOX53) return (double)z32;
    const uint64_t q = ((uint64_t)(w & 0x7fffffffu) << 21) | philox_extra21(r, lap, purpose, index, word);
    /* tail probability (m + 0.5) / 2^32 of the 32-bit draw refined to (q + 0.5) / 2^53 */
    const double z0 = normal53_tail(q);
    return (w >> 31) ? -z0 : z0;
}

static double draw_grid(rng_t *r, int slot)
{
    if (r->mode == MCGP_ORACLE_RNG_MT) return mt_res53(&r->mt->np);          /* :137 choice */
    return philox_uniform(r, 0, PURPOSE_GRID, (uint32_t)slot >> 2, slot & 3);
}
static double draw_event(rng_t *r, int lap, int which)
{
    if (r->mode == MCGP_ORACLE_RNG_MT) return mt_res53(&r->mt->py);          /* :168,171,174,392 */
    return philox_uniform(r, (uint32_t)lap, PURPOSE_EVENT, 0, which);
}
/* `who`: the driver index (lap 1 of the counter-based back-ends; the MT back-end draws in sequence, every lap) */
static double draw_dnf(rng_t *r, int lap, int who)
{
    if (r->mode == MCGP_ORACLE_RNG_MT) return mt_res53(&r->mt->py);          /* :194,287 */
    (void)lap;
    return philox_uniform(r, 1u, PURPOSE_CAR, (uint32_t)who, 0);
}
/* Counter-based back-ends, laps >= 2: the lap on which `driver` retires, 0 = not within `total_laps` (see the table
 * above).  The reference draws u < p afresh every lap (:190-197); the lap of the first success is geometric and is
 * drawn here from one word.  PHILOX: 32-bit word against 32-bit thresholds, exactly what the HIP kernels compute
 * (csrc/race_common.hip.h: draw_retirement_lap).  PHILOX53: the same word refined to 53 bits (left-aligned in 64)
 * against 64-bit thresholds S_2 = q, S_{k+1} = floor(S_k q / 2^64), q = 2^64 - ceil(p 2^64). */
/* (the chain on given words: `w` the draw's word, `extra21` the 21 refinement bits of its companion, PHILOX53 only) */
static int retirement_lap_of_words(uint32_t w, uint32_t extra21, int wide, double p, int total_laps)
{
    if (!(p > 0.0)) return 0;                                              /* also NaN: `u < nan` is false */
    if (!wide) {
        const double x = p * 4294967296.0;                                 /* exact */
        const uint64_t t = x >= 4294967296.0 ? 4294967296ull : (uint64_t)ceil(x);
        const uint32_t q = (uint32_t)(4294967296ull - t);
        uint32_t S = q;
        for (int k = 2; k <= total_laps; k++) {
            if (!(w < S)) return k;
            S = (uint32_t)(((uint64_t)S * (uint64_t)q) >> 32);
        }
        return 0;
    }
    const uint64_t Q = ((uint64_t)w << 32) | ((uint64_t)extra21 << 11);
    if (p >= 1.0) return total_laps >= 2 ? 2 : 0;
    const double x = p * 18446744073709551616.0;                           /* exact: p 2^64 < 2^64 */
    const uint64_t t = (uint64_t)ceil(x);                                  /* >= 1 */
    const uint64_t q = (uint64_t)0 - t;                                    /* 2^64 - t */
    uint64_t S = q;
    for (int k = 2; k <= total_laps; k++) {
        if (!(Q < S)) return k;
        S = (uint64_t)(((unsigned __int128)S * (unsigned __int128)q) >> 64);
    }
    return 0;
}
static int philox_retirement_lap(const rng_t *r, int driver, double p, int total_laps)
{
    const uint32_t w = philox_word(r, 0u, PURPOSE_RETIRE, (uint32_t)driver >> 2, driver & 3);
    const int wide = r->mode == MCGP_ORACLE_RNG_PHILOX53;
    if (!(p > 0.0)) return 0;
    return retirement_lap_of_words(w, wide ? philox_extra21(r, 0u, PURPOSE_RETIRE, (uint32_t)driver >> 2, driver & 3) : 0u,
                                   wide, p, total_laps);
}
/* test hook (tests/test_philox_layers.py): the retirement law on given words */
int orc_retirement_lap(uint32_t w, uint32_t extra21, int wide, double p, int total_laps)
{
    return retirement_lap_of_words(w, extra21, wide, p, total_laps);
}
static double draw_overtake(rng_t *r, int lap, int pass, int attempt)
{
    if (r->mode == MCGP_ORACLE_RNG_MT) return mt_res53(&r->mt->py);          /* :524 */
    return philox_uniform(r, (uint32_t)lap, PURPOSE_OVT, (uint32_t)(8 * pass + (attempt >> 2)), attempt & 3);
}
/* np.random.normal(0, scale) */
static double draw_lap_noise(rng_t *r, int lap, int who, double scale)
{
    if (r->mode == MCGP_ORACLE_RNG_MT) return orc_mt_np_normal(r->mt, 0.0, scale);   /* :330 */
    const double z = lap == 1 ? philox_normal(r, 1u, PURPOSE_CAR, (uint32_t)who, 1)
                              : philox_normal(r, (uint32_t)lap, PURPOSE_CAR, (uint32_t)who >> 2, who & 3);
    return 0.0 + scale * z;
}
static double draw_start_delta(rng_t *r, int driver, double scale)
{
    if (r->mode == MCGP_ORACLE_RNG_MT) return orc_mt_np_normal(r->mt, 0.0, scale);   /* :302 */
    return 0.0 + scale * philox_normal(r, 1u, PURPOSE_CAR, (uint32_t)driver, 2);
}

/* ------------------------------------------------------------------------- */
/* CarState (reference :9-34).  `cars` stays in sampled-grid order (Q1).       */
/* ------------------------------------------------------------------------- */
typedef struct {
    int driver;                 /* index into the per-driver arrays */
    int position;               /* :13, only read on lap 1          */
    int lap;                    /* :14, lap of retirement for DNFs  */
    int tire_compound;          /* :15 */
    int tire_age;               /* :16 */
    double fuel_load;           /* :17 */
    double time_behind_leader;  /* :18 */
    double cumulative_time;     /* :20 */
    int drs_enabled;            /* :21 */
    int dnf;                    /* :22 */
    unsigned used_compounds;    /* :25, bit set over compound ids */
    int laps_completed;         /* :27 */
    double last_lap_time;       /* :29 */
} car_t;

typedef struct {
    const orc_config *cfg;
    const orc_drivers *drv;
    int n;
    rng_t *rng;
} sim_t;

/* Python's sorted()/list.sort() are stable: insertion sort on index lists. */
static void stable_sort_by_time(const car_t *cars, int *idx, int m)
{
    for (int i = 1; i < m; i++) {
        const int x = idx[i];
        const double key = cars[x].cumulative_time;
        int j = i;
        while (j > 0 && cars[idx[j - 1]].cumulative_time > key) { idx[j] = idx[j - 1]; j--; }
        idx[j] = x;
    }
}

static int active_sorted(const car_t *cars, int n, int *idx)
{
    int m = 0;
    for (int i = 0; i < n; i++) if (!cars[i].dnf) idx[m++] = i;
    stable_sort_by_time(cars, idx, m);
    return m;
}

/* _sample_grid, reference :102-145.  Returns driver indices per grid slot. */
static void sample_grid(const double *G, int n, rng_t *rng, uint8_t *grid)
{
    int remaining[MCGP_ORACLE_MAX_CARS];
    double probs[MCGP_ORACLE_MAX_CARS];
    int n_remaining = n;
    for (int d = 0; d < n; d++) remaining[d] = 1;
    for (int pos = 0; pos < n; pos++) {
        /* :119-123 */
        double total = 0.0;
        for (int d = 0; d < n; d++) {
            probs[d] = remaining[d] ? G[d * n + pos] : 0.0;
            total = total + probs[d];
        }
        if (total > 0) {                                   /* :125-126 */
            for (int d = 0; d < n; d++) probs[d] = probs[d] / total;
        } else {                                           /* :127-130 */
            for (int d = 0; d < n; d++) probs[d] = remaining[d] ? 1.0 / n_remaining : 0.0;
        }
        double prob_sum = 0.0;                             /* :133-135 */
        for (int d = 0; d < n; d++) prob_sum = prob_sum + probs[d];
        if (prob_sum > 0 && fabs(prob_sum - 1.0) > 1e-9)
            for (int d = 0; d < n; d++) probs[d] = probs[d] / prob_sum;
        const int sel = choice_from_uniform(probs, n, draw_grid(rng, pos));   /* :137 */
        grid[pos] = (uint8_t)sel;
        if (remaining[sel]) { remaining[sel] = 0; n_remaining--; }           /* :139 */
    }
}

/* _initialize_cars, reference :244-273 */
static void initialize_cars(const sim_t *s, const uint8_t *grid, car_t *cars)
{
    const int tc = s->cfg->track_condition;
    for (int pos = 0; pos < s->n; pos++) {
        car_t *c = &cars[pos];
        memset(c, 0, sizeof(*c));
        c->driver = grid[pos];
        c->position = pos + 1;
        c->lap = 0;
        if (tc == ORC_WET_TRACK) c->tire_compound = ORC_WET;
        else if (tc == ORC_DAMP) c->tire_compound = ORC_INTERMEDIATE;
        else c->tire_compound = pos < 10 ? ORC_SOFT : ORC_MEDIUM;
        c->tire_age = (tc != ORC_DRY) ? 0 : (pos < 10 ? 4 : 0);
        c->fuel_load = 110.0;
        c->time_behind_leader = 0.0;
        c->cumulative_time = 0.0;
        c->used_compounds = 1u << c->tire_compound;       /* __post_init__ :31-34 */
    }
}

/* _calculate_lap_time, reference :313-332 */
static double calculate_lap_time(const sim_t *s, const car_t *car, int lap, int who)
{
    const orc_config *cfg = s->cfg;
    const double base = s->drv->base_pace[car->driver];
    const double deg = s->drv->tire_deg[car->driver];
    const double variance = s->drv->variance[car->driver];
    const double compound_deg = cfg->comp_deg_rate[car->tire_compound];
    const double driver_factor = deg > 0 ? deg / 0.05 : 1.0;
    const double effective_deg = compound_deg * driver_factor;
    const double tire_effect = (double)car->tire_age * effective_deg;
    const double fuel_effect = (110.0 - car->fuel_load) * 0.03;
    const double compound_delta = cfg->comp_pace_delta[car->tire_compound];
    const double drs_gain = car->drs_enabled ? cfg->drs_delta : 0.0;
    const double noise = draw_lap_noise(s->rng, lap, who, variance);
    return base + tire_effect - fuel_effect + compound_delta - drs_gain + noise;
}

/* _update_positions, reference :538-560 */
static void update_positions(const sim_t *s, car_t *cars, int lap, int drs_disabled)
{
    int idx[MCGP_ORACLE_MAX_CARS];
    const int m = active_sorted(cars, s->n, idx);
    for (int i = 0; i < m; i++) {
        car_t *car = &cars[idx[i]];
        car->position = i + 1;
        car->time_behind_leader = car->cumulative_time - cars[idx[0]].cumulative_time;
        if (lap <= 2 || drs_disabled || i == 0) {
            car->drs_enabled = 0;
        } else {
            const double gap = car->cumulative_time - cars[idx[i - 1]].cumulative_time;
            car->drs_enabled = gap < 1.0;
        }
    }
}

/* _simulate_lap_1, reference :275-311 */
static void simulate_lap_1(const sim_t *s, car_t *cars)
{
    for (int i = 0; i < s->n; i++) {
        car_t *car = &cars[i];
        const double base_dnf_rate = s->drv->team_dnf[car->driver];
        if (draw_dnf(s->rng, 1, car->driver) < base_dnf_rate * 4.0) {
            car->dnf = 1;
            car->lap = 1;
            continue;
        }
        const double base_lap_time = calculate_lap_time(s, car, 1, car->driver);
        double position_factor = 0.5 + (double)car->position * 0.1;
        if (!(position_factor < 1.5)) position_factor = 1.5;             /* min(1.5, x) */
        double start_delta = draw_start_delta(s->rng, car->driver, position_factor);
        if (car->position <= 3 && 1.0 < start_delta) start_delta = 1.0;  /* min(start_delta, 1.0) */
        const double lap_time = base_lap_time - start_delta * 0.5;
        car->cumulative_time += lap_time;
        car->tire_age += 1;
        car->fuel_load = (car->fuel_load - 1.5 > 0) ? car->fuel_load - 1.5 : 0.0;
        car->lap = 1;
    }
    update_positions(s, cars, 1, 1);
}

static int red_flag_compound(const orc_config *cfg, int remaining_laps)
{
    if (cfg->track_condition == ORC_WET_TRACK) return ORC_WET;
    if (cfg->track_condition == ORC_DAMP) return ORC_INTERMEDIATE;
    if (remaining_laps > 30) return ORC_HARD;
    if (remaining_laps > 15) return ORC_MEDIUM;
    return ORC_SOFT;
}

/* _handle_safety_car, reference :334-376 */
static void handle_safety_car(const sim_t *s, car_t *cars)
{
    int idx[MCGP_ORACLE_MAX_CARS];
    const int m = active_sorted(cars, s->n, idx);
    if (!m) return;
    const double leader_time = cars[idx[0]].cumulative_time;
    const int leader_laps = cars[idx[0]].laps_completed;
    for (int i = 0; i < m; i++) {
        car_t *car = &cars[idx[i]];
        const int laps_down = leader_laps - car->laps_completed;
        if (laps_down <= 0)
            car->cumulative_time = leader_time + (double)i * 0.5;
        else
            car->cumulative_time = leader_time + ((double)laps_down * 90.0) + (double)i * 0.5;
        car->time_behind_leader = car->cumulative_time - leader_time;
        car->tire_age = car->tire_age - 1 > 0 ? car->tire_age - 1 : 0;
    }
}

/* _handle_vsc, reference :378-395 */
static void handle_vsc(const sim_t *s, car_t *cars, int lap)
{
    int idx[MCGP_ORACLE_MAX_CARS];
    const int m = active_sorted(cars, s->n, idx);
    if (!m) return;
    const double leader_time = cars[idx[0]].cumulative_time;
    for (int i = 0; i < m; i++) {
        car_t *car = &cars[idx[i]];
        const double gap = car->cumulative_time - leader_time;
        car->cumulative_time = leader_time + gap * 0.8;
        car->time_behind_leader = car->cumulative_time - leader_time;
    }
    if (draw_event(s->rng, lap, 3) < 0.3)
        for (int i = 0; i < m; i++) {
            car_t *car = &cars[idx[i]];
            car->tire_age = car->tire_age - 1 > 0 ? car->tire_age - 1 : 0;
        }
}

/* _handle_red_flag, reference :397-431 */
static void handle_red_flag(const sim_t *s, car_t *cars, int lap)
{
    int idx[MCGP_ORACLE_MAX_CARS];
    const int m = active_sorted(cars, s->n, idx);
    if (!m) return;
    const double leader_time = cars[idx[0]].cumulative_time;
    const int remaining_laps = s->cfg->total_laps - lap;
    for (int i = 0; i < m; i++) {
        car_t *car = &cars[idx[i]];
        car->cumulative_time = leader_time + (double)i * 0.1;
        car->time_behind_leader = car->cumulative_time - leader_time;
        car->tire_age = 0;
        car->tire_compound = red_flag_compound(s->cfg, remaining_laps);
        car->used_compounds |= 1u << car->tire_compound;
    }
}

/* _handle_pit_stops, reference :433-494 */
static void handle_pit_stops(const sim_t *s, car_t *cars, int lap)
{
    const orc_config *cfg = s->cfg;
    const int remaining_laps = cfg->total_laps - lap;
    const unsigned dry_compounds = (1u << ORC_SOFT) | (1u << ORC_MEDIUM) | (1u << ORC_HARD);
    const int is_wet = cfg->track_condition != ORC_DRY;
    for (int i = 0; i < s->n; i++) {
        car_t *car = &cars[i];
        if (car->dnf) continue;
        int optimal_laps = cfg->comp_optimal_laps[car->tire_compound];       /* :454-455 */
        const double driver_deg = s->drv->tire_deg_pit[car->driver];         /* :458 */
        if (driver_deg > 0.05) optimal_laps = (int)((double)optimal_laps * 0.85);
        else if (driver_deg < 0.02) optimal_laps = (int)((double)optimal_laps * 1.1);
        if (car->tire_age > optimal_laps && remaining_laps > 5) {            /* :465 */
            car->cumulative_time += cfg->pit_loss;
            int new_compound;
            if (cfg->track_condition == ORC_WET_TRACK) new_compound = ORC_WET;
            else if (cfg->track_condition == ORC_DAMP) new_compound = ORC_INTERMEDIATE;
            else if (remaining_laps > 30) new_compound = ORC_HARD;
            else if (remaining_laps > 15) new_compound = ORC_MEDIUM;
            else new_compound = ORC_SOFT;
            const unsigned used_dry = car->used_compounds & dry_compounds;   /* :481 */
            if (__builtin_popcount(used_dry) == 1 && (used_dry >> new_compound & 1u) && !is_wet) {
                const unsigned available = dry_compounds & ~used_dry;
                /* available.pop() on a 2-element set: explicit rule from the config (Q13) */
                int popped;
                if (available == ((1u << ORC_SOFT) | (1u << ORC_HARD))) popped = cfg->pop_soft_hard;
                else if (available == ((1u << ORC_MEDIUM) | (1u << ORC_HARD))) popped = cfg->pop_medium_hard;
                else popped = ORC_SOFT;  /* {SOFT, MEDIUM}: never popped (both branches find their first choice) */
                if (remaining_laps > 20)
                    new_compound = (available >> ORC_MEDIUM & 1u) ? ORC_MEDIUM : popped;
                else
                    new_compound = (available >> ORC_SOFT & 1u) ? ORC_SOFT : popped;
            }
            car->tire_compound = new_compound;
            car->used_compounds |= 1u << new_compound;
            car->tire_age = 0;
        }
    }
}

/* _simulate_overtakes, reference :496-536 */
static void simulate_overtakes(const sim_t *s, car_t *cars, int lap)
{
    const orc_config *cfg = s->cfg;
    for (int pass = 0; pass < 3; pass++) {
        int overtake_occurred = 0;
        int attempt = 0;
        int idx[MCGP_ORACLE_MAX_CARS];
        for (int i = 0; i < s->n; i++) idx[i] = i;           /* sorted(cars): DNF cars included (Q15) */
        stable_sort_by_time(cars, idx, s->n);
        for (int i = 1; i < s->n; i++) {
            car_t *behind = &cars[idx[i]];
            car_t *ahead = &cars[idx[i - 1]];
            if (behind->dnf || ahead->dnf) continue;
            const double pace_behind = s->drv->base_pace[behind->driver] + (double)behind->tire_age * s->drv->tire_deg[behind->driver];
            const double pace_ahead = s->drv->base_pace[ahead->driver] + (double)ahead->tire_age * s->drv->tire_deg[ahead->driver];
            double pace_delta = pace_ahead - pace_behind;
            if (behind->drs_enabled) pace_delta += cfg->drs_delta;
            if (pace_delta > cfg->overtake_delta) {
                double overtake_prob = pace_delta / 2.0;
                if (!(overtake_prob < 0.5)) overtake_prob = 0.5;       /* min(0.5, x) */
                if (draw_overtake(s->rng, lap, pass, attempt++) < overtake_prob) {
                    double new_behind_time = ahead->cumulative_time - 0.1;
                    if (!(new_behind_time > 0.1)) new_behind_time = 0.1;   /* max(0.1, x) */
                    behind->cumulative_time = new_behind_time;
                    ahead->cumulative_time = new_behind_time + 0.3;
                    overtake_occurred = 1;
                }
            }
        }
        if (!overtake_occurred) break;
    }
}

static void record_trace(const sim_t *s, const car_t *cars, const orc_trace *tr, int64_t sim_index, int lap)
{
    if (!tr || sim_index >= tr->n_trace) return;
    const int n = s->n;
    const size_t base = ((size_t)sim_index * (size_t)s->cfg->total_laps + (size_t)(lap - 1)) * (size_t)n;
    for (int i = 0; i < n; i++) {
        const car_t *c = &cars[i];
        const size_t o = base + (size_t)c->driver;
        if (tr->cum) tr->cum[o] = c->cumulative_time;
        if (tr->tbl) tr->tbl[o] = c->time_behind_leader;
        if (tr->last) tr->last[o] = c->last_lap_time;
        if (tr->age) tr->age[o] = (int16_t)c->tire_age;
        if (tr->comp) tr->comp[o] = (uint8_t)c->tire_compound;
        if (tr->used) tr->used[o] = (uint8_t)c->used_compounds;
        if (tr->dnf) tr->dnf[o] = (uint8_t)c->dnf;
        if (tr->drs) tr->drs[o] = (uint8_t)c->drs_enabled;
        if (tr->dnf_lap) tr->dnf_lap[o] = (int16_t)(c->dnf ? c->lap : 0);
    }
}

/* Field order of the counter-based back-ends (not part of the reference, which draws from sequential streams):
 * place[car] = the car's rank among ALL cars by cumulative time, stable (grid order on equal times) -- the order
 * the overtake step's sorted(cars) at reference :506 would produce. */
static void field_order_sort(const car_t *cars, int n, int *place)
{
    int idx[MCGP_ORACLE_MAX_CARS];
    for (int i = 0; i < n; i++) idx[i] = i;
    stable_sort_by_time(cars, idx, n);
    for (int i = 0; i < n; i++) place[idx[i]] = i;
}
/* two running cars on exactly the same time? (after a VSC's x0.8 the only way the running order can become ambiguous) */
static int running_cars_tie(const car_t *cars, int n)
{
    int idx[MCGP_ORACLE_MAX_CARS];
    const int m = active_sorted(cars, n, idx);
    for (int i = 1; i < m; i++)
        if (cars[idx[i]].cumulative_time == cars[idx[i - 1]].cumulative_time) return 1;
    return 0;
}

/* simulate_race, reference :147-242.  order_out[p] = driver index classified p-th. */
static void simulate_race(const sim_t *s, const uint8_t *grid, uint8_t *order_out,
                          const orc_trace *tr, int64_t sim_index)
{
    const orc_config *cfg = s->cfg;
    const int n = s->n;
    car_t cars[MCGP_ORACLE_MAX_CARS];
    initialize_cars(s, grid, cars);
    simulate_lap_1(s, cars);
    record_trace(s, cars, tr, sim_index, 1);
    int drs_disabled_until = 0;
    int place[MCGP_ORACLE_MAX_CARS];                    /* field order (Philox draw addresses only) */
    field_order_sort(cars, n, place);
    int out_lap[MCGP_ORACLE_MAX_CARS];                  /* by DRIVER: lap of retirement (counter-based back-ends only) */
    const int counter_based = s->rng->mode != MCGP_ORACLE_RNG_MT;
    if (counter_based)
        for (int d = 0; d < n; d++) out_lap[d] = philox_retirement_lap(s->rng, d, s->drv->lap_dnf[d], cfg->total_laps);

    for (int lap = 2; lap <= cfg->total_laps; lap++) {
        /* :168-176, short-circuit chain (Q8) */
        if (draw_event(s->rng, lap, 0) < cfg->red_flag_probability) {
            handle_red_flag(s, cars, lap);
            drs_disabled_until = lap + 2;
        } else if (draw_event(s->rng, lap, 1) < cfg->sc_probability) {
            handle_safety_car(s, cars);
            drs_disabled_until = lap + 2;
        } else if (draw_event(s->rng, lap, 2) < cfg->vsc_probability) {
            handle_vsc(s, cars, lap);
            drs_disabled_until = lap + 1;
            if (running_cars_tie(cars, n)) field_order_sort(cars, n, place);
        }

        /* :179-183 car ahead's last lap, by car (grid index) */
        int idx[MCGP_ORACLE_MAX_CARS];
        double car_ahead_time[MCGP_ORACLE_MAX_CARS];
        const int m = active_sorted(cars, n, idx);
        for (int i = 0; i < n; i++) car_ahead_time[i] = 0.0;            /* .get(driver, 0) */
        for (int i = 1; i < m; i++) car_ahead_time[idx[i]] = cars[idx[i - 1]].last_lap_time;

        /* :186-223 */
        for (int i = 0; i < n; i++) {
            car_t *car = &cars[i];
            if (car->dnf) continue;
            /* :194-197.  MT: the reference's draw.  Counter-based: the lap drawn for this driver before the race. */
            if (counter_based ? out_lap[car->driver] == lap
                              : draw_dnf(s->rng, lap, place[i]) < s->drv->lap_dnf[car->driver]) {
                car->dnf = 1;
                car->lap = lap;
                continue;
            }
            const double clean_air_time = calculate_lap_time(s, car, lap, place[i]);
            double lap_time = clean_air_time;
            if (car->time_behind_leader > 0) {
                const double car_ahead_lap = car_ahead_time[i];
                if (car_ahead_lap > 0 && car->time_behind_leader < cfg->dirty_air_threshold) {
                    const double dirty_air_time = clean_air_time + cfg->dirty_air_penalty;
                    lap_time = (car_ahead_lap > dirty_air_time) ? car_ahead_lap : dirty_air_time;  /* max(a, b) */
                }
            }
            car->cumulative_time += lap_time;
            car->last_lap_time = lap_time;
            car->tire_age += 1;
            car->fuel_load = (car->fuel_load - 1.5 > 0) ? car->fuel_load - 1.5 : 0.0;
            car->lap = lap;
            car->laps_completed += 1;
        }

        handle_pit_stops(s, cars, lap);                                   /* :225 */
        simulate_overtakes(s, cars, lap);                                 /* :226 */
        field_order_sort(cars, n, place);
        update_positions(s, cars, lap, lap <= drs_disabled_until);        /* :227-228 */
        record_trace(s, cars, tr, sim_index, lap);
    }

    /* :230-242 classification */
    int act[MCGP_ORACLE_MAX_CARS], dnf[MCGP_ORACLE_MAX_CARS];
    const int m = active_sorted(cars, n, act);
    int k = 0;
    for (int i = 0; i < n; i++) if (cars[i].dnf) dnf[k++] = i;
    /* sorted(key=(lap, cumulative_time), reverse=True): descending, equal keys keep list order */
    for (int i = 1; i < k; i++) {
        const int x = dnf[i];
        int j = i;
        while (j > 0) {
            const car_t *p = &cars[dnf[j - 1]], *q = &cars[x];
            const int p_less = p->lap < q->lap || (p->lap == q->lap && p->cumulative_time < q->cumulative_time);
            if (!p_less) break;
            dnf[j] = dnf[j - 1];
            j--;
        }
        dnf[j] = x;
    }
    for (int i = 0; i < m; i++) order_out[i] = (uint8_t)cars[act[i]].driver;
    for (int i = 0; i < k; i++) order_out[m + i] = (uint8_t)cars[dnf[i]].driver;
}

static int check_args(const orc_config *cfg, const orc_drivers *drv, int32_t n, int32_t rng, orc_mt_state *mt)
{
    if (!cfg || !drv || n < 1 || n > MCGP_ORACLE_MAX_CARS) return -1;
    if (cfg->total_laps < 1 || cfg->total_laps > 32767) return -1;
    if (rng == MCGP_ORACLE_RNG_MT && !mt) return -1;
    if (rng != MCGP_ORACLE_RNG_MT && rng != MCGP_ORACLE_RNG_PHILOX && rng != MCGP_ORACLE_RNG_PHILOX53) return -1;
    return 0;
}

/* run_monte_carlo, reference :59-100 (histogram as integer counts; the caller divides). */
int orc_run(const orc_config *cfg, const orc_drivers *drv, const double *grid_probs, int32_t n,
            int64_t n_sims, uint64_t sim_offset, uint64_t seed, int32_t rng, orc_mt_state *mt,
            uint64_t *hist, uint8_t *orders, uint8_t *grids, const orc_trace *trace)
{
    if (check_args(cfg, drv, n, rng, mt) || !grid_probs || !hist || n_sims < 0) return -1;
    rng_t r = { rng, mt, { (uint32_t)seed, (uint32_t)(seed >> 32) }, 0 };
    const sim_t s = { cfg, drv, n, &r };
    uint8_t grid[MCGP_ORACLE_MAX_CARS], order[MCGP_ORACLE_MAX_CARS];
    for (int64_t i = 0; i < n_sims; i++) {
        r.sim = sim_offset + (uint64_t)i;
        sample_grid(grid_probs, n, &r, grid);                              /* :85 */
        simulate_race(&s, grid, order, trace, i);                          /* :88 */
        for (int p = 0; p < n; p++) hist[(size_t)order[p] * n + p] += 1;   /* :93-94 */
        if (orders) memcpy(orders + (size_t)i * n, order, (size_t)n);
        if (grids) memcpy(grids + (size_t)i * n, grid, (size_t)n);
    }
    return 0;
}

int orc_simulate_race(const orc_config *cfg, const orc_drivers *drv, const uint8_t *grid, int32_t n,
                      uint64_t sim_id, uint64_t seed, int32_t rng, orc_mt_state *mt, uint8_t *order_out)
{
    if (check_args(cfg, drv, n, rng, mt) || !grid || !order_out) return -1;
    rng_t r = { rng, mt, { (uint32_t)seed, (uint32_t)(seed >> 32) }, sim_id };
    const sim_t s = { cfg, drv, n, &r };
    simulate_race(&s, grid, order_out, NULL, 0);
    return 0;
}

int orc_sample_grid_mt(const double *grid_probs, int32_t n, orc_mt_state *mt, uint8_t *grid_out)
{
    if (!grid_probs || !mt || !grid_out || n < 1 || n > MCGP_ORACLE_MAX_CARS) return -1;
    rng_t r = { MCGP_ORACLE_RNG_MT, mt, { 0, 0 }, 0 };
    sample_grid(grid_probs, n, &r, grid_out);
    return 0;
}

/* ---- grid-probability front end (reference src/elo.py:124-141, src/predictor.py:321-407) ----
 * CPU side of the checker for the device front end: the same header text as the HIP kernel compiles. */
#include "frontend_exp.h"

double orc_fe_exp(double x) { return mcgp_fe_exp(x); }

int orc_grid_probs(const double *rating, const double *teammate_delta, const double *form_score,
                   const double *circuit_affinity, const int32_t *penalty, int32_t n, double *out)
{
    if (!rating || !teammate_delta || !form_score || !circuit_affinity || !penalty || !out) return -1;
    if (n < 1 || n > MCGP_ORACLE_MAX_CARS) return -1;
    double p[MCGP_ORACLE_MAX_CARS], tmp[MCGP_ORACLE_MAX_CARS];
    mcgp_fe_pole_probs(rating, teammate_delta, n, p);
    for (int d = 0; d < n; d++)
        mcgp_fe_grid_row(p[d], form_score[d], circuit_affinity[d], penalty[d], n, out + (size_t)d * n, tmp);
    return 0;
}

/* ---- a season of Elo updates (reference src/elo.py:40-122) ----
 * CPU side of the checker for mcgp_elo_season: the same header text as the HIP kernel compiles (elo_update.h), the
 * events applied one after the other, every delta of an event computed before any is applied. */
#include "elo_update.h"

double orc_elo_pow10(double x) { return mcgp_elo_pow10(x); }

int orc_elo_season(int32_t n, int32_t n_events, const int32_t *kind, const double *k, const uint32_t *count,
                   const uint8_t *who, const double *value, double *ratings, double *after_out)
{
    if (n < 1 || n > MCGP_ORACLE_MAX_CARS || n_events < 0 || !ratings) return -1;
    if (n_events > 0 && (!kind || !k || !count || !who || !value)) return -1;
    for (int e = 0; e < n_events; e++) {
        const int m = (int)count[e];
        if (m > n || (kind[e] != 0 && kind[e] != 1)) return -1;
        double *row = ratings + (kind[e] ? n : 0);
        const uint8_t *w = who + (size_t)e * n;
        const double *v = value + (size_t)e * n;
        if (m >= 2) {                                              /* elo.py:54-56, :93-94 */
            double delta[MCGP_ORACLE_MAX_CARS];
            for (int a = 0; a < m; a++) delta[a] = mcgp_elo_delta(row, w, v, m, a, k[e]);
            for (int a = 0; a < m; a++) row[w[a]] = row[w[a]] + delta[a];             /* :80-83, :119-122 */
        }
        if (after_out)
            for (int i = 0; i < 2 * n; i++) after_out[(size_t)e * 2 * n + i] = ratings[i];
    }
    return 0;
}
