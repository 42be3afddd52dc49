/* elo_update.h -- the all-pairs Elo update of one qualifying session or race, written out in binary64 +, -, *, /.
 *
 * Reference: F1EloSystem.expected_score (src/elo.py:40-43), update_quali_ratings (:45-83), update_race_ratings
 * (:85-122): every participant's delta is computed from the ratings BEFORE the event -- the sum over the other
 * participants, in list order, of  k * (actual - expected) / (m - 1)  -- and all deltas are applied afterwards.
 * The one transcendental is Python's `10 ** exponent` (libm pow); like the front end's exp (frontend_exp.h) it is
 * defined here so that the CPU oracle (gcc) and the device kernel (hipcc, both -ffp-contract=off) agree to the last
 * bit: 10^x = exp(x ln 10) with the product x ln 10 carried in two doubles (Veltkamp split, Dekker product) and the
 * low part applied as a first-order correction.  |relative error| < 1e-15 on the clamped range [-10, 10] (checked
 * against libm in tests/test_elo_season.py); a season of updates stays within a few ulp of the reference's ratings.
 * The same text lives in oracle/elo_update.h (a test keeps the two copies identical). */
#ifndef MCGP_ELO_UPDATE_H
#define MCGP_ELO_UPDATE_H
#include <stdint.h>

#include "frontend_exp.h"

MCGP_FE_FN double mcgp_elo_pow10(double x)             /* 10^x, |x| <= 10 (the reference clamps the exponent) */
{
    /* ln 10 = L + LT,  L = LH + LL exactly with LH holding the upper 26 bits of L's significand */
    const double L = 2.30258509299404590109e+00, LT = -2.17075622338224935e-16;
    const double LH = 2.3025850653648376e+00, LL = L - LH;
    const double c = 134217729.0 * x;                  /* 2^27 + 1: Veltkamp split of x */
    const double xh = c - (c - x), xl = x - xh;
    const double t = x * L;
    const double e = ((xh * LH - t) + xh * LL + xl * LH) + xl * LL;      /* x L = t + e exactly (Dekker) */
    const double lo = e + x * LT;
    const double y = mcgp_fe_exp(t);
    return y + y * lo;
}

MCGP_FE_FN double mcgp_elo_expected(double rating_a, double rating_b)  /* elo.py:40-43 */
{
    double exponent = (rating_b - rating_a) / 400;
    if (exponent > 10) exponent = 10;
    if (exponent < -10) exponent = -10;
    return 1 / (1 + mcgp_elo_pow10(exponent));
}

/* One term of a participant's delta: the entry (rating_a, value_a) against the entry (rating_b, value_b) of an event
 * with m entries; value = lap time (qualifying) or finishing position (race) -- lower wins, equal ties (elo.py:66-77,
 * :106-116). */
MCGP_FE_FN double mcgp_elo_term(double rating_a, double value_a, double rating_b, double value_b, double k, int m)
{
    const double expected = mcgp_elo_expected(rating_a, rating_b);
    const double actual = value_a < value_b ? 1.0 : value_a > value_b ? 0.0 : 0.5;
    return k * (actual - expected) / (m - 1);
}

/* Delta of entry `a`: its terms against the other entries, added up in list order (elo.py:63-78).  who[j] = driver
 * index of the j-th entry of the result list, value[j] its lap time / position; rating[] is indexed by driver and
 * holds the ratings before the event. */
MCGP_FE_FN double mcgp_elo_delta(const double *rating, const uint8_t *who, const double *value, int m, int a, double k)
{
    double delta = 0.0;
    for (int b = 0; b < m; ++b) {
        if (b == a) continue;
        delta = delta + mcgp_elo_term(rating[who[a]], value[a], rating[who[b]], value[b], k, m);
    }
    return delta;
}

#endif
