/* frontend_exp.h -- exp() for the grid-probability front end, written out in binary64 +, -, * only.
 *
 * The reference calls numpy's exp (reference src/elo.py:139, src/predictor.py:365), whose bits depend on the
 * libm of the day; a GPU has no libm at all.  The front end therefore defines its own exp, identical to the last
 * bit wherever this header is compiled (gcc for the CPU oracle, hipcc for gfx950; both with -ffp-contract=off):
 * Cody-Waite reduction x = k ln2 + r, |r| <= ln2 / 2, degree-14 Taylor polynomial by Horner, scaling by 2^k
 * through the exponent field.  |relative error| < 4e-16 on the front end's argument range [-40, 0] (checked
 * against libm in tests/test_frontend.py); the matrices it produces agree with the reference's to ~1e-15.
 * The same text lives in oracle/frontend_exp.h (a test keeps the two copies identical). */
#ifndef MCGP_FRONTEND_EXP_H
#define MCGP_FRONTEND_EXP_H
#include <stdint.h>
#include <string.h>

#ifndef MCGP_FE_FN
#define MCGP_FE_FN static inline
#endif

MCGP_FE_FN double mcgp_fe_pow2(int k)          /* 2^k for -1022 <= k <= 1023 */
{
    const uint64_t bits = (uint64_t)(k + 1023) << 52;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

MCGP_FE_FN double mcgp_fe_exp(double x)
{
    if (x != x) return x;
    if (x > 709.0) return mcgp_fe_pow2(1023) * 2.0;                 /* +inf */
    if (x < -745.0) return 0.0;
    const double kf = x * 1.44269504088896338700e+00;               /* x / ln 2 */
    const int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5));
    const double kd = (double)k;
    const double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    /* 1 / j!, j = 14 .. 0 */
    double p = 1.1470745597729725e-11;
    p = p * r + 1.6059043836821613e-10;
    p = p * r + 2.0876756987868100e-09;
    p = p * r + 2.5052108385441720e-08;
    p = p * r + 2.7557319223985893e-07;
    p = p * r + 2.7557319223985888e-06;
    p = p * r + 2.4801587301587302e-05;
    p = p * r + 1.9841269841269841e-04;
    p = p * r + 1.3888888888888889e-03;
    p = p * r + 8.3333333333333332e-03;
    p = p * r + 4.1666666666666664e-02;
    p = p * r + 1.6666666666666666e-01;
    p = p * r + 5.0000000000000000e-01;
    p = p * r + 1.0;
    p = p * r + 1.0;
    if (k >= -1021) return p * mcgp_fe_pow2(k);
    return (p * mcgp_fe_pow2(k + 1000)) * mcgp_fe_pow2(-1000);      /* gradual underflow */
}

/* Grid-slot distribution of every driver (row-major [driver][slot]) from the quali ratings and features:
 * reference F1EloSystem.predict_quali_probs (src/elo.py:124-141), F1Predictor._predict_quali
 * (src/predictor.py:321-375) and _adjust_for_penalties (:377-407), in the reference's evaluation order.
 * `row` says which driver's row to produce (the HIP kernel runs one thread per row); scratch `p` holds n doubles. */
MCGP_FE_FN void mcgp_fe_pole_probs(const double *rating, const double *teammate_delta, int n, double *p)
{
    /* softmax of rating / 100 with max subtraction (elo.py:132-141) */
    double mx = rating[0] / 100;
    for (int d = 1; d < n; ++d) {
        const double s = rating[d] / 100;
        if (s > mx) mx = s;
    }
    double total = 0.0;
    for (int d = 0; d < n; ++d) {
        p[d] = mcgp_fe_exp(rating[d] / 100 - mx);
        total = total + p[d];
    }
    for (int d = 0; d < n; ++d) p[d] = total > 0 ? p[d] / total : 1.0 / n;
    /* teammate adjustment and renormalisation (predictor.py:334-343) */
    for (int d = 0; d < n; ++d) {
        const double td = teammate_delta[d];
        if (td != 0) {
            double boost = 1 + (td * 0.25);
            if (boost > 1.5) boost = 1.5;
            if (boost < 0.5) boost = 0.5;
            p[d] = p[d] * boost;
        }
    }
    total = 0.0;
    for (int d = 0; d < n; ++d) total = total + p[d];
    if (total > 0)
        for (int d = 0; d < n; ++d) p[d] = p[d] / total;
}

MCGP_FE_FN void mcgp_fe_grid_row(double pole_prob, double form_score, double circuit_affinity, int penalty, int n,
                                 double *row /* n */, double *tmp /* n */)
{
    /* predictor.py:349-374 */
    double adj = pole_prob * (1 + form_score * 0.15 + circuit_affinity * 0.10);
    if (adj > 0.999) adj = 0.999;
    if (adj < 0.001) adj = 0.001;
    double sigma = (double)n / 4;
    if (sigma < 1.0) sigma = 1.0;
    const double expected = (1 - adj) * n;
    const double two_s2 = 2 * (sigma * sigma);
    double total = 0.0;
    for (int pos = 0; pos < n; ++pos) {
        const double dx = (double)pos - expected;
        tmp[pos] = mcgp_fe_exp(-(dx * dx) / two_s2);
        total = total + tmp[pos];
    }
    for (int pos = 0; pos < n; ++pos) tmp[pos] = total > 0 ? tmp[pos] / total : 1.0 / n;
    /* _adjust_for_penalties, predictor.py:391-405 */
    if (penalty > 0) {
        for (int pos = 0; pos < n; ++pos) row[pos] = 0.0;
        if (penalty >= n) {
            row[n - 1] = 1.0;
        } else {
            for (int i = 0; i < n; ++i) {
                const int np = i + penalty < n - 1 ? i + penalty : n - 1;
                row[np] = row[np] + tmp[i];
            }
        }
    } else {
        for (int pos = 0; pos < n; ++pos) row[pos] = tmp[pos];
    }
}

#endif
