// race_kernel_reg.hip.h -- register-resident whole-race kernel, field size N fixed at compile time.
//
// Same algorithm and same results as race_kernel (race_kernel.hip.h), laid out for the
// CDNA4 register file instead of LDS: one lane = one simulation, and the N cars of that
// simulation live in VGPR arrays indexed by TIME RANK over all cars
//
//     cum[r]  f64   cumulative time of the car at rank r      (reference CarState.cumulative_time)
//     pk[r]   u32   grid slot | driver | dirty | drs | dnf | used compounds | compound | tyre age
//
// Every loop over cars is fully unrolled, so all register indices are compile-time
// constants; "the car ahead" is simply rank r-1.  Ordering work is a fixed compare-exchange
// network (Knuth's merge exchange, 97 comparators for N = 20) on (cum, pk) after the lap
// times have been added, and odd-even transposition rounds after an overtake pass (which
// only perturbs the order locally).  Ties compare pk as an integer: the grid slot sits in
// its top bits, which is Python's stable-sort order for the reference's grid-ordered lists.
//
// LDS holds only what must be addressed by DRIVER index per lane (conflict-free, row stride
// = blockDim.x): last lap time [driver] f64, this lap's deviates [driver] f32 (NaN = the DNF
// draw hit), later re-used for the overtake draw words; plus the block-shared tables.
// 12 N + 16 bytes per simulation instead of 21 N: 8 waves per CU instead of 5, and no LDS
// round trip on the hot per-car state.
//
// Random draws: one Philox block per TWO drivers and lap (words: dnf, noise, dnf, noise),
// generated in a driver-indexed pre-pass with wave-uniform thresholds, so the per-lap RNG
// cost is N/2 + 1 Philox calls plus one per four overtake attempts.
#pragma once
#include "race_common.hip.h"
#include "race_isa.hip.h"

// Diagnostic builds only (tools/ablate.sh): bit k set = run section k twice (each section is
// idempotent, results unchanged) so its cost shows up as a time difference.  0 in the product build.
#ifndef MCGP_DUP
#define MCGP_DUP 0
#endif
// Diagnostic: bit k set = leave section k out (results become wrong; timing only).
#ifndef MCGP_SKIP
#define MCGP_SKIP 0
#endif


#include <utility>

namespace mcgp {

// pk word of the register kernel
constexpr uint32_t k2AgeMask = 0x7FFu;         // tyre age; lap of retirement once dnf is set
constexpr int k2CompShift = 11;                // 3 bits
constexpr int k2UsedShift = 14;                // 5 bits
constexpr uint32_t k2Dnf = 1u << 19;
constexpr uint32_t k2Drs = 1u << 20;
constexpr uint32_t k2Dirty = 1u << 21;
constexpr int k2IdShift = 22;                  // 5 bits, driver index
constexpr int k2GposShift = 27;                // 5 bits, grid slot (most significant: tie-break)

__host__ __device__ constexpr size_t per_thread_lds_bytes_reg(int n) { return (size_t)n * 12 + 16; }
// block-shared tables of the register kernel: normal table, per-driver / per-compound constants,
// n x n histogram (u32) and the transposed grid-probability matrix [slot][driver] (f64)
__host__ __device__ constexpr size_t shared_lds_bytes_reg(int n)
{
    return (size_t)kNormalRows * 16 + 4 * kMaxCars * 8 + kMaxCars * 8 + 2 * kCompStride * 8 +
           kMaxCars * kCompStride * 2 + (((size_t)n * n * 4 + 15) / 16) * 16 + (size_t)n * n * 8;
}

// Knuth, TAOCP 5.2.2 Algorithm M (merge exchange): a sorting network for any N.
template <int N>
struct MergeExchange {
    int a[N * 8];
    int b[N * 8];
    int n;
    constexpr MergeExchange() : a{}, b{}, n(0)
    {
        int t = 0;
        while ((1 << t) < N) ++t;
        for (int p = t > 0 ? 1 << (t - 1) : 0; p > 0; p >>= 1) {
            int q = 1 << (t - 1), r = 0, d = p;
            while (true) {
                for (int i = 0; i < N - d; ++i)
                    if ((i & p) == r) { a[n] = i; b[n] = i + d; ++n; }
                if (q == p) break;
                d = q - p;
                q >>= 1;
                r = p;
            }
        }
    }
};

// compare-exchange on (cum, pk): after it, slot A sorts before slot B.
__device__ __forceinline__ bool cmpx(double &ca, uint32_t &pa, double &cb, uint32_t &pb)
{
    const bool sw = (ca > cb) || (ca == cb && pa > pb);
    double c0, c1;
    minmax_f64(ca, cb, c0, c1);
    const uint32_t p0 = sw ? pb : pa, p1 = sw ? pa : pb;
    ca = c0; cb = c1; pa = p0; pb = p1;
    return sw;
}

// compare-exchange on the time alone (ties left as they are): the network's comparator.
__device__ __forceinline__ void cmpx_time(double &ca, uint32_t &pa, double &cb, uint32_t &pb)
{
    const bool sw = ca > cb;
    double c0, c1;
    minmax_f64(ca, cb, c0, c1);
    const uint32_t p0 = sw ? pb : pa, p1 = sw ? pa : pb;
    ca = c0; cb = c1; pa = p0; pb = p1;
}

template <int N, size_t... I>
__device__ __forceinline__ void network_sort_impl(double (&cum)[N], uint32_t (&pk)[N], std::index_sequence<I...>)
{
    constexpr MergeExchange<N> net{};
    (cmpx_time(cum[net.a[I]], pk[net.a[I]], cum[net.b[I]], pk[net.b[I]]), ...);
}

// Is the field in (cumulative_time, grid slot) order?
template <int N>
__device__ __forceinline__ bool in_order(const double (&cum)[N], const uint32_t (&pk)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; ++i)
        bad |= (cum[i] > cum[i + 1]) || (cum[i] == cum[i + 1] && pk[i] > pk[i + 1]);
    return !bad;
}

// After a round that ends with the odd pairs (1,2), (3,4), .. those pairs are in order by
// construction; the field is in order iff the even pairs (0,1), (2,3), .. still are.
template <int N>
__device__ __forceinline__ bool even_pairs_in_order(const double (&cum)[N], const uint32_t (&pk)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2)
        bad |= (cum[i] > cum[i + 1]) || (cum[i] == cum[i + 1] && pk[i] > pk[i + 1]);
    return !bad;
}

// Odd-even transposition rounds until the field is in order: the re-sort after a LOCAL
// perturbation (an overtake pass moves a few cars by 0.1-0.3 s).
template <int N>
__device__ __forceinline__ void transposition_sort(double (&cum)[N], uint32_t (&pk)[N])
{
    do {
#pragma unroll
        for (int i = 0; i + 1 < N; i += 2) (void)cmpx(cum[i], pk[i], cum[i + 1], pk[i + 1]);
#pragma unroll
        for (int i = 1; i + 1 < N; i += 2) (void)cmpx(cum[i], pk[i], cum[i + 1], pk[i + 1]);
    } while (!even_pairs_in_order<N>(cum, pk));
}

// After the time-only network the field is ordered by time; only equal times can still be in the
// wrong (grid) order.
template <int N>
__device__ __forceinline__ bool ties_in_order(const double (&cum)[N], const uint32_t (&pk)[N])
{
    bool bad = false;
#pragma unroll
    for (int i = 0; i + 1 < N; ++i) bad |= (cum[i] == cum[i + 1]) && pk[i] > pk[i + 1];
    return !bad;
}

// Full sort by (cumulative_time, grid slot): Python's stable sorted() of the reference (:506 etc.).
// The network orders by time; equal times (structural: lap-1 retirements at 0.0) are then put in
// grid order by the transposition rounds, which almost never have anything to do.
template <int N>
__device__ __forceinline__ void network_sort(double (&cum)[N], uint32_t (&pk)[N])
{
    constexpr MergeExchange<N> net{};
    network_sort_impl<N>(cum, pk, std::make_index_sequence<(size_t)net.n>{});
    if (!ties_in_order<N>(cum, pk)) transposition_sort<N>(cum, pk);
}

// _update_positions, reference :538-560.
template <int N>
__device__ __forceinline__ void update_positions_reg(const double (&cum)[N], uint32_t (&pk)[N],
                                                     bool drs_allowed, double dirty_thr)
{
    bool first = true;
    double leader = 0.0, prev = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint32_t p = pk[i];
        if (!(p & k2Dnf)) {
            const double t = cum[i];
            if (first) leader = t;
            const double tbl = t - leader;
            p &= ~(k2Drs | k2Dirty);
            if (tbl > 0 && tbl < dirty_thr) p |= k2Dirty;
            if (!first && drs_allowed && (t - prev) < 1.0) p |= k2Drs;
            pk[i] = p;
            prev = t;
            first = false;
        }
    }
}

#ifndef MCGP_MIN_WAVES
#define MCGP_MIN_WAVES 2
#endif
#ifndef MCGP_PREPASS_BLOCKS
#define MCGP_PREPASS_BLOCKS 2
#endif
template <int N>
__global__ void __launch_bounds__(512, MCGP_MIN_WAVES)
race_kernel_reg(const KParams *__restrict__ P, uint64_t n_sims, uint64_t sim_offset,
                uint32_t seed_lo, uint32_t seed_hi, unsigned long long *__restrict__ hist,
                uint8_t *__restrict__ orders, const uint8_t *__restrict__ fixed_grid, uint32_t n_batches)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int B = blockDim.x;
    const int L = P->total_laps;
    const int track = P->track;

    // ---- LDS carve-up: block-shared tables, then the per-lane rows ----
    float4 *t_norm = reinterpret_cast<float4 *>(smem);
    double *t_base = reinterpret_cast<double *>(smem + kNormalRows * 16);
    double *t_factor = t_base + kMaxCars;
    double *t_deg = t_factor + kMaxCars;
    double *t_var = t_deg + kMaxCars;
    unsigned long long *t_dnf = reinterpret_cast<unsigned long long *>(t_var + kMaxCars);
    double *t_cdeg = reinterpret_cast<double *>(t_dnf + kMaxCars);
    double *t_cdelta = t_cdeg + kCompStride;
    uint16_t *t_opt = reinterpret_cast<uint16_t *>(t_cdelta + kCompStride);
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(t_opt + kMaxCars * kCompStride);
    double *t_grid = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(s_hist) + ((N * N * 4 + 15) / 16) * 16);  // [slot][driver]
    double *s_last = t_grid + N * N;                                               // [driver][lane]
    uint32_t *s_word = reinterpret_cast<uint32_t *>(s_last + (size_t)N * B);       // [N + 4][lane]
    float *s_z = reinterpret_cast<float *>(s_word);

#define LAST(d) s_last[(d) * B + tid]
#define ZED(d) s_z[(d) * B + tid]
#define WORD(k) s_word[(k) * B + tid]

    for (int i = tid; i < kNormalRows * 4; i += B)
        reinterpret_cast<uint32_t *>(t_norm)[i] = P->normal_bits[i];
    for (int i = tid; i < kMaxCars; i += B) {
        t_base[i] = P->base_pace[i];
        t_factor[i] = P->factor[i];
        t_deg[i] = P->tire_deg[i];
        t_var[i] = P->variance[i];
        t_dnf[i] = P->t_dnf[i];
    }
    for (int i = tid; i < kCompStride; i += B) {
        t_cdeg[i] = P->comp_deg[i];
        t_cdelta[i] = P->comp_delta[i];
    }
    for (int i = tid; i < kMaxCars * kCompStride; i += B) t_opt[i] = P->opt_laps[i];
    for (int i = tid; i < N * N; i += B) {
        s_hist[i] = 0u;
        t_grid[(i % N) * N + (i / N)] = P->grid_probs[i];       // transposed: lanes gather by driver, conflict-free
    }
    __syncthreads();

    const double pit_loss = P->pit_loss;
    const double overtake_delta = P->overtake_delta;
    const double drs_delta = P->drs_delta;
    const double dirty_thr = P->dirty_thr;
    const double dirty_pen = P->dirty_pen;
    const float kNaN = __uint_as_float(0x7fc00000u);
    const uint32_t pop_sh = (uint32_t)P->pop_sh, pop_mh = (uint32_t)P->pop_mh;
    const uint64_t t_red = P->t_red, t_sc = P->t_sc, t_vsc = P->t_vsc, t_vsc_tire = P->t_vsc_tire;

    // Everything the lap step of one slot reads from LDS, fetched in one batch.
    struct SlotIn {
        float z;
        uint32_t opt;
        double last, base, factor, var, cdeg, cdelta;
    };
    auto load_slot = [&](uint32_t p) -> SlotIn {
        const uint32_t id = (p >> k2IdShift) & 31u;
        const uint32_t comp = (p >> k2CompShift) & 7u;
        SlotIn r;
        r.z = ZED(id);
        r.last = LAST(id);
        r.base = t_base[id];
        r.factor = t_factor[id];
        r.var = t_var[id];
        r.cdeg = t_cdeg[comp];
        r.cdelta = t_cdelta[comp];
        r.opt = t_opt[id * kCompStride + comp];
        return r;
    };

    for (uint32_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const uint64_t local = (uint64_t)batch * (uint64_t)B + (uint64_t)tid;
        if (local >= n_sims) continue;
        const uint64_t sim = sim_offset + local;
        const uint32_t c0 = (uint32_t)sim, c1 = (uint32_t)(sim >> 32);

        double cum[N];
        uint32_t pk[N];

        // ================= _sample_grid, reference :102-145 (probs scratch in the LAST rows) =================
        {
            uint32_t remaining = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
            int n_remaining = N;
            uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
#pragma unroll 1
            for (int pos = 0; pos < N; ++pos) {
                uint32_t sel;
                if (fixed_grid) {
                    sel = fixed_grid[pos];
                } else if (MCGP_SKIP & 4) {
                    sel = (uint32_t)((pos * 7 + (int)(c0 & 3u)) % N);
                    while (!((remaining >> sel) & 1u)) sel = (sel + 1u) % (uint32_t)N;
                } else {
                    if ((pos & 3) == 0)
                        philox4x32_10(c0, c1, 0u, kPurposeGrid | (uint32_t)(pos >> 2), seed_lo, seed_hi, g0, g1, g2, g3);
                    const uint32_t gw = (pos & 3) == 0 ? g0 : (pos & 3) == 1 ? g1 : (pos & 3) == 2 ? g2 : g3;
                    const double u = u32_to_unit(gw);
                    // Only the drivers still unplaced contribute (zeros add nothing to the sums and repeat
                    // the running cdf), so each pass walks the `remaining` bit set: N - pos steps on every lane.
                    const double *gcol = t_grid + pos * N;
                    double total = 0.0;                                   // :119-123
                    for (uint32_t m = remaining; m; m &= m - 1u) total = total + gcol[__ffs((int)m) - 1];
                    double prob_sum = 0.0;                                // :125-133
                    for (uint32_t m = remaining; m; m &= m - 1u) {
                        const int d = __ffs((int)m) - 1;
                        const double p = total > 0 ? gcol[d] / total : 1.0 / (double)n_remaining;
                        LAST(d) = p;
                        prob_sum = prob_sum + p;
                    }
                    const bool renorm = prob_sum > 0 && fabs(prob_sum - 1.0) > 1e-9;   // :134-135
                    // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right')
                    double acc = 0.0;
                    for (uint32_t m = remaining; m; m &= m - 1u) {
                        const int d = __ffs((int)m) - 1;
                        double p = LAST(d);
                        if (renorm) p = p / prob_sum;
                        acc = acc + p;
                        LAST(d) = acc;
                    }
                    const double cdf_last = acc;
                    sel = (uint32_t)N;
                    for (uint32_t m = remaining; m; m &= m - 1u) {        // first driver whose cdf exceeds u
                        const int d = __ffs((int)m) - 1;
                        if (sel == (uint32_t)N && !(LAST(d) / cdf_last <= u)) sel = (uint32_t)d;
                    }
                    if (sel >= (uint32_t)N) sel = 31u - (uint32_t)__clz((int)remaining);   // unreachable: cdf[-1] == 1 > u
                }
                if ((remaining >> sel) & 1u) { remaining &= ~(1u << sel); --n_remaining; }
                WORD(pos) = sel;
            }
        }

        // ================= _initialize_cars, reference :244-273 (slot i = grid position i) =================
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t id = WORD(i);
            uint32_t comp, age;
            if (track == 2) { comp = 4u; age = 0u; }
            else if (track == 1) { comp = 3u; age = 0u; }
            else { comp = i < 10 ? 0u : 1u; age = i < 10 ? 4u : 0u; }
            pk[i] = age | (comp << k2CompShift) | ((1u << comp) << k2UsedShift) | (id << k2IdShift) |
                    ((uint32_t)i << k2GposShift);
            cum[i] = 0.0;
        }

        // ================= _simulate_lap_1, reference :275-311 =================
        // draws by driver (wave-uniform thresholds): ZED = lap-noise deviate or NaN (retired), LAST = start deviate
#pragma unroll 1
        for (int d = 0; d < N; ++d) {
            uint32_t w0, w1, w2, w3;
            philox4x32_10(c0, c1, 1u, kPurposeCar | (uint32_t)d, seed_lo, seed_hi, w0, w1, w2, w3);
            const bool out = (uint64_t)w0 < P->t_dnf1[d];
            ZED(d) = out ? kNaN : normal_from_u32(w1, t_norm);
            LAST(d) = (double)normal_from_u32(w2, t_norm);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t p = pk[i];
            const uint32_t id = (p >> k2IdShift) & 31u;
            const float z = ZED(id);
            const double zs = LAST(id);
            if (z != z) {
                pk[i] = (p & ~k2AgeMask) | k2Dnf | 1u;
            } else {
                const uint32_t comp = (p >> k2CompShift) & 7u;
                const uint32_t age = p & k2AgeMask;
                const double eff = t_cdeg[comp] * t_factor[id];
                const double tire = (double)age * eff;
                const double fuel_effect = (110.0 - 110.0) * 0.03;
                const double noise = 0.0 + t_var[id] * (double)z;
                const double base_lap = t_base[id] + tire - fuel_effect + t_cdelta[comp] - 0.0 + noise;
                double pf = 0.5 + (double)(i + 1) * 0.1;
                if (!(pf < 1.5)) pf = 1.5;
                double sd = 0.0 + pf * zs;
                if (i + 1 <= 3 && 1.0 < sd) sd = 1.0;
                cum[i] = 0.0 + (base_lap - sd * 0.5);
                pk[i] = (p & ~k2AgeMask) | (age + 1u);
            }
        }
#pragma unroll 1
        for (int d = 0; d < N; ++d) LAST(d) = 0.0;          // last_lap_time is not set on lap 1 (Q3)
        network_sort<N>(cum, pk);
        update_positions_reg<N>(cum, pk, false, dirty_thr);

        // ================= laps 2..L, reference :166-228 =================
        int drs_disabled_until = 0;
#pragma unroll 1
        for (int lap = 2; lap <= ((MCGP_SKIP & 8) ? 1 : L); ++lap) {
            const int remaining_laps = L - lap;
            // ---- race-interrupting events, :168-176 ----
            {
                uint32_t e0, e1, e2, e3;
                philox4x32_10(c0, c1, (uint32_t)lap, kPurposeEvent, seed_lo, seed_hi, e0, e1, e2, e3);
                if (MCGP_DUP & 16) {
                    uint32_t f0, f1, f2, f3;
                    philox4x32_10(c0 ^ e0, c1, (uint32_t)lap, kPurposeEvent, seed_lo, seed_hi, f0, f1, f2, f3);
                    if ((f0 | f1 | f2 | f3) == 0u) e0 = f0;     // never true in practice; keeps the block alive
                }
                const bool red = (uint64_t)e0 < t_red;
                const bool sc = !red && (uint64_t)e1 < t_sc;
                const bool vsc = !red && !sc && (uint64_t)e2 < t_vsc;
                if (!(MCGP_SKIP & 2) && (red || sc || vsc)) {
                    const bool dec_age = sc || (vsc && (uint64_t)e3 < t_vsc_tire);
                    const uint32_t newc = stint_compound(track, remaining_laps);
                    int k = 0;
                    double leader = 0.0, prev_nt = -1.0;
                    bool tie = false;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        uint32_t p = pk[i];
                        if (!(p & k2Dnf)) {
                            const double t = cum[i];
                            if (k == 0) leader = t;
                            double nt;
                            if (red) nt = leader + (double)k * 0.1;
                            else if (sc) nt = leader + (double)k * 0.5;
                            else { const double gap = t - leader; nt = leader + gap * 0.8; }
                            tie |= nt == prev_nt;
                            prev_nt = nt;
                            const double tbl = nt - leader;
                            p &= ~k2Dirty;
                            if (tbl > 0 && tbl < dirty_thr) p |= k2Dirty;
                            uint32_t age = p & k2AgeMask;
                            if (red) {
                                age = 0u;
                                p = (p & ~(7u << k2CompShift)) | (newc << k2CompShift) | ((1u << newc) << k2UsedShift);
                            } else if (dec_age) {
                                age = age > 0u ? age - 1u : 0u;
                            }
                            cum[i] = nt;
                            pk[i] = (p & ~k2AgeMask) | age;
                            ++k;
                        }
                    }
                    drs_disabled_until = lap + (vsc ? 1 : 2);
                    // Re-spacing keeps the running cars' relative order (the only order the lap loop below
                    // needs; the full order is rebuilt after it).  x0.8 is monotone but may round two gaps
                    // together, and equal times must fall back to grid order: re-sort only then.
                    if (tie) transposition_sort<N>(cum, pk);
                }
            }

            // ---- this lap's draws, by driver pair: one Philox block = (dnf, noise) x 2 ----
            // MCGP_PREPASS_BLOCKS blocks per iteration: their Philox chains interleave, and the table
            // look-ups of the inverse-normal transform go out as one batch.
#pragma unroll 1
            for (int rep = 0; rep < ((MCGP_DUP & 2) ? 2 : 1); ++rep)
#pragma unroll 1
            for (int b0 = 0; b0 < (N + 1) / 2; b0 += MCGP_PREPASS_BLOCKS) {
                uint32_t w[MCGP_PREPASS_BLOCKS][4];
#pragma unroll
                for (int j = 0; j < MCGP_PREPASS_BLOCKS; ++j) {
                    w[j][0] = w[j][1] = w[j][2] = w[j][3] = 0u;
                    if (b0 + j < (N + 1) / 2)
                        philox4x32_10(c0, c1, (uint32_t)lap, kPurposeCar | (uint32_t)(b0 + j), seed_lo, seed_hi,
                                      w[j][0], w[j][1], w[j][2], w[j][3]);
                }
                float z[MCGP_PREPASS_BLOCKS][2];
                unsigned long long q[MCGP_PREPASS_BLOCKS][2];
#pragma unroll
                for (int j = 0; j < MCGP_PREPASS_BLOCKS; ++j) {
                    const int d0 = 2 * (b0 + j), d1 = d0 + 1;
                    z[j][0] = normal_from_u32(w[j][1], t_norm);
                    z[j][1] = normal_from_u32(w[j][3], t_norm);
                    q[j][0] = t_dnf[d0 < N ? d0 : 0];
                    q[j][1] = t_dnf[d1 < N ? d1 : 0];
                }
#pragma unroll
                for (int j = 0; j < MCGP_PREPASS_BLOCKS; ++j) {
                    const int d0 = 2 * (b0 + j), d1 = d0 + 1;
                    if (d0 < N) ZED(d0) = ((uint64_t)w[j][0] < q[j][0]) ? kNaN : z[j][0];
                    if (d1 < N) ZED(d1) = ((uint64_t)w[j][2] < q[j][1]) ? kNaN : z[j][1];
                }
            }

            // ---- every running car's lap (:179-223) with its pit stop (:433-494), in time-rank order ----
            // All LDS gathers of a slot (deviate, last lap, per-driver and per-compound constants) are issued
            // in one batch, one slot ahead of their use, so the wave does not park on s_waitcnt per access.
            {
                double fuel = 110.0 - 1.5 * (double)(lap - 1);
                if (!(fuel > 0)) fuel = 0.0;
                const double fuel_effect = (110.0 - fuel) * 0.03;
                double carry = 0.0;
                SlotIn cur = load_slot(pk[0]);
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    SlotIn nxt = cur;
                    if (i + 1 < N) nxt = load_slot(pk[i + 1]);
                    uint32_t p = pk[i];
                    if (!(p & k2Dnf)) {
                        const uint32_t id = (p >> k2IdShift) & 31u;
                        const double ahead_last = carry;
                        carry = cur.last;
                        const float z = cur.z;
                        if (z != z) {
                            pk[i] = (p & ~k2AgeMask) | k2Dnf | (uint32_t)lap;
                        } else {
                            uint32_t comp = (p >> k2CompShift) & 7u;
                            uint32_t age = p & k2AgeMask;
                            const double eff = cur.cdeg * cur.factor;
                            const double tire = (double)age * eff;
                            const double drs_gain = (p & k2Drs) ? drs_delta : 0.0;
                            const double noise = 0.0 + cur.var * (double)z;
                            const double clean = cur.base + tire - fuel_effect + cur.cdelta - drs_gain + noise;
                            double lap_time = clean;
                            if ((p & k2Dirty) && ahead_last > 0) {
                                const double dirty_time = clean + dirty_pen;
                                lap_time = ahead_last > dirty_time ? ahead_last : dirty_time;
                            }
                            double t = cum[i] + lap_time;
                            age += 1u;
                            if ((int)age > (int)cur.opt && remaining_laps > 5) {
                                t = t + pit_loss;
                                uint32_t newc = stint_compound(track, remaining_laps);
                                const uint32_t used_dry = (p >> k2UsedShift) & 7u;
                                if (track == 0 && __popc(used_dry) == 1 && ((used_dry >> newc) & 1u)) {
                                    const uint32_t avail = 7u & ~used_dry;
                                    const uint32_t popped = avail == 5u ? pop_sh : avail == 6u ? pop_mh : 0u;
                                    if (remaining_laps > 20) newc = (avail & 2u) ? 1u : popped;
                                    else newc = (avail & 1u) ? 0u : popped;
                                }
                                comp = newc;
                                p = (p & ~(7u << k2CompShift)) | (comp << k2CompShift) | ((1u << comp) << k2UsedShift);
                                age = 0u;
                            }
                            cum[i] = t;
                            LAST(id) = lap_time;
                            pk[i] = (p & ~k2AgeMask) | age;
                        }
                    }
                    cur = nxt;
                }
            }

            // ---- _simulate_overtakes, :496-536 ----
            network_sort<N>(cum, pk);
            if (MCGP_DUP & 1) network_sort<N>(cum, pk);
#pragma unroll 1
            for (int pass = 0; pass < ((MCGP_SKIP & 1) ? 0 : 3); ++pass) {
                // pace of every slot (:514-515) and the pace delta of every adjacent pair.  The per-driver
                // constants are gathered half a field at a time (one s_waitcnt per half, not per slot).
                double delta[N];
                uint32_t cand = 0u;
                {
                    constexpr int H = (N + 1) / 2;
                    double pace_prev = 0.0;
#pragma unroll
                    for (int h = 0; h < N; h += H) {
                        double pb[H], pd[H];
#pragma unroll
                        for (int j = 0; j < H; ++j) {
                            if (h + j < N) {
                                const uint32_t id = (pk[h + j] >> k2IdShift) & 31u;
                                pb[j] = t_base[id];
                                pd[j] = t_deg[id];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < H; ++j) {
                            const int i = h + j;
                            if (i < N) {
                                const double pace = pb[j] + (double)(pk[i] & k2AgeMask) * pd[j];
                                if (i > 0) {
                                    double dl = pace_prev - pace;                                   // :516
                                    if (pk[i] & k2Drs) dl += drs_delta;                             // :519-520
                                    delta[i] = dl;
                                    if (!((pk[i] | pk[i - 1]) & k2Dnf) && dl > overtake_delta) cand |= 1u << i;   // :511,522
                                }
                                pace_prev = pace;
                            }
                        }
                    }
                }
                if (cand == 0u) break;
                // the k-th attempt of this pass reads word k & 3 of block 8 * pass + k / 4
                const int n_cand = __popc(cand);
#pragma unroll 1
                for (int b = 0; 4 * b < n_cand; ++b) {
                    uint32_t o0, o1, o2, o3;
                    philox4x32_10(c0, c1, (uint32_t)lap, kPurposeOvt | (uint32_t)(8 * pass + b), seed_lo, seed_hi,
                                  o0, o1, o2, o3);
                    WORD(4 * b + 0) = o0;
                    WORD(4 * b + 1) = o1;
                    WORD(4 * b + 2) = o2;
                    WORD(4 * b + 3) = o3;
                }
                // which attempts succeed (:523-524) does not depend on the times: all draw words are
                // fetched in one batch and compared before the sequential write-back chain
                uint32_t ow[N];
#pragma unroll
                for (int i = 1; i < N; ++i) ow[i] = WORD(__popc(cand & ((1u << i) - 1u)));
                uint32_t succ = 0u;
#pragma unroll
                for (int i = 1; i < N; ++i) {
                    // u < min(0.5, delta / 2)  <=>  w < 2^31  and  w * 2^-31 < delta   (u = w * 2^-32, exact scalings)
                    const bool hit = ow[i] < 0x80000000u && (double)ow[i] * (1.0 / 2147483648.0) < delta[i];
                    if (((cand >> i) & 1u) && hit) succ |= 1u << i;
                }
                if (succ == 0u) break;
                // :525-531, in sorted order, each pair seeing the previous pair's mutation (Q15)
#pragma unroll
                for (int i = 1; i < N; ++i) {
                    if ((succ >> i) & 1u) {
                        double nb = cum[i - 1] - 0.1;
                        if (!(nb > 0.1)) nb = 0.1;
                        cum[i] = nb;
                        cum[i - 1] = nb + 0.3;
                    }
                }
                transposition_sort<N>(cum, pk);     // sorted again for the next pass / _update_positions
                if (MCGP_DUP & 8) transposition_sort<N>(cum, pk);
            }
            update_positions_reg<N>(cum, pk, lap > 2 && lap > drs_disabled_until, dirty_thr);   // :227-228
            if (MCGP_DUP & 4) update_positions_reg<N>(cum, pk, lap > 2 && lap > drs_disabled_until, dirty_thr);
        }

        // ================= classification, reference :230-242 =================
        // rows to LDS (cum -> LAST rows, pk -> WORD rows), insertion sort with the classification
        // order: running cars by time, then retired cars by (lap, time) descending, stable.
#pragma unroll
        for (int i = 0; i < N; ++i) {
            LAST(i) = cum[i];
            WORD(i) = pk[i];
        }
#pragma unroll 1
        for (int i = 1; i < N; ++i) {
            const uint32_t pkx = WORD(i);
            const double kx = LAST(i);
            int j = i;
            while (j > 0) {
                const uint32_t pky = WORD(j - 1);
                const double ky = LAST(j - 1);
                bool y_after_x;
                if (!(pky & k2Dnf)) y_after_x = false;            // runners are already in order and ahead of retirees
                else if (!(pkx & k2Dnf)) y_after_x = true;
                else {
                    const uint32_t ly = pky & k2AgeMask, lx = pkx & k2AgeMask;
                    y_after_x = ly < lx || (ly == lx && (ky < kx || (ky == kx && pky > pkx)));
                }
                if (!y_after_x) break;
                WORD(j) = pky;
                LAST(j) = ky;
                --j;
            }
            WORD(j) = pkx;
            LAST(j) = kx;
        }
#pragma unroll 1
        for (int p = 0; p < N; ++p) {
            const uint32_t d = (WORD(p) >> k2IdShift) & 31u;
            atomicAdd(&s_hist[d * N + p], 1u);                       // reference :93-94
            if (orders) orders[local * (uint64_t)N + (uint64_t)p] = (uint8_t)d;
        }
    }

    __syncthreads();
    for (int i = tid; i < N * N; i += B) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
#undef LAST
#undef ZED
#undef WORD
}

}  // namespace mcgp
