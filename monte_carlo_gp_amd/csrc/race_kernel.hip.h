// race_kernel.hip.h -- the whole-race Monte Carlo kernel for CDNA4 (gfx950).
//
// One LANE runs one simulation (one full race: grid sampling, lap 1, laps 2..L,
// classification); a 64-lane wavefront therefore advances 64 independent races in
// lock step and every per-car quantity is a structure-of-arrays row in LDS:
//
//     cum [car][lane]  f64   cumulative race time        reference CarState.cumulative_time :20
//     last[car][lane]  f64   last lap time               reference CarState.last_lap_time   :29
//     pk  [car][lane]  u32   tyre age | compound | used compounds | grid slot | dnf | drs | dirty-air flag
//     ord [rank][lane] u8    driver index at time-rank r (ALL cars, stable by grid slot)
//
// "car" is the DRIVER index (row order of grid_probs), not the grid slot: with a
// counter-based RNG the per-car loops of the reference (src/simulation.py:186,284,450)
// carry no order dependence, so they run in time-rank order, which lets the
// dirty-air "car ahead" value (reference :179-183) ride along in a register.
// Row stride = blockDim.x, so lane l touches bank (2l mod 64) whatever car it
// indexes: every LDS access of the kernel, including the per-lane gathers by
// driver index, is bank-conflict free.
//
// Everything that orders cars follows Python's stable sort: key
// (cumulative_time, grid slot) -- reference :179,231,353,384,410,506,549 sort lists
// that are in grid order (Q1), so equal times keep grid order.
//
// Floating point: IEEE binary64 in the reference's evaluation order; this file
// is compiled with -ffp-contract=off.  The only fused operations are the three
// explicit binary32 fmaf of the inverse-normal cubic.
#pragma once
#include "race_common.hip.h"

namespace mcgp {

// pk word of THIS kernel (the register kernel has its own layout: k3* in race_kernel_reg.hip.h)
constexpr uint32_t kAgeMask = 0x3FFu;          // tyre age; lap of retirement once dnf is set
constexpr int kCompShift = 10;                 // 3 bits
constexpr int kUsedShift = 13;                 // 5 bits, one per compound
constexpr int kGposShift = 18;                 // 5 bits, grid slot
constexpr uint32_t kDnf = 1u << 23;
constexpr uint32_t kDrs = 1u << 24;
constexpr uint32_t kDirty = 1u << 25;          // 0 < time_behind_leader < dirty_air_threshold


// Per-lane view of the LDS rows.
struct Rows {
    double *cum;
    double *last;
    uint32_t *pk;
    uint8_t *ord;
    uint16_t *out;   // lap of the driver's retirement (laps >= 2), 0 = not in this race: drawn once per race
    int B;      // row stride (threads per block)
    int tid;
    __device__ __forceinline__ double &Cum(uint32_t d) const { return cum[d * B + tid]; }
    __device__ __forceinline__ double &Last(uint32_t d) const { return last[d * B + tid]; }
    __device__ __forceinline__ uint32_t &Pk(uint32_t d) const { return pk[d * B + tid]; }
    __device__ __forceinline__ uint8_t &Ord(int i) const { return ord[i * B + tid]; }
    __device__ __forceinline__ uint16_t &Out(uint32_t d) const { return out[d * B + tid]; }
};

__device__ __forceinline__ uint32_t gpos_of(uint32_t pk) { return (pk >> kGposShift) & 31u; }

// Stable sort of `ord` over ALL cars by (cumulative_time, grid slot): Python's
// sorted(cars, key=cumulative_time) on the grid-ordered list (reference :506).
// Restricted to running cars it is also the order of reference :179,231,353,384,410,549.
// Insertion sort: the field is nearly sorted from the previous lap.
__device__ __forceinline__ void sort_by_time(const Rows &s, int n)
{
    uint32_t dtop = s.Ord(0);
    double ktop = s.Cum(dtop);
    uint32_t gtop = gpos_of(s.Pk(dtop));
    for (int i = 1; i < n; ++i) {
        const uint32_t x = s.Ord(i);
        const double kx = s.Cum(x);
        const uint32_t gx = gpos_of(s.Pk(x));
        if (ktop > kx || (ktop == kx && gtop > gx)) {
            int j = i;
            s.Ord(j) = (uint8_t)dtop;
            --j;
            while (j > 0) {
                const uint32_t y = s.Ord(j - 1);
                const double ky = s.Cum(y);
                if (!(ky > kx || (ky == kx && gpos_of(s.Pk(y)) > gx))) break;
                s.Ord(j) = (uint8_t)y;
                --j;
            }
            s.Ord(j) = (uint8_t)x;
        } else {
            dtop = x; ktop = kx; gtop = gx;
        }
    }
}

// _update_positions, reference :538-560: gap to the leader (kept as the dirty-air
// predicate of reference :209-212) and the DRS flag.
__device__ __forceinline__ void update_positions(const Rows &s, int n, bool drs_allowed, double dirty_thr)
{
    bool first = true;
    double leader = 0.0, prev = 0.0;
    for (int i = 0; i < n; ++i) {
        const uint32_t d = s.Ord(i);
        uint32_t pk = s.Pk(d);
        if (pk & kDnf) continue;
        const double t = s.Cum(d);
        if (first) { leader = t; }
        const double tbl = t - leader;
        pk &= ~(kDrs | kDirty);
        if (tbl > 0 && tbl < dirty_thr) pk |= kDirty;
        if (!first && drs_allowed && (t - prev) < 1.0) pk |= kDrs;
        s.Pk(d) = pk;
        prev = t;
        first = false;
    }
}

__global__ void __launch_bounds__(512)
race_kernel(const KParams *__restrict__ P, uint64_t n_sims, uint64_t sim_offset,
            uint32_t seed_lo, uint32_t seed_hi, unsigned long long *__restrict__ hist,
            uint8_t *__restrict__ orders, const uint8_t *__restrict__ fixed_grid, uint32_t n_batches,
            uint32_t * /*ticket: the register kernel's work counter; batches are dealt out by block index here*/,
            uint32_t * /*retire_ws: the register kernel's retirement lists; an LDS row per driver here*/)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int B = blockDim.x;
    const int n = P->n;
    const int L = P->total_laps;
    const int track = P->track;

    // ---- LDS carve-up (offsets are multiples of 16) ----
    float4 *t_norm = reinterpret_cast<float4 *>(smem);
    double *t_base = reinterpret_cast<double *>(smem + kNormalRows * 16);
    double *t_factor = t_base + kMaxCars;
    double *t_deg = t_factor + kMaxCars;
    double *t_var = t_deg + kMaxCars;
    unsigned long long *t_dnf = reinterpret_cast<unsigned long long *>(t_var + kMaxCars);
    unsigned long long *t_dnf1 = t_dnf + kMaxCars;
    double *t_cdeg = reinterpret_cast<double *>(t_dnf1 + kMaxCars);
    double *t_cdelta = t_cdeg + kCompStride;
    uint16_t *t_opt = reinterpret_cast<uint16_t *>(t_cdelta + kCompStride);
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(t_opt + kMaxCars * kCompStride);
    Rows s;
    s.cum = reinterpret_cast<double *>(s_hist + kMaxCars * kMaxCars);
    s.last = s.cum + (size_t)n * B;
    s.pk = reinterpret_cast<uint32_t *>(s.last + (size_t)n * B);
    s.ord = reinterpret_cast<uint8_t *>(s.pk + (size_t)n * B);
    s.out = reinterpret_cast<uint16_t *>(s.ord + (size_t)n * B);
    s.B = B;
    s.tid = tid;

    for (int i = tid; i < kNormalRows * 4; i += B)
        reinterpret_cast<uint32_t *>(t_norm)[i] = P->normal_bits[i];
    for (int i = tid; i < kMaxCars; i += B) {
        t_base[i] = P->base_pace[i];
        t_factor[i] = P->factor[i];
        t_deg[i] = P->tire_deg[i];
        t_var[i] = P->variance[i];
        t_dnf[i] = P->t_dnf[i];
        t_dnf1[i] = P->t_dnf1[i];
    }
    for (int i = tid; i < kCompStride; i += B) {
        t_cdeg[i] = P->comp_deg[i];
        t_cdelta[i] = P->comp_delta[i];
    }
    for (int i = tid; i < kMaxCars * kCompStride; i += B) t_opt[i] = P->opt_laps[i];
    for (int i = tid; i < n * n; i += B) s_hist[i] = 0u;
    __syncthreads();

    const double pit_loss = P->pit_loss;
    const double overtake_delta = P->overtake_delta;
    const double drs_delta = P->drs_delta;
    const double dirty_thr = P->dirty_thr;
    const double dirty_pen = P->dirty_pen;

    for (uint32_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const uint64_t local = (uint64_t)batch * (uint64_t)B + (uint64_t)tid;
        if (local >= n_sims) continue;      // tail lanes idle; no barrier inside the loop
        const uint64_t sim = sim_offset + local;
        const uint32_t c0 = (uint32_t)sim, c1 = (uint32_t)(sim >> 32);

        // ================= _sample_grid, reference :102-145 =================
        // probs / cdf scratch lives in the `last` rows (not needed until lap 2).
        {
            uint32_t remaining = (n >= 32) ? 0xffffffffu : ((1u << n) - 1u);
            int n_remaining = n;
            uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
            for (int pos = 0; pos < n; ++pos) {
                uint32_t sel;
                if (fixed_grid) {
                    sel = fixed_grid[pos];
                } else {
                    if ((pos & 3) == 0)
                        philox4x32_10(c0, c1, 0u, kPurposeGrid | (uint32_t)(pos >> 2), seed_lo, seed_hi, g0, g1, g2, g3);
                    const uint32_t gw = (pos & 3) == 0 ? g0 : (pos & 3) == 1 ? g1 : (pos & 3) == 2 ? g2 : g3;
                    const double u = u32_to_unit(gw);
                    double total = 0.0;                                   // :119-123
                    for (int d = 0; d < n; ++d) {
                        const double p = ((remaining >> d) & 1u) ? P->grid_probs[d * n + pos] : 0.0;
                        total = total + p;
                    }
                    double prob_sum = 0.0;                                // :125-133
                    for (int d = 0; d < n; ++d) {
                        const bool rem = (remaining >> d) & 1u;
                        double p;
                        if (total > 0) p = (rem ? P->grid_probs[d * n + pos] : 0.0) / total;
                        else p = rem ? 1.0 / (double)n_remaining : 0.0;
                        s.Last(d) = p;
                        prob_sum = prob_sum + p;
                    }
                    const bool renorm = prob_sum > 0 && fabs(prob_sum - 1.0) > 1e-9;   // :134-135
                    // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right')
                    double acc = 0.0;
                    for (int d = 0; d < n; ++d) {
                        double p = s.Last(d);
                        if (renorm) p = p / prob_sum;
                        acc = (d == 0) ? p : acc + p;
                        s.Last(d) = acc;
                    }
                    const double cdf_last = acc;
                    sel = 0;
                    for (int d = 0; d < n; ++d)
                        if (s.Last(d) / cdf_last <= u) sel = (uint32_t)d + 1u;
                    if (sel >= (uint32_t)n) sel = (uint32_t)n - 1u;       // unreachable: cdf[-1] == 1 > u
                }
                if ((remaining >> sel) & 1u) { remaining &= ~(1u << sel); --n_remaining; }
                // _initialize_cars, reference :244-273
                uint32_t comp, age;
                if (track == 2) { comp = 4u; age = 0u; }
                else if (track == 1) { comp = 3u; age = 0u; }
                else { comp = pos < 10 ? 0u : 1u; age = pos < 10 ? 4u : 0u; }
                s.Pk(sel) = age | (comp << kCompShift) | ((1u << comp) << kUsedShift) | ((uint32_t)pos << kGposShift);
                s.Cum(sel) = 0.0;
                s.Ord(pos) = (uint8_t)sel;
            }
            for (int d = 0; d < n; ++d) s.Last(d) = 0.0;
        }

        // ================= _simulate_lap_1, reference :275-311 =================
        for (int pos = 0; pos < n; ++pos) {
            const uint32_t d = s.Ord(pos);
            uint32_t pk = s.Pk(d);
            uint32_t w0, w1, w2, w3;
            philox4x32_10(c0, c1, 1u, kPurposeCar | d, seed_lo, seed_hi, w0, w1, w2, w3);
            if ((uint64_t)w0 < t_dnf1[d]) {
                s.Pk(d) = (pk & ~kAgeMask) | kDnf | 1u;
                continue;
            }
            const uint32_t comp = (pk >> kCompShift) & 7u;
            const uint32_t age = pk & kAgeMask;
            const double eff = t_cdeg[comp] * t_factor[d];
            const double tire = (double)age * eff;
            const double fuel_effect = (110.0 - 110.0) * 0.03;
            const double noise = 0.0 + t_var[d] * (double)normal_from_u32(w1, t_norm);
            const double base_lap = t_base[d] + tire - fuel_effect + t_cdelta[comp] - 0.0 + noise;
            double pf = 0.5 + (double)(pos + 1) * 0.1;
            if (!(pf < 1.5)) pf = 1.5;
            double sd = 0.0 + pf * (double)normal_from_u32(w2, t_norm);
            if (pos + 1 <= 3 && 1.0 < sd) sd = 1.0;
            const double lap_time = base_lap - sd * 0.5;
            s.Cum(d) = 0.0 + lap_time;
            s.Pk(d) = (pk & ~kAgeMask) | (age + 1u);
        }
        sort_by_time(s, n);
        update_positions(s, n, false, dirty_thr);

        // ================= retirements of laps 2..L (:190-197), drawn once per race: race_common.hip.h =================
        {
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            for (int d = 0; d < n; ++d) {
                if ((d & 3) == 0)
                    philox4x32_10(c0, c1, 0u, kPurposeRetire | (uint32_t)(d >> 2), seed_lo, seed_hi, r0, r1, r2, r3);
                const uint32_t rw = (d & 3) == 0 ? r0 : (d & 3) == 1 ? r1 : (d & 3) == 2 ? r2 : r3;
                s.Out(d) = (uint16_t)draw_retirement_lap(rw, t_dnf[d], L);
            }
        }

        // ================= laps 2..L, reference :166-228 =================
        int drs_disabled_until = 0;
        for (int lap = 2; lap <= L; ++lap) {
            const int remaining_laps = L - lap;
            // ---- race-interrupting events, :168-176 (short-circuit chain, Q8) ----
            {
                uint32_t e0, e1, e2, e3;
                philox4x32_10(c0, c1, (uint32_t)lap, kPurposeEvent, seed_lo, seed_hi, e0, e1, e2, e3);
                const bool red = (uint64_t)e0 < P->t_red;
                const bool sc = !red && (uint64_t)e1 < P->t_sc;
                const bool vsc = !red && !sc && (uint64_t)e2 < P->t_vsc;
                if (red || sc || vsc) {
                    // _handle_red_flag :397-431 / _handle_safety_car :334-376 / _handle_vsc :378-395
                    const bool dec_age = sc || (vsc && (uint64_t)e3 < P->t_vsc_tire);
                    const uint32_t newc = stint_compound(track, remaining_laps);
                    int k = 0;
                    double leader = 0.0, prev_nt = -1.0;
                    bool tie = false;
                    for (int i = 0; i < n; ++i) {
                        const uint32_t d = s.Ord(i);
                        uint32_t pk = s.Pk(d);
                        if (pk & kDnf) continue;
                        const double t = s.Cum(d);
                        if (k == 0) leader = t;
                        double nt;
                        if (red) nt = leader + (double)k * 0.1;
                        else if (sc) nt = leader + (double)k * 0.5;
                        else { const double gap = t - leader; nt = leader + gap * 0.8; }
                        tie |= nt == prev_nt;
                        prev_nt = nt;
                        const double tbl = nt - leader;
                        pk &= ~kDirty;
                        if (tbl > 0 && tbl < dirty_thr) pk |= kDirty;
                        uint32_t age = pk & kAgeMask;
                        if (red) {
                            age = 0u;
                            pk = (pk & ~(7u << kCompShift)) | (newc << kCompShift) | ((1u << newc) << kUsedShift);
                        } else if (dec_age) {
                            age = age > 0u ? age - 1u : 0u;
                        }
                        pk = (pk & ~kAgeMask) | age;
                        s.Cum(d) = nt;
                        s.Pk(d) = pk;
                        ++k;
                    }
                    drs_disabled_until = lap + (vsc ? 1 : 2);
                    // x0.8 is monotone but may round two gaps together: equal times fall back to grid order.  Otherwise
                    // the field order (`ord`, which also addresses this lap's draws) stays what the last lap left.
                    if (tie) sort_by_time(s, n);
                }
            }

            // ---- every running car's lap, :179-223, with the pit stop of :433-494 folded in ----
            {
                double fuel = 110.0 - 1.5 * (double)(lap - 1);     // fuel_load before this lap (Q7)
                if (!(fuel > 0)) fuel = 0.0;
                const double fuel_effect = (110.0 - fuel) * 0.03;
                double carry = 0.0;                                 // car_ahead_times.get(driver, 0)
                for (int i = 0; i < n; ++i) {
                    const uint32_t d = s.Ord(i);
                    uint32_t pk = s.Pk(d);
                    if (pk & kDnf) continue;
                    const double ahead_last = carry;
                    carry = s.Last(d);
                    if ((int)s.Out(d) == lap) {                     // :194-197, drawn before the race
                        s.Pk(d) = (pk & ~kAgeMask) | kDnf | (uint32_t)lap;
                        continue;
                    }
                    uint32_t w0, w1, w2, w3;
                    // the lap noise is addressed by the car's place i in the field order: one block serves places 4j .. 4j+3
                    philox4x32_10(c0, c1, (uint32_t)lap, kPurposeCar | ((uint32_t)i >> 2), seed_lo, seed_hi, w0, w1, w2, w3);
                    const uint32_t w1x = (i & 3) == 0 ? w0 : (i & 3) == 1 ? w1 : (i & 3) == 2 ? w2 : w3;
                    w1 = w1x;
                    uint32_t comp = (pk >> kCompShift) & 7u;
                    uint32_t age = pk & kAgeMask;
                    // _calculate_lap_time :313-332
                    const double eff = t_cdeg[comp] * t_factor[d];
                    const double tire = (double)age * eff;
                    const double drs_gain = (pk & kDrs) ? drs_delta : 0.0;
                    const double noise = 0.0 + t_var[d] * (double)normal_from_u32(w1, t_norm);
                    const double clean = t_base[d] + tire - fuel_effect + t_cdelta[comp] - drs_gain + noise;
                    double lap_time = clean;
                    if ((pk & kDirty) && ahead_last > 0) {          // :209-216
                        const double dirty_time = clean + dirty_pen;
                        lap_time = ahead_last > dirty_time ? ahead_last : dirty_time;
                    }
                    double t = s.Cum(d) + lap_time;
                    age += 1u;
                    // _handle_pit_stops :454-492 (no randomness, no cross-car dependence)
                    if ((int)age > (int)t_opt[d * kCompStride + comp] && remaining_laps > 5) {
                        t = t + pit_loss;
                        uint32_t newc = stint_compound(track, remaining_laps);
                        const uint32_t used_dry = (pk >> kUsedShift) & 7u;
                        if (track == 0 && __popc(used_dry) == 1 && ((used_dry >> newc) & 1u)) {
                            const uint32_t avail = 7u & ~used_dry;
                            const uint32_t popped = avail == 5u ? (uint32_t)P->pop_sh : avail == 6u ? (uint32_t)P->pop_mh : 0u;
                            if (remaining_laps > 20) newc = (avail & 2u) ? 1u : popped;
                            else newc = (avail & 1u) ? 0u : popped;
                        }
                        comp = newc;
                        pk = (pk & ~(7u << kCompShift)) | (comp << kCompShift) | ((1u << comp) << kUsedShift);
                        age = 0u;
                    }
                    s.Cum(d) = t;
                    s.Last(d) = lap_time;
                    s.Pk(d) = (pk & ~kAgeMask) | age;
                }
            }

            // ---- _simulate_overtakes, :496-536 ----
            bool need_sort = true;
            for (int pass = 0; pass < 3; ++pass) {
                sort_by_time(s, n);
                need_sort = false;
                // adjacent pairs that attempt a pass: neither retired, pace delta over the threshold.
                uint32_t cand = 0u;
                {
                    uint32_t dp = s.Ord(0);
                    uint32_t pkp = s.Pk(dp);
                    double pace_p = t_base[dp] + (double)(pkp & kAgeMask) * t_deg[dp];
                    for (int i = 1; i < n; ++i) {
                        const uint32_t d = s.Ord(i);
                        const uint32_t pk = s.Pk(d);
                        const double pace = t_base[d] + (double)(pk & kAgeMask) * t_deg[d];
                        double delta = pace_p - pace;
                        if (pk & kDrs) delta += drs_delta;
                        if (!((pk | pkp) & kDnf) && delta > overtake_delta) cand |= 1u << i;
                        pkp = pk;
                        pace_p = pace;
                    }
                }
                if (cand == 0u) break;
                bool success = false;
                uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
                for (uint32_t k = 0; cand != 0u; ++k) {
                    if ((k & 3u) == 0u)
                        philox4x32_10(c0, c1, (uint32_t)lap, kPurposeOvt | ((uint32_t)(8 * pass) + (k >> 2)),
                                      seed_lo, seed_hi, o0, o1, o2, o3);
                    const uint32_t ow = (k & 3u) == 0u ? o0 : (k & 3u) == 1u ? o1 : (k & 3u) == 2u ? o2 : o3;
                    const int i = __ffs((int)cand) - 1;
                    cand &= cand - 1u;
                    const uint32_t db = s.Ord(i), da = s.Ord(i - 1);
                    const uint32_t pkb = s.Pk(db), pka = s.Pk(da);
                    const double pace_b = t_base[db] + (double)(pkb & kAgeMask) * t_deg[db];
                    const double pace_a = t_base[da] + (double)(pka & kAgeMask) * t_deg[da];
                    double delta = pace_a - pace_b;
                    if (pkb & kDrs) delta += drs_delta;
                    double prob = delta / 2.0;
                    if (!(prob < 0.5)) prob = 0.5;
                    if (u32_to_unit(ow) < prob) {
                        double nb = s.Cum(da) - 0.1;
                        if (!(nb > 0.1)) nb = 0.1;
                        s.Cum(db) = nb;
                        s.Cum(da) = nb + 0.3;
                        success = true;
                    }
                }
                if (!success) break;
                need_sort = true;
            }
            if (need_sort) sort_by_time(s, n);
            update_positions(s, n, lap > 2 && lap > drs_disabled_until, dirty_thr);   // :227-228
        }

        // ================= classification, reference :230-242 =================
        // running cars by time, then retired cars by (lap, time) descending, stable.
        {
            int n_dnf = 0;
            int w = 0;
            // compact running cars to the front (order kept), retired ones to scratch in `pk`-free bytes:
            // simple two-pass walk using the `last` rows (dead now) as byte scratch is avoided; instead
            // insertion-sort `ord` in place with the classification comparator.
            for (int i = 1; i < n; ++i) {
                const uint32_t x = s.Ord(i);
                const uint32_t pkx = s.Pk(x);
                const double kx = s.Cum(x);
                int j = i;
                while (j > 0) {
                    const uint32_t y = s.Ord(j - 1);
                    const uint32_t pky = s.Pk(y);
                    bool y_after_x;   // does y classify behind x?
                    if (!(pky & kDnf)) {
                        // y running: it is ahead of every retired car; among runners `ord` is already sorted
                        y_after_x = false;
                    } else if (!(pkx & kDnf)) {
                        y_after_x = true;          // retired behind running
                    } else {
                        const uint32_t ly = pky & kAgeMask, lx = pkx & kAgeMask;
                        const double ky = s.Cum(y);
                        // descending (lap, time); ties keep grid order
                        y_after_x = ly < lx || (ly == lx && (ky < kx || (ky == kx && gpos_of(pky) > gpos_of(pkx))));
                    }
                    if (!y_after_x) break;
                    s.Ord(j) = (uint8_t)y;
                    --j;
                }
                s.Ord(j) = (uint8_t)x;
            }
            (void)n_dnf; (void)w;
            for (int p = 0; p < n; ++p) {
                const uint32_t d = s.Ord(p);
                atomicAdd(&s_hist[d * n + p], 1u);                       // reference :93-94
                if (orders) orders[local * (uint64_t)n + (uint64_t)p] = (uint8_t)d;
            }
        }
    }

    __syncthreads();
    for (int i = tid; i < n * n; i += B) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

}  // namespace mcgp
