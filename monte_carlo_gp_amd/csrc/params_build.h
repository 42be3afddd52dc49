// params_build.h -- folds the C-ABI inputs (include/mcgp.h) into the kernel's parameter block.
// Plain C++ (no HIP calls): shared by the C-ABI library (mcgp_hip.hip) and the host debugging build
// of the kernel sources (tools/emu).
#pragma once
#include "../../include/mcgp.h"
#include "normal_table.h"
#include "race_common.hip.h"

#include <cmath>
#include <cstring>

namespace mcgp {

// u < p for u = w / 2^32  <=>  w < ceil(p * 2^32)   (p * 2^32 is exact in binary64)
inline uint64_t threshold(double p)
{
    if (!(p > 0.0)) return 0;                  // also NaN: `u < nan` is false
    const double x = p * 4294967296.0;
    if (x >= 4294967296.0) return 4294967296ull;
    return (uint64_t)std::ceil(x);
}

// u < p for the 53-bit uniform u = q / 2^53  <=>  q < ceil(p * 2^53)
inline uint64_t threshold53(double p)
{
    if (!(p > 0.0)) return 0;
    const double x = p * 9007199254740992.0;
    if (x >= 9007199254740992.0) return 9007199254740992ull;
    return (uint64_t)std::ceil(x);
}
// survival factor of the 64-bit retirement chain: 2^64 - ceil(p 2^64); 0 for p >= 1 (and for p <= 0, where the chain is
// not used: t_dnf = 0 says "never")
inline uint64_t survival64(double p)
{
    if (!(p > 0.0) || p >= 1.0) return 0;
    return (uint64_t)0 - (uint64_t)std::ceil(p * 18446744073709551616.0);      // p 2^64 <= 2^64 - 2^11: fits
}

// Returns MCGP_OK or MCGP_E_BAD_ARG with *err set to a static message.
inline int build_params(const mcgp_config *cfg, const mcgp_drivers *drv, const double *grid_probs, uint32_t n,
                        KParams *kp, const char **err)
{
    if (!cfg || !drv) { *err = "cfg / drv is NULL"; return MCGP_E_BAD_ARG; }
    if (n < 1 || n > MCGP_MAX_CARS) { *err = "n must be in [1, 32]"; return MCGP_E_BAD_ARG; }
    if (cfg->total_laps < 1 || cfg->total_laps > MCGP_MAX_LAPS)
        { *err = "total_laps must be in [1, 1000]"; return MCGP_E_BAD_ARG; }
    if (cfg->track_condition < MCGP_DRY || cfg->track_condition > MCGP_WET_TRACK)
        { *err = "track_condition must be 0 (dry), 1 (damp) or 2 (wet)"; return MCGP_E_BAD_ARG; }
    if (cfg->pop_soft_hard != MCGP_SOFT && cfg->pop_soft_hard != MCGP_HARD)
        { *err = "pop_soft_hard must be SOFT or HARD"; return MCGP_E_BAD_ARG; }
    if (cfg->pop_medium_hard != MCGP_MEDIUM && cfg->pop_medium_hard != MCGP_HARD)
        { *err = "pop_medium_hard must be MEDIUM or HARD"; return MCGP_E_BAD_ARG; }
    if (!drv->base_pace || !drv->tire_deg || !drv->tire_deg_pit || !drv->variance || !drv->team_dnf ||
        !drv->lap_dnf)
        { *err = "a per-driver array is NULL"; return MCGP_E_BAD_ARG; }
    std::memset(kp, 0, sizeof(*kp));
    kp->n = (int32_t)n;
    kp->total_laps = cfg->total_laps;
    kp->track = cfg->track_condition;
    kp->pop_sh = cfg->pop_soft_hard;
    kp->pop_mh = cfg->pop_medium_hard;
    kp->pit_loss = cfg->pit_loss;
    kp->overtake_delta = cfg->overtake_delta;
    kp->overtake_delta_31 = cfg->overtake_delta * 2147483648.0;
    kp->drs_delta = cfg->drs_delta;
    kp->dirty_thr = cfg->dirty_air_threshold;
    kp->dirty_pen = cfg->dirty_air_penalty;
    kp->t_red = threshold(cfg->red_flag_probability);
    kp->t_sc = threshold(cfg->sc_probability);
    kp->t_vsc = threshold(cfg->vsc_probability);
    kp->t_vsc_tire = threshold(0.3);                                   // reference :392
    if (cfg->deviates != MCGP_DEVIATES_32 && cfg->deviates != MCGP_DEVIATES_53)
        { *err = "deviates must be MCGP_DEVIATES_32 or MCGP_DEVIATES_53"; return MCGP_E_BAD_ARG; }
    kp->wide = cfg->deviates == MCGP_DEVIATES_53;
    kp->t53_red = threshold53(cfg->red_flag_probability);
    kp->t53_sc = threshold53(cfg->sc_probability);
    kp->t53_vsc = threshold53(cfg->vsc_probability);
    kp->t53_vsc_tire = threshold53(0.3);
    for (int c = 0; c < 5; ++c) {
        kp->comp_deg[c] = cfg->comp_deg_rate[c];
        kp->comp_delta[c] = cfg->comp_pace_delta[c];
        if (cfg->comp_optimal_laps[c] < 0 || cfg->comp_optimal_laps[c] > 50000)
            { *err = "comp_optimal_laps out of range"; return MCGP_E_BAD_ARG; }
    }
    for (uint32_t d = 0; d < n; ++d) {
        const double deg = drv->tire_deg[d];
        kp->base_pace[d] = drv->base_pace[d];
        kp->factor[d] = deg > 0 ? deg / 0.05 : 1.0;                    // reference :321
        kp->tire_deg[d] = deg;
        kp->variance[d] = drv->variance[d];
        kp->t_dnf1[d] = threshold(drv->team_dnf[d] * 4.0);             // reference :282,286-287
        kp->t_dnf[d] = threshold(drv->lap_dnf[d]);
        kp->t53_dnf1[d] = threshold53(drv->team_dnf[d] * 4.0);
        kp->q64_dnf[d] = survival64(drv->lap_dnf[d]);
        const double pit_deg = drv->tire_deg_pit[d];
        for (int c = 0; c < 5; ++c) {
            int opt = cfg->comp_optimal_laps[c];                       // reference :455-462
            if (pit_deg > 0.05) opt = (int)((double)opt * 0.85);
            else if (pit_deg < 0.02) opt = (int)((double)opt * 1.1);
            kp->opt_laps[d * kCompStride + c] = (uint16_t)opt;
        }
    }
    if (grid_probs)
        for (uint32_t d = 0; d < n; ++d)
            for (uint32_t s = 0; s < n; ++s) {
                const double p = grid_probs[(size_t)d * n + s];
                if (!(p >= 0.0)) { *err = "grid_probs has a negative or NaN entry"; return MCGP_E_BAD_ARG; }
                kp->grid_probs[(size_t)d * n + s] = p;
            }
    static_assert(sizeof(mcgp_normal_table_bits) == sizeof(kp->normal_bits), "normal table size");
    std::memcpy(kp->normal_bits, mcgp_normal_table_bits, sizeof(kp->normal_bits));
    return MCGP_OK;
}

}  // namespace mcgp
