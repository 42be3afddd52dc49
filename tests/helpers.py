"""Shared test helpers: build product-side objects from a golden case dict."""
import numpy as np

import oracle_py as O
from monte_carlo_gp_amd import RaceConfig, RaceSimulator


def product_sim(case, device=0, deviates=32):
    cfg = RaceConfig(**case['config'])
    return RaceSimulator(cfg, device=device, set_pop=O.load_cases()['set_pop'], deviates=deviates)


def product_run(case, n_sims, seed, sim_offset=0, orders=False, device=0, deviates=32):
    sim = product_sim(case, device, deviates)
    out = sim.run_monte_carlo(n_sims, case['grid_probs'], case['base_pace'], case['tire_deg'],
                              case['driver_variance'], case['driver_dnf_rates'], seed=seed,
                              track_condition=case['track_condition'], sim_offset=sim_offset,
                              return_orders=orders)
    if orders:
        probs, o = out
        return sim.last_histogram, probs, o
    return sim.last_histogram, out, None


def probs_to_hist(probs, drivers, n_sims):
    n = len(drivers)
    h = np.zeros((n, n), np.int64)
    for i, d in enumerate(drivers):
        for pos, p in probs.get(d, {}).items():
            h[i, pos - 1] = int(round(p * n_sims))
    return h
