#!/usr/bin/env python3
"""Device time of mcgp_run_batch for a season-sized sweep (24 twenty-car races), per simulations-per-race; MCGP_LIB selects a variant.
    python tools/batch_time.py [sims ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_py as O
from monte_carlo_gp_amd import RaceConfig, run_monte_carlo_batch, _native as N
names = ['S60', 'S78', 'S50', 'EVT', 'DMP', 'WET']
cases = {k: O.load_case(k) for k in names}
def problem(i):
    c = cases[names[i % 6]]
    return dict(config=RaceConfig(**c['config']), grid_probs=c['grid_probs'], base_pace=c['base_pace'], tire_deg=c['tire_deg'],
                driver_variance=c['driver_variance'], driver_dnf_rates=c['driver_dnf_rates'], seed=1000 + i, track_condition=c['track_condition'])
probs = [problem(i) for i in range(24)]
for n in [int(x) for x in sys.argv[1:]] or [10_000, 100_000]:
    best = 1e9
    for rep in range(6):
        run_monte_carlo_batch(probs, n, device=0, set_pop=O.load_cases()['set_pop'])
        ms = C.c_float(); N.check(N.lib().mcgp_last_kernel_ms(0, C.byref(ms)))
        best = min(best, ms.value)
    print(f'batch 24 x {n}: {best:.3f} ms  ({os.path.basename(os.environ.get("MCGP_LIB", "product"))})')
