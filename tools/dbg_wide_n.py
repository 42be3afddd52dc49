import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_py as O
from test_gpu_parity import _field
from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N
for n in [int(x) for x in sys.argv[1:]] or [29, 30, 31, 32]:
    c = _field(n)
    ref = O.Problem(c).run(1500, rng=O.RNG_PHILOX53, seed=7, want_orders=True)
    sim = RaceSimulator(RaceConfig(**c['config']), set_pop=O.load_cases()['set_pop'], deviates=53)
    _, orders = sim.run_monte_carlo(1500, c['grid_probs'], c['base_pace'], c['tire_deg'], c['driver_variance'],
                                    c['driver_dnf_rates'], seed=7, track_condition=c['track_condition'], return_orders=True)
    bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
    print(n, N.lib().mcgp_last_kernel_name(0).decode(), 'bad', bad.size, bad[:10])
