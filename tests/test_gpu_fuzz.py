"""HIP vs oracle (Philox) on the 84 configurations (72 random + 12 corner cases) of tests/golden/fuzz_cases.json
(the same configurations on which the oracle's MT back-end is pinned to the reference)."""
import json
import os

import numpy as np
import pytest

import oracle_py as O
from helpers import product_run

pytestmark = pytest.mark.gpu


def _check(cases, names, n_sims):
    from monte_carlo_gp_amd import _native as N
    kernels = set()
    for name in names:
        c = cases[name]
        ref = O.Problem(c).run(n_sims, rng=O.RNG_PHILOX, seed=c['seed'], want_orders=True)
        hist, _, orders = product_run(c, n_sims, c['seed'], orders=True)
        bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
        assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
        assert np.array_equal(hist, ref['hist']), name
        k = N.lib().mcgp_last_kernel_name(0).decode()
        if not os.environ.get('MCGP_FORCE_GENERIC'):
            # every configuration, certain retirements (X_all_out_lap2: probability >= 1) included, runs on the register
            # kernel -- a retirement lap is drawn once per race, p >= 1 is just a survival threshold of 0 -- except a
            # NEGATIVE overtake_delta (X_all_attempt: attempts at a pace deficit), which reg_kernel_serves hands to the
            # generic kernel
            assert k.startswith('mcgp::race_kernel_reg<') != (c['config']['overtake_delta'] < 0), (name, k)
        kernels.add(k)
    return kernels


def test_fuzzed_configurations_match_oracle(require_gpu, monkeypatch):
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        cases = json.load(f)
    names = list(cases)
    kernels = _check(cases, names, 1500)
    assert len([k for k in kernels if k.startswith('mcgp::race_kernel_reg<')]) >= 10
    # the generic LDS kernel on every third configuration
    monkeypatch.setenv('MCGP_FORCE_GENERIC', '1')
    assert _check(cases, names[::3], 1000) == {'mcgp::race_kernel'}


def test_fuzzed_configurations_at_reference_width(require_gpu):
    """deviates = 53 against the oracle's PHILOX53 back-end on every fuzz configuration the register kernel takes, whatever
    its field size (the corner cases among them -- everybody retiring, events on every lap, a pit stop every lap,
    one-lap races)."""
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        cases = json.load(f)
    done = 0
    for name, c in cases.items():
        if c['config']['overtake_delta'] < 0:
            continue
        ref = O.Problem(c).run(800, rng=O.RNG_PHILOX53, seed=c['seed'], want_orders=True)
        sim = RaceSimulator(RaceConfig(**c['config']), set_pop=O.load_cases()['set_pop'], deviates=53)
        _, orders = sim.run_monte_carlo(800, c['grid_probs'], c['base_pace'], c['tire_deg'], c['driver_variance'],
                                        c.get('driver_dnf_rates'), seed=c['seed'], track_condition=c['track_condition'],
                                        return_orders=True)
        assert N.lib().mcgp_last_kernel_name(0).decode().startswith('mcgp::race_kernel_reg_wide<'), name
        bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
        assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
        assert np.array_equal(sim.last_histogram, ref['hist']), name
        done += 1
    assert done >= 80


def test_fuzzed_configurations_in_one_batch_launch(require_gpu):
    """mcgp_run_batch on the 24 twenty-car fuzz configurations the register kernel takes, in ONE launch: every
    problem's histogram is the oracle's (different circuits, lap counts from 1 to 78, tyre tables, weather, retirement
    rates up to certainty side by side in one grid of blocks)."""
    from monte_carlo_gp_amd import RaceConfig, run_monte_carlo_batch
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        cases = json.load(f)
    names = [k for k, c in cases.items() if len(c['grid_probs']) == 20 and c['config']['overtake_delta'] >= 0]
    assert len(names) >= 24
    n_sims = 1500
    problems = [dict(config=RaceConfig(**cases[k]['config']), grid_probs=cases[k]['grid_probs'], base_pace=cases[k]['base_pace'],
                     tire_deg=cases[k]['tire_deg'], driver_variance=cases[k]['driver_variance'],
                     driver_dnf_rates=cases[k].get('driver_dnf_rates'), seed=cases[k]['seed'],
                     track_condition=cases[k]['track_condition'], sim_offset=7 * i) for i, k in enumerate(names)]
    out = run_monte_carlo_batch(problems, n_sims, set_pop=O.load_cases()['set_pop'])
    for i, (k, (_, hist)) in enumerate(zip(names, out)):
        ref = O.Problem(cases[k]).run(n_sims, rng=O.RNG_PHILOX, seed=cases[k]['seed'], sim_offset=7 * i)['hist']
        assert np.array_equal(hist, ref), k


def test_batch_with_more_problems_than_blocks(require_gpu):
    """mcgp_run_batch with 700 ten-car problems of 130 simulations (3 wave-chunks, the last one ragged): more problems than the
    launch has blocks, so every block walks through several of them (table reloads, the 64-at-a-time scan of the per-problem
    ticket counters going round the end of the list), and most of a block's waves find a problem's counter already exhausted.
    Every histogram is the oracle's; seeds, offsets and retirement rates differ from problem to problem."""
    import copy
    from monte_carlo_gp_amd import RaceConfig, run_monte_carlo_batch, _native as N
    base = O.load_case('N10')
    set_pop = O.load_cases()['set_pop']
    n_sims, k = 130, 700
    drivers = list(base['base_pace'])
    cases = []
    for i in range(k):
        c = copy.deepcopy(base)
        c['config']['total_laps'] = 12 + i % 9
        c['driver_dnf_rates'] = {d: 0.002 * (1 + (i + j) % 7) for j, d in enumerate(drivers)}
        c['base_pace'] = {d: v + 0.01 * ((i * 7 + j) % 13) for j, (d, v) in enumerate(base['base_pace'].items())}
        cases.append(c)
    problems = [dict(config=RaceConfig(**c['config']), grid_probs=c['grid_probs'], base_pace=c['base_pace'], tire_deg=c['tire_deg'],
                     driver_variance=c['driver_variance'], driver_dnf_rates=c['driver_dnf_rates'], seed=1000 + i,
                     track_condition=c['track_condition'], sim_offset=64 * i + 5) for i, c in enumerate(cases)]
    out = run_monte_carlo_batch(problems, n_sims, set_pop=set_pop)
    assert N.lib().mcgp_last_kernel_name(0).decode() == 'mcgp::race_kernel_reg_batch<10>'
    assert len(out) == k
    for i, (c, (_, hist)) in enumerate(zip(cases, out)):
        assert (hist.sum(axis=0) == n_sims).all() and (hist.sum(axis=1) == n_sims).all(), i
        if i % 7 == 0 or i >= k - 3:
            ref = O.Problem(c).run(n_sims, rng=O.RNG_PHILOX, seed=1000 + i, sim_offset=64 * i + 5)['hist']
            assert np.array_equal(hist, ref), i
