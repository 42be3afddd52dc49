"""HIP vs oracle (Philox) on the 84 configurations (72 random + 12 corner cases) of tests/golden/fuzz_cases.json
(the same configurations on which the oracle's MT back-end is pinned to the reference)."""
import json
import os

import numpy as np
import pytest

import oracle_py as O
from helpers import product_run

pytestmark = pytest.mark.gpu


def _certain_retirement(c):
    """A driver whose per-lap DNF probability is >= 1: the register kernel (32-bit thresholds) hands the problem to
    the generic kernel (csrc/race_kernel_reg.hip.h: reg_kernel_serves)."""
    teams, rates = c['config']['driver_teams'], c['config']['dnf_rates']
    ddr = c.get('driver_dnf_rates') or {}
    return any(ddr.get(d, rates.get(teams.get(d, 'Unknown'), 0.002)) >= 1.0 for d in c['grid_probs'])


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
            assert k.startswith('mcgp::race_kernel_reg<') != _certain_retirement(c), (name, k)
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
