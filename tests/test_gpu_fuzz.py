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
