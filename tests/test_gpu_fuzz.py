"""HIP vs oracle (Philox) on the 84 configurations (72 random + 12 corner cases) of tests/golden/fuzz_cases.json
(the same configurations on which the oracle's MT back-end is pinned to the reference)."""
import json

import numpy as np
import pytest

import oracle_py as O
from helpers import product_run

pytestmark = pytest.mark.gpu


def test_fuzzed_configurations_match_oracle(require_gpu):
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        cases = json.load(f)
    kernels = set()
    for name, c in cases.items():
        n_sims = 1500
        ref = O.Problem(c).run(n_sims, rng=O.RNG_PHILOX, seed=c['seed'], want_orders=True)
        hist, _, orders = product_run(c, n_sims, c['seed'], orders=True)
        bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
        assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
        assert np.array_equal(hist, ref['hist']), name
        from monte_carlo_gp_amd import _native as N
        kernels.add(N.lib().mcgp_last_kernel_name(0).decode())
    # both kernel families were exercised
    assert 'mcgp::race_kernel' in kernels and any(k.startswith('mcgp::race_kernel_reg<') for k in kernels)
