#!/usr/bin/env python3
"""Deep parity sweep on the GPU box: HIP kernel vs CPU oracle (Philox back-end), finishing order by finishing order,
at sizes well beyond the test suite's (rare paths: re-sort fallback, equal-time ties, event storms, mass retirements).

    python tools/deep_parity.py [sims per fuzz configuration] [sims per golden case]  > profiles/r2_deep_parity.txt
    DEVIATES=53 ...: the reference-width kernel against the oracle's PHILOX53 back-end (configurations the register kernel takes)

Test infrastructure (uses the oracle as the checker); prints one line per configuration and a summary."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from helpers import product_run  # noqa: E402


DEVIATES = int(os.environ.get('DEVIATES', '32'))
RNG = O.RNG_PHILOX53 if DEVIATES == 53 else O.RNG_PHILOX


def oracle_orders(case, n, seed, threads=16):
    chunk = (n + threads - 1) // threads

    def part(i):
        lo = i * chunk
        cnt = max(0, min(chunk, n - lo))
        if cnt == 0:
            return np.zeros((0, len(case['grid_probs'])), np.uint8)
        return O.Problem(case).run(cnt, rng=RNG, seed=seed, sim_offset=lo, want_orders=True)['orders']
    with ThreadPoolExecutor(threads) as ex:
        return np.vstack(list(ex.map(part, range(threads))))


def main():
    n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    n_gold = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    with open(os.path.join(O.GOLDEN_DIR, 'fuzz_cases.json')) as f:
        fuzz = json.load(f)
    jobs = [(name, O.load_case(name), 42, n_gold) for name in ('S60', 'S78', 'S50', 'EVT', 'HET', 'DMP', 'WET', 'N10')]
    jobs += [(name, c, c['seed'], n_fuzz) for name, c in fuzz.items()]
    bad_total, sims_total, t0 = 0, 0, time.time()
    print(f'deviates = {DEVIATES}', flush=True)
    for name, case, seed, n in jobs:
        if DEVIATES == 53 and case['config']['overtake_delta'] < 0:
            continue                                  # (generic-kernel problems have no reference-width build)
        hist, _, orders = product_run(case, n, seed, orders=True, deviates=DEVIATES)
        ref = oracle_orders(case, n, seed)
        bad = int((orders != ref).any(axis=1).sum())
        bad_total += bad
        sims_total += n
        print(f'{name:24s} n={len(case["grid_probs"]):2d} laps={case["config"]["total_laps"]:3d} sims={n:7d} differing finishing orders={bad}',
              flush=True)
    print(f'TOTAL {sims_total} simulations over {len(jobs)} configurations, {bad_total} differing finishing orders, '
          f'{time.time() - t0:.0f} s')
    return 1 if bad_total else 0


if __name__ == '__main__':
    sys.exit(main())
