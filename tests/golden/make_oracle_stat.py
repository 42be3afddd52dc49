#!/usr/bin/env python3
"""Big-N histograms of the oracle's MT back-end -- the reference-equivalent path (tests/test_oracle_golden.py:
identical to the reference's own runs, integer for integer) -- for the statistical link of tests/test_gpu_scale.py.

    python tests/golden/make_oracle_stat.py [S60 S78 HET EVT]      -> tests/golden/oracle_mt_stat_<case>.npz

Each file: `hist` = summed n x n integer histogram over `seeds` (one MT stream pair per seed, exactly what
RaceSimulator.run_monte_carlo(n_each, seed=s) of the reference would return, reference src/simulation.py:59-100),
`per_seed`, `seeds`, `n_each`.  2x10^7 simulations for S60 / S78 (BASELINE configs[1], [2]), 10^7 for HET / EVT:
a win probability near 0.5 then carries a standard error of 1.1x10^-4 / 1.6x10^-4, against 1.1x10^-3 for the 2x10^5
reference simulations the link used to rest on.  Runs on the host cores only (ctypes releases the GIL: one thread
per core), about twenty minutes on 8 cores.  Test infrastructure; the script that made the committed files."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as O  # noqa: E402

PLAN = {'S60': (32, 625_000), 'S78': (32, 625_000), 'HET': (16, 625_000), 'EVT': (16, 625_000)}
FIRST_SEED = {'S60': 6000, 'S78': 7000, 'HET': 8000, 'EVT': 9000}


def main():
    names = sys.argv[1:] or list(PLAN)
    threads = int(os.environ.get('MCGP_STAT_THREADS', str(os.cpu_count() or 1)))
    for name in names:
        n_seeds, n_each = PLAN[name]
        case = O.load_case(name)
        seeds = [FIRST_SEED[name] + i for i in range(n_seeds)]

        def one(seed):
            return O.Problem(case).run(n_each, rng=O.RNG_MT, seed=seed)['hist']
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:
            hs = list(ex.map(one, seeds))
        np.savez_compressed(os.path.join(HERE, f'oracle_mt_stat_{name}.npz'), hist=np.sum(hs, axis=0),
                            per_seed=np.array(hs), seeds=np.array(seeds), n_each=n_each)
        print(f'{name}: {n_seeds} x {n_each} in {time.time() - t0:.0f} s', flush=True)


if __name__ == '__main__':
    main()
