#!/usr/bin/env python3
"""GPU histogram against the CPU oracle's, cell for cell, at a size the test suite cannot afford (GPU box; the oracle runs
on every host core of the job's share, in slices, with a progress line per slice).

    python tools/exact_hist.py [WORKLOAD] [N_SIMS] [SEED] [OFFSET]   -> stdout (profiles/r3_exact_hist.txt)

Test infrastructure: the oracle is the checker."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_py as O
from helpers import product_run
from monte_carlo_gp_amd import _native as N

name = sys.argv[1] if len(sys.argv) > 1 else 'S60'
n_sims = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
base = int(float(sys.argv[4])) if len(sys.argv) > 4 else 0
threads = int(os.environ.get('THREADS', '16'))
deviates = int(os.environ.get('DEVIATES', '32'))        # 53: the reference-width kernel against the oracle's PHILOX53 back-end
rng = O.RNG_PHILOX53 if deviates == 53 else O.RNG_PHILOX
case = O.load_case(name)
n = len(case['grid_probs'])
t0 = time.time()
gpu = product_run(case, n_sims, seed, sim_offset=base, deviates=deviates)[0]
print(f'{name}: {n_sims} simulations from id {base}, seed {seed}, deviates {deviates}, build {N.source_hash()}; GPU {time.time() - t0:.1f} s', flush=True)
problems = [O.Problem(case) for _ in range(threads)]
slice_sims, chunk = 5_000_000, 25_000
ref = np.zeros((n, n), np.int64)
for s0 in range(0, n_sims, slice_sims):
    cnt = min(slice_sims, n_sims - s0)
    offs = list(range(s0, s0 + cnt, chunk))

    def work(k):
        h = np.zeros((n, n), np.int64)
        for off in offs[k::threads]:
            h += problems[k].run(min(chunk, s0 + cnt - off), rng=rng, seed=seed, sim_offset=base + off)['hist']
        return h
    with ThreadPoolExecutor(threads) as ex:
        ref += sum(ex.map(work, range(threads)))
    # the slice itself, so that a difference is located early
    part = product_run(case, cnt, seed, sim_offset=base + s0, deviates=deviates)[0]
    whole = product_run(case, s0 + cnt, seed, sim_offset=base, deviates=deviates)[0]
    ok = np.array_equal(whole, ref)
    print(f'  oracle through {s0 + cnt:>11d}: {time.time() - t0:7.1f} s  histogram so far equal: {ok}', flush=True)
    if not ok:
        print('  DIFFERENCE in slice', s0, int(np.abs(whole - ref).sum()))
        sys.exit(1)
print(f'{name}: GPU histogram == oracle histogram, all {n * n} cells, {n_sims} simulations (oracle: {threads} threads, {time.time() - t0:.0f} s)')
