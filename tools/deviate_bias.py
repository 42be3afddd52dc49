#!/usr/bin/env python3
"""What the product's 32-bit deviates change, measured by COMMON RANDOM NUMBERS on the CPU oracle.

The reference draws 53-bit uniforms and exact normals (reference src/simulation.py:137,194,302,330,524);
the HIP kernel and the oracle's Philox back-end draw uniforms w / 2^32 and normals from a binary32 cubic
table.  Comparing either with the reference's Mersenne-Twister runs only resolves +-0.5 pp.  Here both
precisions run on the SAME Philox words: back-end PHILOX (what the GPU computes, bit for bit) against
PHILOX53 (every uniform refined by 21 more bits, every normal the binary64 inverse CDF at that refined
point; oracle/mcgp_oracle.c).  A simulation whose finishing order is the same under both is unaffected by
the substitution; the rest bound the histogram change.

    python tools/deviate_bias.py [--sims 10000000] [--threads 8] [--cases S60 S78] [--out profiles/r3_deviate_bias.txt]

Test infrastructure (uses oracle/ only); nothing here touches the GPU.
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_py as O  # noqa: E402


def run_case(name, n_sims, threads, seed, chunk=50_000):
    case = O.load_case(name)
    n = len(case['grid_probs'])
    offsets = list(range(0, n_sims, chunk))
    problems = [O.Problem(case) for _ in range(threads)]

    def work(k):
        P = problems[k]
        h32 = np.zeros((n, n), np.int64)
        h53 = np.zeros((n, n), np.int64)
        differ = first = 0
        examples = []
        for off in offsets[k::threads]:
            m = min(chunk, n_sims - off)
            a = P.run(m, rng=O.RNG_PHILOX, seed=seed, sim_offset=off, want_orders=True)
            b = P.run(m, rng=O.RNG_PHILOX53, seed=seed, sim_offset=off, want_orders=True)
            h32 += a['hist']
            h53 += b['hist']
            d = (a['orders'] != b['orders'])
            rows = np.nonzero(d.any(axis=1))[0]
            differ += len(rows)
            first += int(d[:, 0].sum())
            for r in rows[:3]:
                if len(examples) < 3:
                    examples.append((off + int(r), int(d[r].sum())))
        return h32, h53, differ, first, examples

    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:               # ctypes releases the GIL
        parts = list(ex.map(work, range(threads)))
    dt = time.time() - t0
    h32 = sum(p[0] for p in parts)
    h53 = sum(p[1] for p in parts)
    differ = sum(p[2] for p in parts)
    first = sum(p[3] for p in parts)
    examples = [e for p in parts for e in p[4]][:6]
    return dict(name=name, n=n, n_sims=n_sims, seed=seed, h32=h32, h53=h53, differ=differ, winner_differs=first,
                examples=examples, seconds=dt)


def report(r):
    N = r['n_sims']
    d = r['h53'] - r['h32']
    lines = [f"== {r['name']}: {N} simulations, seed {r['seed']}, both back-ends on the same Philox words "
             f"({r['seconds']:.0f} s)"]
    lines.append(f"simulations whose finishing order differs: {r['differ']}  ({r['differ'] / N:.3e} of all)")
    lines.append(f"simulations whose WINNER differs:           {r['winner_differs']}  ({r['winner_differs'] / N:.3e})")
    lines.append(f"histogram cells that differ: {int((d != 0).sum())} of {d.size}; "
                 f"max |count delta| {int(np.abs(d).max())} = {np.abs(d).max() / N:.3e} in probability; "
                 f"sum |delta| / 2N = {np.abs(d).sum() / 2 / N:.3e}")
    win = d[:, 0]
    lines.append(f"win-probability delta per driver (53-bit minus 32-bit), max |.| = {np.abs(win).max() / N:.3e}: "
                 + ' '.join(f'{x / N:+.1e}' for x in win))
    # the delta any histogram cell can take is bounded by the differing simulations; compare with sampling noise
    p = np.maximum(r['h32'] / N, 1e-12)
    se = np.sqrt(p * (1 - p) / N)
    lines.append(f"largest |delta| in units of the binomial standard error of a {N}-simulation run: "
                 f"{float(np.max(np.abs(d) / N / se)):.3f}")
    if r['examples']:
        lines.append('first differing simulation ids (id, positions changed): ' + ', '.join(map(str, r['examples'])))
    return '\n'.join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sims', type=int, default=10_000_000)
    ap.add_argument('--threads', type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument('--cases', nargs='+', default=['S60', 'S78'])
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    O.lib()
    head = ('# Effect of the 32-bit deviates on results, by common random numbers (tools/deviate_bias.py):\n'
            '# oracle back-end PHILOX (= the HIP kernel, bit for bit) vs PHILOX53 (53-bit uniforms, binary64 inverse\n'
            '# normal CDF) on the same (seed, simulation id, lap, purpose, index) words.\n')
    out = [head]
    for name in a.cases:
        r = run_case(name, a.sims, a.threads, a.seed)
        text = report(r)
        print(text, flush=True)
        out.append(text + '\n')
    if a.out:
        with open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, 'w') as f:
            f.write('\n'.join(out))


if __name__ == '__main__':
    main()
