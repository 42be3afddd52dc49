#!/usr/bin/env python3
"""Generate the inverse-normal-CDF table used by the Philox back-end.

The counter-based path turns ONE 32-bit Philox word w into a standard normal
deviate with a branch-free piecewise cubic (no log / sqrt / division, so the
CPU oracle and the HIP kernel produce the same bits by construction):

    sign = w >> 31,  m = w & 0x7fffffff,  tail probability p = (m + 0.5) / 2^32 in (0, 0.5)
    mm = m + 16                   (16 .. 2^31 + 15: no special case for the smallest m)
    f  = float(mm)                (binary32, round to nearest even -- v_cvt_f32_u32 / a C cast), b = its bit pattern
    e  = exponent of f (4 .. 31), k = its four leading mantissa bits, the other 19 = the offset inside the cell
    off = (b >> 15) & 0x1ff0      = 16 bytes x (16 (E mod 32) + k),  E = e + 127 the biased exponent: E mod 32 = 3 .. 30
    t  = float(b & 0x7ffff)                               = (cell coordinate in [0, 1)) x 2^19
    z0 = fmaf(fmaf(fmaf(c3, t, c2), t, c1), t, c0)       (binary32; the row's coefficients carry the 2^-19 per power)
    z  = bits(z0) ^ (w & 0x80000000)                      (z0 = Phi^-1(p) < 0; the sign of w flips it)

The float conversion does what a count-leading-zeros, three shifts, two shift-and-merge operations and a conversion did
in rounds 1-4 (seven instructions for eleven on the device).  Cell (e, k) holds the f with f in
[2^e (1 + k/16), 2^e (1 + (k+1)/16)); up to 2^24 every mm is its own f, above that f is mm rounded to 24 bits (a
relative 2^-24 in the tail probability: 1e-7 in the deviate at most).  Every row holds the cubic that interpolates
Phi^-1((f - 15.5) / 2^32) at the four Chebyshev nodes of the cell, rounded to binary32; the 16 cells of width one
(e = 4: the 16 smallest m) hold the constant.  Rows are stored in the order of `off`, 16 (e - 4) + k, and addressed
through a base that lies 48 rows (kNormalRowBias) before the table: 448 rows x 4 coefficients = 7168 B.

Writes TWO byte-identical copies of the table (bit patterns as uint32):
  oracle/normal_table.h                      (test infrastructure)
  monte_carlo_gp_amd/csrc/normal_table.h     (product)
The product never includes anything from oracle/; tests assert that both
copies carry the same words.
"""
import os
import sys

import numpy as np
from scipy.special import ndtri

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NROWS = 28 * 16


ROW_BIAS = 48                      # (E mod 32) * 16 of the first exponent, e = 4: E = 131


def rows():
    tab = np.zeros((NROWS, 4), np.float64)
    # Chebyshev nodes on [0, 1]
    j = np.arange(4)
    nodes = 0.5 - 0.5 * np.cos((2 * j + 1) * np.pi / 8)
    V = np.vander(nodes, 4, increasing=True)
    for e in range(4, 32):
        width = 2.0 ** (e - 4)
        for k in range(16):
            lo = 2.0 ** e * (1 + k / 16.0)
            r = 16 * (e - 4) + k
            if e == 4:
                tab[r, 0] = ndtri((lo - 15.5) / 2.0 ** 32)                    # one m per cell: the constant
                continue
            p = (lo + nodes * width - 15.5) / 2.0 ** 32
            coef = np.linalg.solve(V, ndtri(p))                               # in the cell coordinate
            tab[r] = coef * 2.0 ** (-19.0 * np.arange(4))                     # in t = coordinate x 2^19
    return tab.astype(np.float32)


def eval_table(tab, w):
    """numpy model of the transform (float64 emulation of the binary32 fma chain)."""
    w = w.astype(np.uint64)
    sign = (w >> np.uint64(31)) & np.uint64(1)
    m = (w & np.uint64(0x7fffffff)).astype(np.int64)
    mm = (m + 16).astype(np.uint32)
    b = mm.astype(np.float32).view(np.uint32).astype(np.int64)               # round to nearest even
    off = (b >> 15) & 0x1ff0
    t = (b & 0x7ffff).astype(np.float32)
    cf = tab[(off >> 4) - ROW_BIAS]
    f = lambda a, b_, cc: (a.astype(np.float64) * b_.astype(np.float64) + cc.astype(np.float64)).astype(np.float32)
    z = f(f(f(cf[:, 3], t, cf[:, 2]), t, cf[:, 1]), t, cf[:, 0])
    return np.where(sign == 1, -z, z), (m + 0.5) / 2.0 ** 32, sign


MAX_ABS_ERR = 4.8e-7               # stated accuracy of the transform; main() asserts it


def every_cell_words(points=9):
    """Draw words that visit every one of the 448 cells at `points` places (first, last, evenly spaced between), both signs."""
    out = []
    for e in range(4, 32):
        for k in range(16):
            lo = (16 + k) << (e - 4)                   # mm = m + 16 at the start of the cell
            width = 1 << (e - 4)
            for j in range(points):
                mm = lo + min(width - 1, (width * j) // (points - 1))
                out += [mm - 16, (mm - 16) | 0x80000000]
    return np.array(sorted(set(out)), dtype=np.uint64)


def main():
    tab = rows()
    rng = np.random.default_rng(1)
    w = np.concatenate([rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint64),
                        np.arange(0, 4096, dtype=np.uint64),
                        (np.uint64(1) << np.arange(0, 32, dtype=np.uint64)),
                        (np.uint64(1) << np.arange(1, 32, dtype=np.uint64)) - np.uint64(1),
                        (np.uint64(1) << np.arange(5, 32, dtype=np.uint64)) - np.uint64(17),
                        np.uint64(0x7fffffff) - np.arange(0, 64, dtype=np.uint64)])
    z, p, sign = eval_table(tab, w)
    exact = np.where(sign == 1, -ndtri(p), ndtri(p))
    err = np.abs(z - exact)
    print(f'rows={NROWS} max abs err={err.max():.3e} max rel err={np.max(err / np.maximum(np.abs(exact), 1e-3)):.3e}',
          file=sys.stderr)
    # the bound the documentation states (bench.py dtype_note) is ASSERTED, over random words, the special
    # words above and every cell of the table at nine points and both signs
    cells = every_cell_words()
    zc, pc, sc = eval_table(tab, cells)
    cell_err = np.abs(zc - np.where(sc == 1, -ndtri(pc), ndtri(pc)))
    print(f'every cell x 9 points x 2 signs: max abs err={cell_err.max():.3e}', file=sys.stderr)
    assert max(err.max(), cell_err.max()) <= MAX_ABS_ERR, (err.max(), cell_err.max())
    words = tab.view(np.uint32).reshape(-1)
    if '--check' in sys.argv[1:]:
        # regenerate, assert the bound (above) and compare with the committed copies; nothing is written
        import re
        for path in ('oracle/normal_table.h', 'monte_carlo_gp_amd/csrc/normal_table.h'):
            with open(os.path.join(ROOT, path)) as f:
                have = [int(x[:-1], 16) for x in re.findall(r'0x[0-9a-f]{8}u', f.read())]
            assert have == [int(x) for x in words], f'{path} is not what this generator writes'
        return 0
    body = ',\n'.join('  ' + ', '.join(f'0x{x:08x}u' for x in words[i:i + 8]) for i in range(0, len(words), 8))
    for path, guard, who in (
            ('oracle/normal_table.h', 'MCGP_ORACLE_NORMAL_TABLE_H', 'oracle (test infrastructure)'),
            ('monte_carlo_gp_amd/csrc/normal_table.h', 'MCGP_NORMAL_TABLE_H', 'product')):
        full = os.path.join(ROOT, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, 'w') as f:
            f.write(f'/* GENERATED by tools/gen_normal_table.py -- do not edit.  Copy for: {who}.\n'
                    f' * Piecewise-cubic inverse normal CDF, {NROWS} rows (16 (exponent - 4) + k) x 4 binary32 coefficients (c0..c3),\n'
                    f' * stored as IEEE-754 bit patterns.  See the generator for the row/t mapping. */\n'
                    f'#ifndef {guard}\n#define {guard}\n'
                    f'#define MCGP_NORMAL_ROWS {NROWS}\n'
                    f'static const unsigned int mcgp_normal_table_bits[{NROWS * 4}] = {{\n{body}\n}};\n#endif\n')
    return 0


if __name__ == '__main__':
    sys.exit(main())
