#!/usr/bin/env python3
"""Generate the binary64 inverse-normal-CDF table of the reference-width (53-bit) deviates.

The reference draws exact normals (reference src/simulation.py:302,330: numpy's polar method on 53-bit uniforms); the
product's default deviates carry 32 bits (tools/gen_normal_table.py).  The `deviates = 53` mode of the library and the
oracle's PHILOX53 back-end turn a 52-bit tail-probability index q (the 31 magnitude bits m of the draw's word w followed
by 21 bits e of its companion word) and the sign bit of w into a normal deviate with a branch-free piecewise polynomial
of degree 7 in binary64 (explicit fma only: the CPU oracle and the HIP kernel produce the same bits by construction):

    sign = w >> 31,  q = m << 21 | e  in [0, 2^52),  tail probability p = (q + 0.5) / 2^53 in (0, 0.5)
    q < 16 :  row = q,                 t = 0
    else   :  hb = floor(log2 q) (4..51), sh = hb - 4, k = (q >> sh) & 15,
              r = q & (2^sh - 1),      t = (double(r) + 0.5) * 2^-sh,     row = 16 + 16*sh + k
    z0 = fma(..fma(fma(c7, t, c6), t, c5).., t, c0)        (binary64, row coefficients)
    z  = sign ? -z0 : z0                                    (z0 = Phi^-1(p) < 0)

Every row holds the polynomial that interpolates Phi^-1 (scipy.special.ndtri) at the eight Chebyshev nodes of its
sub-interval.  784 rows x 8 coefficients = 50 176 B.  Writes two byte-identical copies (bit patterns as uint64):
  oracle/normal53_table.h                      (test infrastructure)
  monte_carlo_gp_amd/csrc/normal53_table.h     (product: uploaded to device memory once per device)
"""
import os
import sys

import numpy as np
from scipy.special import ndtri

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEG = 7
NROWS = 16 + 48 * 16


def rows():
    tab = np.zeros((NROWS, DEG + 1), np.float64)
    for q in range(16):
        tab[q, 0] = ndtri((q + 0.5) / 2.0 ** 53)
    j = np.arange(DEG + 1)
    nodes = 0.5 - 0.5 * np.cos((2 * j + 1) * np.pi / (2 * (DEG + 1)))
    V = np.vander(nodes, DEG + 1, increasing=True)
    for sh in range(48):
        for k in range(16):
            q_lo = float((16 + k) << sh)
            p = (q_lo + nodes * 2.0 ** sh) / 2.0 ** 53
            tab[16 + 16 * sh + k] = np.linalg.solve(V, ndtri(p))
    return tab


def eval_table(tab, q):
    """numpy model of the transform on tail indices q (uint64): plain binary64 Horner (the C versions use fma; the
    difference is below the last digit shown by the check)."""
    q = q.astype(np.int64)
    small = q < 16
    qq = np.where(small, 16, q)
    hb = np.floor(np.log2(qq.astype(np.float64))).astype(np.int64)
    hb = np.where((1 << hb) > qq, hb - 1, hb)
    sh = hb - 4
    k = (qq >> sh) & 15
    r = qq & ((1 << sh) - 1)
    t = np.where(small, 0.0, (r.astype(np.float64) + 0.5) * 2.0 ** (-sh.astype(np.float64)))
    row = np.where(small, q, 16 + 16 * sh + k)
    c = tab[row]
    z = c[:, DEG]
    for d in range(DEG - 1, -1, -1):
        z = z * t + c[:, d]
    return z


MAX_ABS_ERR = 4e-15                # stated accuracy of the transform: 3.6e-15 measured in the far tail (|z| = 8.2, where an ulp of z is 1.8e-15), 2.7e-15 over random indices; main() asserts it


def every_cell_indices(points=9):
    """Tail indices q that visit every one of the 784 rows: the 16 single-index rows, and `points` places of every cell."""
    out = list(range(16))
    for sh in range(48):
        for k in range(16):
            lo, width = (16 + k) << sh, 1 << sh
            out += [lo + min(width - 1, (width * j) // (points - 1)) for j in range(points)]
    return np.array(sorted(set(out)), dtype=np.uint64)


def main():
    tab = rows()
    rng = np.random.default_rng(1)
    q = np.concatenate([rng.integers(0, 2 ** 52, 2_000_000, dtype=np.uint64), np.arange(0, 4096, dtype=np.uint64),
                        (np.uint64(1) << np.arange(0, 52, dtype=np.uint64)),
                        (np.uint64(1) << np.arange(1, 53, dtype=np.uint64)) - np.uint64(1)])
    z = eval_table(tab, q)
    exact = ndtri((q.astype(np.float64) + 0.5) / 2.0 ** 53)
    err = np.abs(z - exact)
    print(f'rows={NROWS} degree={DEG} max abs err={err.max():.3e} max rel err={np.max(err / np.abs(exact)):.3e}',
          file=sys.stderr)
    # the stated bound is ASSERTED: random indices, the special ones, and every cell at nine points
    qc = every_cell_indices()
    cell_err = np.abs(eval_table(tab, qc) - ndtri((qc.astype(np.float64) + 0.5) / 2.0 ** 53))
    print(f'every cell x 9 points: max abs err={cell_err.max():.3e}', file=sys.stderr)
    assert max(err.max(), cell_err.max()) <= MAX_ABS_ERR, (err.max(), cell_err.max())
    words = tab.view(np.uint64).reshape(-1)
    if '--check' in sys.argv[1:]:
        # regenerate, assert the bound (above) and compare with the committed copies; nothing is written
        import re
        for path in ('oracle/normal53_table.h', 'monte_carlo_gp_amd/csrc/normal53_table.h'):
            with open(os.path.join(ROOT, path)) as f:
                have = [int(x[:-3], 16) for x in re.findall(r'0x[0-9a-f]{16}ull', f.read())]
            assert have == [int(x) for x in words], f'{path} is not what this generator writes'
        return 0
    body = ',\n'.join('  ' + ', '.join(f'0x{x:016x}ull' for x in words[i:i + 4]) for i in range(0, len(words), 4))
    for path, guard, who in (
            ('oracle/normal53_table.h', 'MCGP_ORACLE_NORMAL53_TABLE_H', 'oracle (test infrastructure)'),
            ('monte_carlo_gp_amd/csrc/normal53_table.h', 'MCGP_NORMAL53_TABLE_H', 'product')):
        with open(os.path.join(ROOT, path), 'w') as f:
            f.write(f'/* GENERATED by tools/gen_normal53_table.py -- do not edit.  Copy for: {who}.\n'
                    f' * Piecewise inverse normal CDF in binary64, {NROWS} rows x {DEG + 1} coefficients (c0..c{DEG}),\n'
                    f' * stored as IEEE-754 bit patterns.  See the generator for the row/t mapping. */\n'
                    f'#ifndef {guard}\n#define {guard}\n'
                    f'#define MCGP_NORMAL53_ROWS {NROWS}\n#define MCGP_NORMAL53_COEFFS {DEG + 1}\n'
                    f'static const unsigned long long mcgp_normal53_table_bits[{NROWS * (DEG + 1)}] = {{\n{body}\n}};\n#endif\n')
    return 0


if __name__ == '__main__':
    sys.exit(main())
