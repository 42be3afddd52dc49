#!/usr/bin/env python3
"""Generate csrc/sort_networks.h: the comparator lists of the register kernel's full sort for the field sizes where a
composed network beats Batcher's merge exchange (the kernel's default, generated in place for every other size).

Construction: the two halves of the field are sorted by the size-optimal networks for up to 10 inputs (1, 3, 5, 9, 12,
16, 19, 25, 29 comparators: Knuth, TAOCP 5.3.4; larger halves recursively) and merged by Batcher's odd-even merge in its
general (m, n) form: merge the odd-numbered elements of the two runs, merge the even-numbered ones, then compare-exchange
w_i : v_{i+1}.  With runs of odd length the merged sequence does not come out along the concatenated wire order, so the
merge returns the order its output IS sorted along and the wires of the whole network are renamed at the end so that
this order becomes 0, 1, .. n-1 -- a renaming costs nothing in a register array; what it leaves behind is a few
comparators that put their minimum on the higher-numbered wire, which the kernel's compare-exchange does not mind.

    n                  9   10   17   18   19   20   21   22   25   26
    merge exchange    26   31   74   82   91   97  107  114  138  146
    composed          25   29   73   80   88   93  105  112  137  144

Every network is checked here on ALL 2^n zero-one inputs (bit-parallel: one bit vector of 2^n bits per wire, a
comparator is an AND and an OR), which proves it sorts (zero-one principle); tests/test_host.py repeats the check on the
lists parsed from the header.  Comparators are written in layers (as soon as both wires are free), the order the kernel
issues them in.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OPTIMAL = {
    1: [],
    2: [[(0, 1)]],
    3: [[(0, 2)], [(0, 1)], [(1, 2)]],
    4: [[(0, 2), (1, 3)], [(0, 1), (2, 3)], [(1, 2)]],
    5: [[(0, 3), (1, 4)], [(0, 2), (1, 3)], [(0, 1), (2, 4)], [(1, 2), (3, 4)], [(2, 3)]],
    6: [[(0, 5), (1, 3), (2, 4)], [(1, 2), (3, 4)], [(0, 3), (2, 5)], [(0, 1), (2, 3), (4, 5)], [(1, 2), (3, 4)]],
    7: [[(0, 6), (2, 3), (4, 5)], [(0, 2), (1, 4), (3, 6)], [(0, 1), (2, 5), (3, 4)], [(1, 2), (4, 6)], [(2, 3), (4, 5)],
        [(1, 2), (3, 4), (5, 6)]],
    8: [[(0, 2), (1, 3), (4, 6), (5, 7)], [(0, 4), (1, 5), (2, 6), (3, 7)], [(0, 1), (2, 3), (4, 5), (6, 7)], [(2, 4), (3, 5)],
        [(1, 4), (3, 6)], [(1, 2), (3, 4), (5, 6)]],
    9: [[(0, 3), (1, 7), (2, 5), (4, 8)], [(0, 7), (2, 4), (3, 8), (5, 6)], [(0, 2), (1, 3), (4, 5), (7, 8)],
        [(1, 4), (3, 6), (5, 7)], [(0, 1), (2, 4), (3, 5), (6, 8)], [(2, 3), (4, 5), (6, 7)], [(1, 2), (3, 4), (5, 6)]],
    10: [[(4, 9), (3, 8), (2, 7), (1, 6), (0, 5)], [(1, 4), (6, 9), (0, 3), (5, 8)], [(0, 2), (3, 6), (7, 9)],
         [(0, 1), (2, 4), (5, 7), (8, 9)], [(1, 2), (4, 6), (7, 8), (3, 5)], [(2, 5), (6, 8), (1, 3), (4, 7)], [(2, 3), (6, 7)],
         [(3, 4), (5, 6)], [(4, 5)]],
}
# size of the first half for n > 10 (the split that minimises the comparator count of this construction)
SPLIT = {11: 3, 12: 4, 13: 5, 14: 6, 15: 7, 16: 8, 17: 8, 18: 8, 19: 9, 20: 10, 21: 10, 22: 10, 23: 7, 24: 8, 25: 9, 26: 10,
         27: 11, 28: 12, 29: 13, 30: 14, 31: 15, 32: 16}
SIZES = (9, 10, 17, 18, 19, 20, 21, 22, 25, 26)        # where the composed network is the smaller one


def merge(a, b, out):
    """a, b: wire lists, each sorted along the list.  Appends comparators (minimum to the first wire) and returns the wire
    list the union is sorted along afterwards."""
    if not a:
        return list(b)
    if not b:
        return list(a)
    if len(a) == 1 and len(b) == 1:
        out.append((a[0], b[0]))
        return [a[0], b[0]]
    v = merge(a[0::2], b[0::2], out)
    w = merge(a[1::2], b[1::2], out)
    z = [v[0]]
    for i in range(len(w)):
        if i + 1 < len(v):
            out.append((w[i], v[i + 1]))
            z += [w[i], v[i + 1]]
        else:
            z.append(w[i])
    z += v[len(w) + 1:]
    return z


def sorter(wires, out):
    n = len(wires)
    if n in OPTIMAL:
        out += [(wires[x], wires[y]) for layer in OPTIMAL[n] for x, y in layer]
        return list(wires)
    m = SPLIT[n]
    return merge(sorter(wires[:m], out), sorter(wires[m:], out), out)


def build(n):
    out = []
    order = sorter(list(range(n)), out)
    name = [0] * n
    for k, wire in enumerate(order):
        name[wire] = k
    net = [(name[x], name[y]) for x, y in out]
    # layers: a comparator goes into the first layer after the last use of either wire
    free = [0] * n
    layer = []
    for x, y in net:
        s = max(free[x], free[y])
        layer.append(s)
        free[x] = free[y] = s + 1
    idx = sorted(range(len(net)), key=lambda i: (layer[i], i))
    return [net[i] for i in idx], [layer[i] for i in idx]


def merge_exchange_size(n):
    t = 0
    while (1 << t) < n:
        t += 1
    count = 0
    p = 1 << (t - 1) if t > 0 else 0
    while p > 0:
        q, r, d = 1 << (t - 1), 0, p
        while True:
            count += sum(1 for i in range(n - d) if (i & p) == r)
            if q == p:
                break
            d, q, r = q - p, q >> 1, p
        p >>= 1
    return count


def sorts_all_zero_one_inputs(n, net):
    """Bit-parallel zero-one check: wire i starts as the bit vector (input j has bit i set), j = 0 .. 2^n - 1."""
    words = max(1, (1 << n) // 64)
    wires = []
    j = np.arange(words, dtype=np.uint64)
    for i in range(n):
        if i < 6:
            pattern = sum(1 << b for b in range(64) if (b >> i) & 1)
            if n < 6:
                pattern &= (1 << (1 << n)) - 1
            wires.append(np.full(words, pattern, dtype=np.uint64))
        else:
            wires.append(np.where((j >> np.uint64(i - 6)) & np.uint64(1), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)))
    for x, y in net:
        lo, hi = wires[x] & wires[y], wires[x] | wires[y]
        wires[x], wires[y] = lo, hi
    return all(not (wires[i] & ~wires[i + 1]).any() for i in range(n - 1))


def main():
    lines = ['// Generated by tools/gen_sort_networks.py -- do not edit.  Comparator lists (minimum to the first wire, in layers) of',
             '// the full sort for the field sizes where two size-optimal half sorters + Batcher\'s odd-even merge need fewer',
             '// comparators than merge exchange; every list is verified on all 2^n zero-one inputs by the generator and by',
             '// tests/test_host.py.', '#pragma once', '', 'namespace mcgp {', '',
             'struct SortNetworkTable {', '    int n, size;', '    const signed char (*pairs)[3];       // {first wire, second wire, layer}',
             '};', '']
    for n in SIZES:
        net, layer = build(n)
        assert len(net) < merge_exchange_size(n), n
        assert sorts_all_zero_one_inputs(n, net), n
        print(f'n = {n}: {len(net)} comparators in {max(layer) + 1} layers (merge exchange {merge_exchange_size(n)}): sorts all 2^{n} inputs')
        lines.append(f'constexpr signed char kSortNetwork{n}[{len(net)}][3] = {{')
        row = '   '
        for (x, y), s in zip(net, layer):
            item = f' {{{x}, {y}, {s}}},'
            if len(row) + len(item) > 118:
                lines.append(row)
                row = '   '
            row += item
        lines.append(row)
        lines.append('};')
    lines.append('constexpr SortNetworkTable kSortNetworks[] = {')
    for n in SIZES:
        lines.append(f'    {{{n}, (int)(sizeof(kSortNetwork{n}) / sizeof(kSortNetwork{n}[0])), kSortNetwork{n}}},')
    lines += ['};', '', '}  // namespace mcgp', '']
    path = os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc', 'sort_networks.h')
    with open(path, 'w') as f:
        f.write('\n'.join(lines))
    print('wrote', path)


if __name__ == '__main__':
    sys.exit(main())
