#!/usr/bin/env python3
"""Instruction-class mix of the register kernel's lap loop (N = 20) and the VALU issue ceiling it implies.

    python tools/valu_mix.py [listing.s]      (default: compiles reg_inst.hip with -gline-tables-only)   -> profiles/r3_valu_mix.json

Classes and their cost come from tools/valu_peak.hip (profiles/r3_valu_peak.json, wall-clock measurements; the 4-waves-per-SIMD column is used: 2, 4 and 8 agree within 3 %): binary64 operations and every VOP3-encoded (`_e64`, three-operand or SGPR-mask) or 64-bit integer
instruction occupy a SIMD for ~4.15 cycles per wave64 instruction; VOP1 / VOP2 32-bit instructions (`_e32`) for ~2.2.
Weights are STATIC instruction counts of the lap loop (straight-line code for the most part; the rarely executed general
paths -- transposition sort, >8 attempts -- are left out by source line).  Not product code."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc', 'race_kernel_reg.hip.h')


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/reg20_lines.s'
    if len(sys.argv) <= 1:
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off',
                               '-fno-fast-math', '-gline-tables-only', '-DMCGP_INST_N=20', '-S', '--cuda-device-only',
                               '-o', path, os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc', 'reg_inst.hip')],
                              stderr=subprocess.DEVNULL)
    src = open(SRC).read().split('\n')
    lo = next(i for i, l in enumerate(src, 1) if 'for (int lap = 2;' in l)
    hi = next(i for i, l in enumerate(src, 1) if '// ================= classification' in l)
    # rarely executed general paths inside the loop, by source text
    cold = [i for i, l in enumerate(src, 1) if 'transposition_sort<N>(cum, pk)' in l or 'MCGP_ANY(n_cand > kWordRows)' in l]
    cold_ranges = []
    for i in cold:
        if 'MCGP_ANY' in src[i - 1]:
            j = next(k for k in range(i, hi) if src[k - 1].startswith('                } else {'))
            cold_ranges.append((i, j))
    # ... and by function: the exact tie-aware sort and its order tests (the fallback of both sorts)
    cold_fn = []
    for i, l in enumerate(src, 1):
        if re.search(r'void transposition_sort\(|bool in_order\(|bool even_pairs_in_order\(|bool ties_in_order\(', l):
            j = next(k for k in range(i, len(src)) if src[k - 1] == '}')
            cold_fn.append((i, j))
    in_cold = lambda n: any(a <= n <= b for a, b in cold_ranges)
    in_cold_fn = lambda n: any(a <= n <= b for a, b in cold_fn)
    cur, classes, ops = None, collections.Counter(), collections.Counter()
    for line in open(path):
        s = line.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            chain = re.findall(r'race_kernel_reg\.hip\.h:(\d+):', s)
            cur = int(chain[-2]) if len(chain) >= 2 else int(m.group(2))
            inner = int(m.group(2))
            continue
        m = re.match(r'(v_\w+)', s)
        if not m or cur is None or not (lo <= cur <= hi) or in_cold(cur) or in_cold_fn(inner):
            continue
        op = m.group(1)
        four = ('f64' in op or op.endswith('_e64') or 'u64' in op or 'b64' in op or 'i64' in op or
                re.match(r'v_(bfe|lshl_add|lshl_or|and_or|or3|xor3|add3|mad_|bfi|alignbit|perm|med3|min3|max3|cndmask_b32_e64|readlane|writelane|mbcnt)', op) is not None
                and not op.endswith('_e32'))
        classes['4-cycle class' if four else '2-cycle class'] += 1
        ops[op] += 1
    n4, n2 = classes['4-cycle class'], classes['2-cycle class']
    peak = json.load(open(os.path.join(ROOT, 'profiles', 'r3_valu_peak.json')))['waves_per_simd']['4']
    c4 = sum(peak[k] for k in ('v_add_f64', 'v_mul_f64', 'v_min_f64', 'v_bfe_u32', 'v_and_or_b32', 'v_cndmask_b32_e64 (SGPR mask)')) / 6
    c2 = sum(peak[k] for k in ('v_and_b32', 'v_xor_b32', 'v_add_u32')) / 3
    mean = (n4 * c4 + n2 * c2) / (n4 + n2)
    out = dict(static_valu_in_lap_loop=n4 + n2, four_cycle_class=n4, two_cycle_class=n2, share_four_cycle=n4 / (n4 + n2),
               cycles_four_cycle_class=c4, cycles_two_cycle_class=c2,
               mean_cycles_per_instruction_for_this_mix=mean,
               peak_T_wave_instructions_per_s_for_this_mix=1024 * 2.4e9 / mean / 1e12,
               top_opcodes=ops.most_common(14))
    sys.path.insert(0, ROOT)
    from monte_carlo_gp_amd import _native as N
    out['source_hash'] = N.source_hash()
    out['costs_from'] = 'profiles/r3_valu_peak.json (tools/valu_peak.hip), 4 waves per SIMD'
    with open(os.path.join(ROOT, 'profiles', 'r3_valu_mix.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
