#!/usr/bin/env python3
"""Instruction-class mix of the register kernel's lap loop (N = 20) and the VALU issue ceiling it implies.

    TAG=r5 python tools/valu_mix.py [listing.s]      (default: compiles reg_inst.hip with -gline-tables-only)   -> profiles/<TAG>_valu_mix.json

Classes and their cost come from tools/valu_peak.hip (profiles/<TAG>_valu_peak.json if that round re-took it, else profiles/r3_valu_peak.json; wall-clock measurements; the 4-waves-per-SIMD column is used: 2, 4 and 8 agree within 3 %): binary64 operations and every VOP3-encoded (`_e64`, three-operand or SGPR-mask) or 64-bit integer
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
    tag = os.environ.get('TAG', 'r5')
    peak_file = f'{tag}_valu_peak.json' if os.path.exists(os.path.join(ROOT, 'profiles', f'{tag}_valu_peak.json')) else 'r3_valu_peak.json'
    path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/reg20_lines.s'
    if len(sys.argv) <= 1:
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off',
                               '-fno-fast-math', '-gline-tables-only', '-DMCGP_INST_N=20', '-S', '--cuda-device-only',
                               '-o', path, os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc', 'reg_inst.hip')],
                              stderr=subprocess.DEVNULL)
    src = open(SRC).read().split('\n')
    find = lambda text: next(i for i, l in enumerate(src, 1) if text in l)
    lo, hi = find('for (int lap = 2;'), find('// ================= classification')
    # rarely executed general path inside the loop (a lane with more than eight attempts in a pass), by source text
    g0 = find('// A wave with a lane that has more than eight attempts')
    g1 = next(i for i in range(g0, hi) if src[i - 1].startswith('                // ---- overtakes: success test'))
    # ... and by function: the exact tie-aware sort and its order tests (the fallback of both sorts)
    cold_fn = set()
    for i, l in enumerate(src, 1):
        if re.search(r'void transposition_sort\(|bool in_order\(|bool even_pairs_in_order\(|bool ties_in_order\(|^__device__ __forceinline__ bool cmpx\(', l):
            j = next(k for k in range(i, len(src)) if src[k - 1] == '}')
            cold_fn.update(range(i, j + 1))
    # the listing holds three kernels (default, reference-width, batch): the default one only
    text = open(path).read()
    funcs = re.split(r'\n(?=_ZN4mcgp\w+:)', text)
    body = next(f for f in funcs if f.startswith('_ZN4mcgp15race_kernel_regILi20'))      # (the first one: the default block shape)
    cur, cold, classes, ops, fine = None, False, collections.Counter(), collections.Counter(), collections.Counter()
    for line in body.split('\n'):
        s = line.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            chain = [int(x) for x in re.findall(r'race_kernel_reg\.hip\.h:(\d+):', s)]
            cur = chain[-2] if len(chain) >= 2 else int(m.group(2))
            cold = any(x in cold_fn for x in chain)
            continue
        m = re.match(r'(v_\w+)', s)
        if not m or cur is None or not (lo <= cur < hi) or cold or g0 <= cur < g1:
            continue
        op = m.group(1)
        four = (('f64' in op or op.endswith('_e64') or 'u64' in op or 'b64' in op or 'i64' in op or
                 re.match(r'v_(bfe|lshl_add|lshl_or|and_or|or3|xor3|add3|mad_|bfi|alignbit|perm|med3|min3|max3|cndmask_b32_e64|readlane|writelane|mbcnt)', op) is not None)
                and not op.endswith('_e32'))
        classes['4-cycle class' if four else '2-cycle class'] += 1
        fine['compare' if op.startswith('v_cmp') else 'convert64' if re.match(r'v_cvt_(f64_|u32_f64|i32_f64)', op) else
             'mad_u64' if op.startswith('v_mad_u64') else 'other 4-cycle' if four else '2-cycle'] += 1
        ops[op] += 1
    n4, n2 = classes['4-cycle class'], classes['2-cycle class']
    peak = json.load(open(os.path.join(ROOT, 'profiles', peak_file)))['waves_per_simd']['4']
    c4 = sum(peak[k] for k in ('v_add_f64', 'v_mul_f64', 'v_min_f64', 'v_bfe_u32', 'v_and_or_b32', 'v_cndmask_b32_e64 (SGPR mask)')) / 6
    c2 = sum(peak[k] for k in ('v_and_b32', 'v_xor_b32', 'v_add_u32')) / 3
    mean = (n4 * c4 + n2 * c2) / (n4 + n2)
    out = dict(static_valu_in_lap_loop=n4 + n2, four_cycle_class=n4, two_cycle_class=n2, share_four_cycle=n4 / (n4 + n2),
               cycles_four_cycle_class=c4, cycles_two_cycle_class=c2,
               mean_cycles_per_instruction_for_this_mix=mean,
               peak_T_wave_instructions_per_s_for_this_mix=1024 * 2.4e9 / mean / 1e12,
               top_opcodes=ops.most_common(14))
    # The same mix at the kernel's OWN occupancy (N = 20: one block of 768 threads, 3 waves per SIMD), where the measured costs
    # are higher for several classes: a VOP2 instruction 2.85 cycles instead of 2.25, a compare 5.4, a 64-bit conversion 5.1,
    # v_mad_u64_u32 5.65 (profiles/r3_valu_peak.json, column "3 (one block of 768)").
    p3 = json.load(open(os.path.join(ROOT, 'profiles', peak_file)))['waves_per_simd']['3 (one block of 768)']
    cost3 = {'compare': (p3['v_cmp_gt_f64 (vcc)'] + p3['v_cmp_lt_u32 (vcc)']) / 2, 'convert64': p3['v_cvt_f64_u32'],
             'mad_u64': p3['v_mad_u64_u32'],
             'other 4-cycle': sum(p3[k] for k in ('v_add_f64', 'v_mul_f64', 'v_min_f64', 'v_bfe_u32', 'v_and_or_b32', 'v_cndmask_b32_e64 (SGPR mask)')) / 6,
             '2-cycle': sum(p3[k] for k in ('v_and_b32', 'v_xor_b32', 'v_add_u32')) / 3}
    mean3 = sum(fine[k] * cost3[k] for k in fine) / sum(fine.values())
    out['at_three_waves_per_simd'] = dict(classes=dict(fine), cycles=cost3, mean_cycles_per_instruction=mean3,
                                          peak_T_wave_instructions_per_s=1024 * 2.4e9 / mean3 / 1e12)
    sys.path.insert(0, ROOT)
    from monte_carlo_gp_amd import _native as N
    out['source_hash'] = N.source_hash()
    out['costs_from'] = f'profiles/{peak_file} (tools/valu_peak.hip), 4 waves per SIMD'
    with open(os.path.join(ROOT, 'profiles', f'{tag}_valu_mix.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
