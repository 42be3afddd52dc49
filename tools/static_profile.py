#!/usr/bin/env python3
"""Static instruction attribution of the register kernel: VALU / SALU / LDS instruction counts per source region.

    hipcc ... -gline-tables-only -S --cuda-device-only -o reg20_lines.s reg_inst.hip
    python tools/static_profile.py reg20_lines.s

Each machine instruction is attributed to the innermost race_kernel_reg.hip.h line of its .loc (inlined callees in
race_common.hip.h are attributed through the inlined-at chain when present, else to their own file) and the lines
are grouped into the kernel's sections.  Counts are STATIC (one per instruction in the listing): rolled loops
(the RNG pre-pass, the overtake pass loop) count once.  Not product code.
"""
import collections
import re
import sys

SECTIONS = [            # (first line, label) in race_kernel_reg.hip.h, ascending
    (0, 'prologue / tables'),
]


def main():
    path = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else 'monte_carlo_gp_amd/csrc/race_kernel_reg.hip.h'
    marks = []
    for i, line in enumerate(open(src), 1):
        m = re.search(r'// @section (.+)$', line)
        if m:
            marks.append((i, m.group(1).strip()))
    files = {}
    cur = (None, 0)
    counts = collections.defaultdict(lambda: collections.Counter())
    ops = collections.defaultdict(lambda: collections.Counter())
    for line in open(path):
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
            continue
        m = re.match(r'(v_|s_|ds_|global_|buffer_|flat_)(\w+)', s)
        if not m:
            continue
        op = m.group(0)
        kind = 'VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'LDS' if op.startswith('ds_') else 'VMEM'
        if op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_barrier', 's_endpgm')):
            kind = 'CTRL'
        f, ln = cur
        if f and f.endswith('race_kernel_reg.hip.h'):
            sec = 'before first mark'
            for first, label in marks:
                if ln >= first:
                    sec = label
        else:
            sec = f'[{f}]'
        counts[sec][kind] += 1
        ops[sec][op.split('_e32')[0].split('_e64')[0]] += 1
    tot = collections.Counter()
    print(f'{"section":58s} {"VALU":>6s} {"SALU":>6s} {"LDS":>5s} {"CTRL":>5s}')
    for sec, c in counts.items():
        print(f'{sec:58s} {c["VALU"]:6d} {c["SALU"]:6d} {c["LDS"]:5d} {c["CTRL"]:5d}')
        tot.update(c)
    print(f'{"TOTAL":58s} {tot["VALU"]:6d} {tot["SALU"]:6d} {tot["LDS"]:5d} {tot["CTRL"]:5d}')
    if '-v' in sys.argv:
        for sec, c in ops.items():
            print('\n##', sec)
            print('   ' + ', '.join(f'{k} {v}' for k, v in c.most_common(14)))


if __name__ == '__main__':
    main()
