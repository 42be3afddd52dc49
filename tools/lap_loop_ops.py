#!/usr/bin/env python3
"""Opcode histogram of the lap loop of a register kernel (static; the three overtake passes are three copies).
    python tools/lap_loop_ops.py [N] [reg|batch|wide] [-D...]
Instructions are attributed to source lines of race_kernel_reg.hip.h through the .loc chain (the frame inside
reg_simulate); the lap loop is the line range from `laps 2..L` to `classification`.  Not product code."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc')
SRC = os.path.join(CSRC, 'race_kernel_reg.hip.h')

def main():
    args = [a for a in sys.argv[1:] if not a.startswith('-D')]
    defs = [a for a in sys.argv[1:] if a.startswith('-D')]
    n = args[0] if args else '20'
    which = args[1] if len(args) > 1 else 'reg'
    symbol = {'reg': 'race_kernel_regILi', 'batch': 'race_kernel_reg_batchILi', 'wide': 'race_kernel_reg_wideILi'}[which]
    out = f'/tmp/reg{n}_lines.s'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off',
                           '-fno-fast-math', '-gline-tables-only', f'-DMCGP_INST_N={n}', '-S', '--cuda-device-only',
                           '-o', out, os.path.join(CSRC, 'reg_inst.hip')] + defs, stderr=subprocess.DEVNULL)
    first = last = None
    sections = []
    for i, line in enumerate(open(SRC), 1):
        if '// ================= laps 2..L' in line: first = i
        if '// ================= classification' in line: last = i
        m = re.search(r'// ---- (.+?) ----\s*$', line)
        if m and first and not last: sections.append((i, m.group(1)[:48]))
    sections = [(first, 'lap loop head')] + sections
    inside, cur, first_fn = False, 0, None
    ops = collections.Counter()
    by_sec = collections.defaultdict(collections.Counter)
    for line in open(out):
        if re.match(r'_ZN4mcgp\w+:', line):
            inside = symbol in line and first_fn in (None, line)      # (the first instantiation: the default block shape)
            if inside: first_fn = line
        if line.startswith('.Lfunc_end'): inside = False
        s = line.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            chain = re.findall(r'race_kernel_reg\.hip\.h:(\d+):', s)
            cur = int(chain[-2]) if len(chain) >= 2 else (int(m.group(2)) if 'race_kernel_reg' in s or True else 0)
            if len(chain) < 2:
                # a .loc directly in some file: only count it if that file is race_kernel_reg.hip.h (file numbers vary: use the text)
                cur = int(m.group(2)) if re.search(r'\.loc\s+\d+\s+\d+\s+\d+\s*(;.*race_kernel_reg\.hip\.h.*)?$', s) else cur
            continue
        m = re.match(r'([a-z][a-z_0-9]+)', s)
        if not m or not inside or not (first <= cur < last): continue
        op = m.group(1)
        if not op.startswith(('v_', 's_', 'ds_', 'global_', 'scratch_', 'buffer_', 'flat_')): continue
        ops[op] += 1
        sec = [label for ln, label in sections if cur >= ln][-1]
        kind = 'VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'LDS' if op.startswith('ds_') else 'MEM'
        by_sec[sec][kind] += 1
        if op.startswith('v_cmp'): by_sec[sec]['cmp'] += 1
        if op.startswith('v_cndmask'): by_sec[sec]['cnd'] += 1
        if op in ('v_readlane_b32', 'v_writelane_b32'): by_sec[sec]['lane'] += 1
        if op == 's_nop': by_sec[sec]['nop'] += 1
        if op.startswith('v_mov'): by_sec[sec]['mov'] += 1
    print(f'{"section":50s} VALU  SALU  LDS  MEM | cmp  cnd  mov lane nop')
    for ln, label in sections:
        c = by_sec.get(label)
        if c: print(f'{label:50s} {c["VALU"]:5d} {c["SALU"]:5d} {c["LDS"]:4d} {c["MEM"]:4d} | {c["cmp"]:4d} {c["cnd"]:4d} {c["mov"]:4d} {c["lane"]:4d} {c["nop"]:3d}')
    tot = collections.Counter()
    for c in by_sec.values(): tot.update(c)
    print(f'{"TOTAL":50s} {tot["VALU"]:5d} {tot["SALU"]:5d} {tot["LDS"]:4d} {tot["MEM"]:4d} | {tot["cmp"]:4d} {tot["cnd"]:4d} {tot["mov"]:4d} {tot["lane"]:4d} {tot["nop"]:3d}')
    print()
    for op, k in ops.most_common(70): print(f'{op:28s} {k}')

if __name__ == '__main__':
    main()
