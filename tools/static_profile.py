#!/usr/bin/env python3
"""Static instruction attribution of the register kernel (N = 20): VALU / SALU / LDS counts per source region.

    python tools/static_profile.py [N] [reg|batch|wide] [-DMCGP_...]   # compiles reg_inst.hip with -gline-tables-only, prints the table

Every machine instruction of the listing is attributed to the source line of its .loc; lines of
race_kernel_reg.hip.h are grouped by the `// @region name` markers found in that file (helpers above the
kernel are grouped by function).  Counts are STATIC: rolled loops (grid sampling, lap 1 draws, the rare general paths) count once; the three overtake passes are three copies.
Not product code.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc')
SRC = os.path.join(CSRC, 'race_kernel_reg.hip.h')


def regions():
    marks = [(0, 'top')]
    for i, line in enumerate(open(SRC), 1):
        m = re.search(r'__device__ __forceinline__ \w[\w<> ]* (\w+)\(', line)
        if m:
            marks.append((i, 'fn ' + m.group(1)))
        m = re.search(r'// =+ (.+?) =+\s*$', line) or re.search(r'// ---- (.+?) ----\s*$', line)
        if m and i > 340:
            marks.append((i, m.group(1)[:52]))
    return marks


def common_fn(ln, _cache={}):
    """name of the function of race_common.hip.h that line `ln` belongs to"""
    if not _cache:
        marks = [(0, 'top')]
        for i, line in enumerate(open(os.path.join(CSRC, 'race_common.hip.h')), 1):
            m = re.search(r'__device__ (?:__forceinline__|inline) \w[\w<> ]* (\w+)\(', line)
            if m:
                marks.append((i, m.group(1)))
        _cache['m'] = marks
    return [label for first, label in _cache['m'] if ln >= first][-1]


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
    marks = regions()
    files, cur = {}, (None, 0)
    counts = collections.OrderedDict()
    inside, first_fn = False, None
    for line in open(out):
        if re.match(r'_ZN4mcgp\w+:', line):
            inside = symbol in line and first_fn in (None, line)      # (the first instantiation: the default block shape)
            if inside:
                first_fn = line
        if line.startswith('.Lfunc_end'):
            inside = False
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
            # inlined-at chain: attribute to the line inside reg_simulate (the frame just below the __global__ kernel)
            chain = re.findall(r'race_kernel_reg\.hip\.h:(\d+):', s)
            if len(chain) >= 2:
                cur = ('race_kernel_reg.hip.h', int(chain[-2]))
            continue
        m = re.match(r'(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)', s)
        if not m or not inside:
            continue
        op = m.group(0)
        kind = ('VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'LDS' if op.startswith('ds_')
                else 'SCR' if op.startswith('scratch_') else 'VMEM')
        if op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_barrier', 's_endpgm')):
            kind = 'CTRL'
        f, ln = cur
        if f and f.endswith('race_kernel_reg.hip.h'):
            sec = [label for first, label in marks if ln >= first][-1]
        elif f and f.endswith('race_common.hip.h'):
            sec = 'race_common: ' + common_fn(ln)
        elif f and f.endswith('race_isa.hip.h'):
            sec = 'race_isa: v_min_f64 / v_max_f64'
        else:
            sec = f'[{os.path.basename(f or "?")}]'
        counts.setdefault(sec, collections.Counter())[kind] += 1
    tot = collections.Counter()
    print(f'{"region":62s} {"VALU":>6s} {"SALU":>6s} {"LDS":>5s} {"CTRL":>5s} {"VMEM":>5s} {"SCR":>5s}')
    for sec, c in sorted(counts.items(), key=lambda kv: -kv[1]['VALU']):
        print(f'{sec:62s} {c["VALU"]:6d} {c["SALU"]:6d} {c["LDS"]:5d} {c["CTRL"]:5d} {c["VMEM"]:5d} {c["SCR"]:5d}')
        tot.update(c)
    print(f'{"TOTAL":62s} {tot["VALU"]:6d} {tot["SALU"]:6d} {tot["LDS"]:5d} {tot["CTRL"]:5d} {tot["VMEM"]:5d} {tot["SCR"]:5d}')


if __name__ == '__main__':
    main()
