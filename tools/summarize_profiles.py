#!/usr/bin/env python3
"""rocprofv3 output of tools/profile.sh (gpurun_out/<tag>_<workload>_<pass>/) -> tracked summaries under profiles/.

    python tools/summarize_profiles.py r4 S60 S78 HET N10 N25 S60_orders      (one script for every round's tag)

Writes / updates
    profiles/<tag>_counters.json     {"source_hash": ..., "workloads": {name: per-launch counter averages + derived figures}}
    profiles/<tag>_<name>_kernel_stats.csv   verbatim `--kernel-trace --stats` summary of that run
    profiles/<tag>_summary.md        the per-workload table

source_hash is what the LOADED library reported on the GPU box during the profiled run (mcgp_build_hash(): the hash of
the sources compiled into the binary); bench.py quotes a workload's counters only while the library it loads reports
the same hash.
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB, separate passes; FETCH_SIZE x 2 (gfx950
tallies 128-B requests at 64 B) -- this kernel reads a few KB per block, so raw and doubled figures are both kept.
VALUBusy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8), the gfx9 derived-metric formula.
"""
import collections
import csv
import re
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles')
PASSES = ('sq1', 'sq2', 'grbm', 'fetch', 'write')
SIMS = 10_000_000


def newest(paths):
    """gpurun merges each call's files INTO gpurun_out/: a pass directory may still hold an earlier call's file
    (other PID in the name).  Keep the newest one."""
    paths = sorted(paths, key=os.path.getmtime)
    return paths[-1:]


def counters(tag, name, sub):
    acc, meta = collections.defaultdict(list), {}
    for f in newest(glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_{name}_{sub}', '*', '*_counter_collection.csv'))):
        for r in csv.DictReader(open(f)):
            if 'race_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count',
                                          'Accum_VGPR_Count', 'SGPR_Count', 'Scratch_Size') if k in r}
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta


def one(tag, name):
    c, meta = {}, {}
    for sub in PASSES:
        cc, m = counters(tag, name, sub)
        c.update(cc)
        meta = m or meta
    if not c:
        return None
    stats = newest(glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_{name}_stats', '*', '*_kernel_stats.csv')))
    kernel_ms = calls = None
    if stats:
        shutil.copy(stats[0], os.path.join(OUT, f'{tag}_{name}_kernel_stats.csv'))
        for r in csv.DictReader(open(stats[0])):
            if 'race_kernel' in r['Name']:
                kernel_ms, calls = float(r['AverageNs']) / 1e6, int(r['Calls'])
    bench = None
    try:
        with open(os.path.join(ROOT, 'gpurun_out', f'{tag}_{name}_stats.json')) as f:
            bench = json.loads(f.read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    hash_file = os.path.join(ROOT, 'gpurun_out', f'{tag}_{name}_hash.txt')
    src = open(hash_file).read().strip() if os.path.exists(hash_file) else None
    w = dict(sims_per_launch=SIMS, kernel=meta, kernel_ms_avg_stats=kernel_ms, calls=calls, source_hash=src, **c)
    if bench:
        w['bench_under_rocprof'] = {'kernel_ms_avg_hip_events': bench['roofline']['kernel_ms_avg'],
                                    'value': bench['value'], 'workload': bench['config']['workload']}
    g = c.get
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        w['fetch_bytes_raw'] = c['FETCH_SIZE'] * 1024
        w['write_bytes'] = c['WRITE_SIZE'] * 1024
        w['hbm_bytes_per_launch'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
    if 'SQ_THREAD_CYCLES_VALU' in c and 'SQ_ACTIVE_INST_VALU' in c:
        w['active_lane_ratio'] = c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU'] / 64
    if 'GRBM_GUI_ACTIVE' in c:
        gui = c['GRBM_GUI_ACTIVE'] / 8
        if 'SQ_ACTIVE_INST_VALU' in c:
            w['valu_busy'] = 4 * c['SQ_ACTIVE_INST_VALU'] / 1024 / gui
        if 'SQ_WAVE_CYCLES' in c:
            w['waves_per_simd'] = 4 * c['SQ_WAVE_CYCLES'] / 1024 / gui
        if kernel_ms:
            w['shader_clock_ghz'] = gui / (kernel_ms * 1e-3) / 1e9
    if kernel_ms and 'SQ_INSTS_VALU' in c:
        w['valu_issue_frac_of_peak'] = c['SQ_INSTS_VALU'] / (kernel_ms * 1e-3) / (1024 * 2.4e9 / 2)
    if 'SQ_WAVE_CYCLES' in c:
        for k, label in (('SQ_ACTIVE_INST_ANY', 'issue_frac'), ('SQ_WAIT_ANY', 'wait_frac'), ('SQ_WAIT_INST_ANY', 'issue_stall_frac')):
            if k in c:
                w[label] = c[k] / c['SQ_WAVE_CYCLES']
    return w


def main():
    tag, names = sys.argv[1], sys.argv[2:]
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f'{tag}_counters.json')
    doc = {'source_hash': None, 'workloads': {}}
    if os.path.exists(path):
        with open(path) as f:
            doc = json.load(f)
    for name in names:
        w = one(tag, name)
        if w:
            doc['workloads'][name] = w
    hashes = {w.get('source_hash') for w in doc['workloads'].values()}
    doc['source_hash'] = hashes.pop() if len(hashes) == 1 else None      # one hash for the file only if all runs agree
    with open(path, 'w') as f:
        json.dump(doc, f, indent=1)
    lines = [f'# {tag}: rocprofv3 per-workload summary (10^7 simulations per launch, 1 MI355X)', '',
             f'Source hash of the profiled build: `{doc["source_hash"]}` (per workload in `{tag}_counters.json`). Passes: '
             '`--kernel-trace --stats`; PMC sets in separate runs (tools/profile.sh).', '',
             '| workload | kernel | grid x block | VGPR | kernel ms (stats) | sims/s | VALU / SALU / LDS per wave-lap | active lanes | VALUBusy | '
             'VALU issue frac of peak | issue / wait of wave-cycles | clock GHz | HBM bytes per launch (fetch x2 + write) | LDS conflict share |',
             '|---|---|---|---|---|---|---|---|---|---|---|---|---|---|']
    laps = {'S60': 60, 'S78': 78, 'HET': 60, 'N10': 60, 'N25': 60, 'S50': 50, 'EVT': 34}
    for name, w in doc['workloads'].items():
        k = w.get('kernel', {})
        base = name.split('_')[0]
        L = laps.get(base, 60)
        m = re.search(r'(\d+) laps', w.get('bench_under_rocprof', {}).get('workload', ''))
        if m:
            L = int(m.group(1))
        waves = w['sims_per_launch'] / 64
        ms = w.get('kernel_ms_avg_stats')
        per = lambda key: f"{w[key] / waves / L:.0f}" if key in w else '?'
        f2 = lambda key, fmt='{:.2f}': fmt.format(w[key]) if key in w and w[key] is not None else '?'
        conflict = (w['SQ_LDS_BANK_CONFLICT'] / w['SQ_LDS_IDX_ACTIVE']) if 'SQ_LDS_IDX_ACTIVE' in w and w['SQ_LDS_IDX_ACTIVE'] else None
        lines.append(f"| {name} | `{k.get('Kernel_Name', '?').split('(')[0].replace('void ', '')}` | {k.get('Grid_Size')} x {k.get('Workgroup_Size')} | "
                     f"{k.get('VGPR_Count')} | {ms:.2f} | {w['sims_per_launch'] / (ms * 1e-3):.3g} | {per('SQ_INSTS_VALU')} / {per('SQ_INSTS_SALU')} / {per('SQ_INSTS_LDS')} | "
                     f"{f2('active_lane_ratio')} | {f2('valu_busy', '{:.1%}')} | {f2('valu_issue_frac_of_peak')} | {f2('issue_frac')} / {f2('wait_frac')} | "
                     f"{f2('shader_clock_ghz')} | {f2('hbm_bytes_per_launch', '{:.3g}')} | {conflict:.3f} |" if ms else f'| {name} | incomplete |')
    lines += ['', 'Columns: "VALU / SALU / LDS per wave-lap" = SQ_INSTS_* per launch / (simulations / 64) / laps; "active lanes" = '
              'SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64; "VALUBusy" = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); '
              '"VALU issue frac of peak" = SQ_INSTS_VALU / kernel time / (1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction); '
              'HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB, separate passes (gfx950 correction for FETCH_SIZE).  '
              f'`VGPR` is rocprofv3\'s VGPR_Count = allocated registers / 2 (168 -> 84 for N = 20).', '',
              f'Companion files: `{tag}_ablate.txt` (tools/ablate.sh: DUP = section run twice, SKIP = section left out; kernel ms at 4e6 '
              f'simulations), `{tag}_ab.txt` (same-box A/B runs of the optimisation log, tools/ab.sh), `{tag}_deep_parity.txt` '
              f'(tools/deep_parity.py), `{tag}_deviate_bias.txt` (tools/deviate_bias_gpu.py), `{tag}_<workload>_kernel_stats.csv` '
              '(verbatim `--kernel-trace --stats`).']
    with open(os.path.join(OUT, f'{tag}_summary.md'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
