#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged into gpurun_out/<tag>_* into the tracked summaries under profiles/.

    python tools/summarize_profiles.py r1        # reads gpurun_out/r1_stats, r1_fetch, r1_write, r1_sq1, r1_sq2, r1_grbm

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --kernel-trace --stats summary),
profiles/<tag>_counters.json (per-launch averages of every PMC counter collected) and
profiles/<tag>_summary.md.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB
units of 64-B requests, collected in separate passes; on gfx950 FETCH_SIZE under-reports wide
streaming reads by 2x -- this kernel reads a few KB per block, so both the raw and the doubled
figure are given.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(tag, sub):
    files = glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_{sub}', '*', '*_counter_collection.csv'))
    acc = collections.defaultdict(list)
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if 'race_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size',
                                          'VGPR_Count', 'SGPR_Count', 'Scratch_Size') if k in r}
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
    sims = float(sys.argv[2]) if len(sys.argv) > 2 else 1e7
    laps, n = 60, 20
    out_dir = os.path.join(ROOT, 'profiles')
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(ROOT, 'gpurun_out', f'{tag}_stats', '*', '*_kernel_stats.csv'))
    kernel_ms = None
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, f'{tag}_kernel_stats.csv'))
        for r in csv.DictReader(open(stats[0])):
            if 'race_kernel' in r['Name']:
                kernel_ms = float(r['AverageNs']) / 1e6
                calls = int(r['Calls'])
    bench = os.path.join(ROOT, 'gpurun_out', f'{tag}_stats_bench.json')
    if os.path.exists(bench):
        shutil.copy(bench, os.path.join(out_dir, f'{tag}_bench_under_rocprof.json'))
    allc, meta = {}, {}
    for sub in ('fetch', 'write', 'sq1', 'sq2', 'grbm'):
        c, m = counters(tag, sub)
        allc.update(c)
        meta = m or meta
    stored = os.path.join(out_dir, f'{tag}_counters.json')
    if not allc and os.path.exists(stored):
        # raw rocprofv3 output (gpurun_out/, scratch) is gone: re-render from the tracked counters file
        with open(stored) as f:
            old = json.load(f)
        allc, meta, kernel_ms = old['counters'], old['kernel'], old['kernel_ms_avg']
        calls = old.get('calls', 6)
    traffic = None
    if 'FETCH_SIZE' in allc and 'WRITE_SIZE' in allc:
        traffic = dict(fetch_bytes_raw=allc['FETCH_SIZE'] * 1024, write_bytes=allc['WRITE_SIZE'] * 1024,
                       hbm_bytes_per_launch=(2 * allc['FETCH_SIZE'] + allc['WRITE_SIZE']) * 1024,
                       note='FETCH_SIZE doubled per the gfx950 correction; per launch of %g simulations' % sims)
    if traffic:
        with open(os.path.join(out_dir, 'traffic.json'), 'w') as f:
            json.dump(dict(workload='S60', sims_per_launch=int(sims), source=f'{tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE',
                           **traffic), f, indent=1)
    with open(os.path.join(out_dir, f'{tag}_counters.json'), 'w') as f:
        json.dump(dict(kernel=meta, kernel_ms_avg=kernel_ms, calls=calls if kernel_ms else None, sims_per_launch=sims, counters=allc, traffic=traffic),
                  f, indent=1)
    waves = sims / 64
    g = lambda k: allc.get(k, float('nan'))
    lines = [f'# {tag}: rocprofv3 summary of `python bench.py` (S60, {sims:g} simulations per launch, 1 MI355X)', '']
    lines.append(f'* kernel: `{meta.get("Kernel_Name", "?")[:60]}`  grid {meta.get("Grid_Size")} x {meta.get("Workgroup_Size")}, '
                 f'VGPR {meta.get("VGPR_Count")}, SGPR {meta.get("SGPR_Count")}, LDS {meta.get("LDS_Block_Size")} B, scratch {meta.get("Scratch_Size")}')
    if kernel_ms:
        lines.append(f'* `--kernel-trace --stats`: average duration **{kernel_ms:.3f} ms** over {calls} launches '
                     f'({sims / (kernel_ms * 1e-3):.3g} simulations/s kernel-only); `{tag}_kernel_stats.csv`')
    if traffic:
        lines.append(f'* HBM traffic per launch (PMC, separate passes): FETCH_SIZE {traffic["fetch_bytes_raw"]:.3g} B raw '
                     f'(x2 = {2 * traffic["fetch_bytes_raw"]:.3g} B), WRITE_SIZE {traffic["write_bytes"]:.3g} B; '
                     f'algorithmic bytes in histogram-only mode: {n * n * 8} B per launch '
                     f'(20 B/simulation = {20 * sims:.3g} B only when finishing orders are requested)')
    lines += ['', '| per simulated lap and wave (64 simulations) | value |', '|---|---|']
    for k, label in (('SQ_INSTS_VALU', 'VALU instructions'), ('SQ_INSTS_SALU', 'SALU instructions'),
                     ('SQ_INSTS_LDS', 'LDS instructions'), ('SQ_INSTS_SMEM', 'SMEM instructions')):
        lines.append(f'| {label} | {g(k) / waves / laps:.0f} |')
    lines += ['', '| ratio | value |', '|---|---|']
    lines.append(f'| active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64) | {g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") / 64:.2f} |')
    wc = g('SQ_WAVE_CYCLES')
    for k, label in (('SQ_ACTIVE_INST_ANY', 'wave-cycles issuing any instruction'), ('SQ_ACTIVE_INST_VALU', '... VALU'),
                     ('SQ_ACTIVE_INST_SCA', '... scalar'), ('SQ_ACTIVE_INST_LDS', '... LDS'),
                     ('SQ_WAIT_ANY', 'wave-cycles waiting (s_waitcnt)'), ('SQ_WAIT_INST_ANY', 'wave-cycles stalled on issue')):
        lines.append(f'| {label} / SQ_WAVE_CYCLES | {g(k) / wc:.2f} |')
    lines.append(f'| LDS bank-conflict cycles / LDS active cycles | {g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"):.3f} |')
    if 'GRBM_GUI_ACTIVE' in allc:
        gui = g('GRBM_GUI_ACTIVE') / 8          # rocprofv3 sums the 8 XCDs
        lines.append(f'| **VALUBusy** = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) (gfx9 derived-metric formula) | **{400 * g("SQ_ACTIVE_INST_VALU") / 1024 / gui:.1f} %** |')
        lines.append(f'| resident waves per SIMD, time average (4 x SQ_WAVE_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8)) | {4 * wc / 1024 / gui:.2f} |')
        if kernel_ms:
            lines.append(f'| shader clock (GRBM_GUI_ACTIVE / 8 / kernel time) | {gui / (kernel_ms * 1e-3) / 1e9:.2f} GHz |')
    if kernel_ms:
        simd_cycles = 1024 * kernel_ms * 1e-3 * 2.4e9
        lines.append(f'| VALU instructions / (1024 SIMDs x kernel time x 2.4 GHz) | {g("SQ_INSTS_VALU") / simd_cycles:.3f} per SIMD-cycle |')
    abl = os.path.join(ROOT, 'gpurun_out', 'ablate.txt')
    if not os.path.exists(abl) or 'DUP=0' not in open(abl).read():
        abl = os.path.join(out_dir, f'{tag}_ablate.txt')      # the tracked copy of the DUP/SKIP run
    if os.path.exists(abl):
        vals = collections.defaultdict(list)
        for line in open(abl):
            k, _, v = line.strip().partition(' kernel_ms=')
            if v:
                vals[k].append(float(v))
        if 'DUP=0' in vals:
            if os.path.abspath(abl) != os.path.abspath(os.path.join(out_dir, f'{tag}_ablate.txt')):
                shutil.copy(abl, os.path.join(out_dir, f'{tag}_ablate.txt'))
            base = sum(vals['DUP=0']) / len(vals['DUP=0'])
            names = {'DUP=1': 'sorting network after the lap step (run twice)', 'DUP=2': 'per-lap RNG pre-pass: 10 Philox blocks + 20 deviates (run twice)',
                     'DUP=4': '_update_positions (run twice)', 'DUP=8': 'one extra transposition re-sort per successful overtake pass',
                     'SKIP=1': 'overtake pass loop left out', 'SKIP=2': 'event handlers left out', 'SKIP=4': 'grid sampling left out',
                     'SKIP=8': 'laps 2..L left out (grid + lap 1 + classification remain)'}
            lines += ['', f'## Where the time goes (tools/ablate.sh, same box, 4e6 simulations, baseline {base:.2f} ms)', '',
                      'DUP = section run twice (idempotent, results unchanged); SKIP = section left out (timing only).', '',
                      '| variant | kernel ms | share of baseline |', '|---|---|---|']
            for k in ('DUP=1', 'DUP=2', 'DUP=4', 'DUP=8', 'SKIP=1', 'SKIP=2', 'SKIP=4', 'SKIP=8'):
                if k in vals:
                    m = sum(vals[k]) / len(vals[k])
                    share = (m - base) / base if k.startswith('DUP') else (base - m) / base
                    if k == 'SKIP=8':
                        share = m / base
                    lines.append(f'| {k}: {names[k]} | {m:.2f} | {share:.1%} |')
    mb = os.path.join(out_dir, f'{tag}_mapping_microbench.json')
    if os.path.exists(mb):
        d = json.load(open(mb))
        lines += ['', '## Lane mapping, ordering step only (tools/mapping_microbench.hip)', '',
                  f'* lane-per-car, rank by counting over ds_bpermute, 3 races per wave: {d["lane_per_car"]["ordering_steps_per_s"]:.3g} field sorts/s',
                  f'* lane-per-race, 97-comparator network in VGPRs, 64 races per wave: {d["lane_per_race"]["ordering_steps_per_s"]:.3g} field sorts/s '
                  f'(**{d["ratio_race_over_car"]:.1f}x**)']
    with open(os.path.join(out_dir, f'{tag}_summary.md'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
