"""bench.py's host-side helpers (no GPU): counters are quoted only for the sources they were profiled from;
the CPU legs report what they ran on."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from monte_carlo_gp_amd import _native as N  # noqa: E402


def test_source_hash_covers_the_kernel_sources(tmp_path):
    h = N.source_hash()
    assert len(h) == 16 and h == N.source_hash()
    names = {os.path.basename(p) for p in N._sources()}
    assert {'race_kernel_reg.hip.h', 'race_common.hip.h', 'race_isa.hip.h', 'mcgp_hip.hip', 'params_build.h',
            'frontend_exp.h', 'normal_table.h', 'Makefile', 'mcgp.h', 'source_hash.py', 'reg_inst.hip'} <= names


def test_profiled_counters_are_refused_for_other_sources(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'profiles')
    w, why = bench.profiled_counters('S60', 10_000_000, False)
    assert w is None and 'no usable' in why
    doc = {'source_hash': 'deadbeefdeadbeef', 'workloads': {'S60': {'sims_per_launch': 10_000_000, 'SQ_INSTS_VALU': 1.0}}}
    (tmp_path / 'profiles' / bench.COUNTERS_FILE).write_text(json.dumps(doc))
    w, why = bench.profiled_counters('S60', 10_000_000, False)
    assert w is None and 're-profile' in why
    doc['source_hash'] = N.build_hash()            # what the LOADED binary reports (= the tree's hash, or lib() refuses)
    assert N.build_hash() == N.source_hash()
    (tmp_path / 'profiles' / bench.COUNTERS_FILE).write_text(json.dumps(doc))
    w, why = bench.profiled_counters('S60', 10_000_000, False)
    assert why is None and w['SQ_INSTS_VALU'] == 1.0
    assert bench.profiled_counters('S60', 10_000_000, True)[0] is None              # MCGP_LIB set
    assert bench.profiled_counters('S78', 10_000_000, False)[0] is None             # workload not profiled
    assert bench.profiled_counters('S60', 1_000_000, False)[0] is None              # other launch size


def test_host_info_and_bounded_cpu_legs():
    info = bench.host_info()
    assert info['nproc'] >= 1 and 1 <= info['bench_threads'] <= max(16, info['usable_cores']) and info['bench_threads_source']
    r = bench.cpu_baseline('N10', seconds=0.5, all_core_seconds=0.5)
    assert r['kind'] == 'port' and r['cores'] == 1 and r['value'] > 0
    a = r['all_cores']
    assert a['cores'] == info['bench_threads'] and a['value'] > 0 and 'Philox' in a['sample']
    assert r['reference_python_sims_per_s']['S60'] == 180


def test_synthetic_25_car_workload_is_well_formed():
    case, set_pop = bench.load_workload('N25')
    assert len(case['grid_probs']) == 25 and all(abs(sum(v) - 1) < 1e-12 for v in case['grid_probs'].values())
    assert set(case['base_pace']) == set(case['grid_probs']) and set_pop
