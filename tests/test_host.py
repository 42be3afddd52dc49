"""Host-side logic and the C-ABI surface; no GPU needed (no compute entry point is called
successfully here: without a device they must fail loudly)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import oracle_py as O
from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N, histogram_to_probs
from monte_carlo_gp_amd import config as K
from monte_carlo_gp_amd.simulation import _Problem, DEFAULT_SET_POP


def test_library_exports_every_declared_symbol():
    L = N.lib()
    with open(O.ROOT + '/include/mcgp.h') as f:
        header = f.read()
    declared = set(re.findall(r'\b(mcgp_[a-z_]+)\s*\(', header))
    assert declared == set(N.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mcgp_abi_version() == N.ABI_VERSION
    assert L.mcgp_device_count() >= 0


def test_struct_layout_matches_header():
    # 2 x int32, 8 x double, 5 + 5 double, 5 int32, 2 int32
    assert C.sizeof(N.McgpConfig) == 8 + 8 * 8 + 10 * 8 + 5 * 4 + 2 * 4 + 4  # trailing pad to 8
    assert C.sizeof(N.McgpDrivers) == 6 * C.sizeof(C.c_void_p)
    assert C.sizeof(O.OrcConfig) == C.sizeof(N.McgpConfig)


def test_constants_match_reference_tables():
    import json
    with open(O.GOLDEN_DIR + '/misc.json') as f:
        m = json.load(f)
    assert K.DRIVER_TEAMS == m['driver_teams'] and list(K.DRIVER_TEAMS) == list(m['driver_teams'])
    assert K.DEFAULT_DNF_RATES == m['dnf_rates']
    assert K.TIRE_COMPOUNDS == m['tire_compounds']
    assert K.CIRCUITS == m['circuits'] and list(K.CIRCUITS) == list(m['circuits'])
    assert m['race_config_bahrain'] == [57, 21.0, 0.6, K.SC_PROBABILITY, K.VSC_PROBABILITY,
                                        K.RED_FLAG_PROBABILITY, K.DRS_DELTA]


def test_default_set_pop_is_the_hashseed0_outcome():
    assert DEFAULT_SET_POP == {k: v for k, v in O.load_cases()['set_pop'].items() if k in DEFAULT_SET_POP}


def test_problem_resolution_applies_reference_defaults():
    case = O.load_case('HET')          # has missing dict entries and an unknown driver
    cfg = RaceConfig(**case['config'])
    drivers = list(case['grid_probs'])
    p = _Problem(cfg, drivers, case['base_pace'], case['tire_deg'], case['driver_variance'],
                 case['driver_dnf_rates'], 'dry', DEFAULT_SET_POP)
    ref = O.Problem(case)
    for k in ref.arr:
        assert np.array_equal(p.arrays[k], ref.arr[k]), k
    i = drivers.index('XXX')
    assert p.arrays['team_dnf'][i] == 0.002 and p.arrays['tire_deg'][i] == 0.05 and p.arrays['tire_deg_pit'][i] == 0.0
    assert p.arrays['base_pace'][drivers.index('HUL')] == 90.0
    assert p.arrays['variance'][drivers.index('OCO')] == 0.15
    j = drivers.index('VER')
    assert p.arrays['lap_dnf'][j] == p.arrays['team_dnf'][j]
    assert bytes(p.cfg) == bytes(ref.cfg)


def test_histogram_to_probs_shape():
    h = np.array([[3, 0, 1], [1, 3, 0], [0, 1, 3]])
    out = histogram_to_probs(h, ['A', 'B', 'C'], 4)
    assert out == {'A': {1: 0.75, 3: 0.25}, 'B': {1: 0.25, 2: 0.75}, 'C': {2: 0.25, 3: 0.75}}


def test_argument_validation_and_loud_failure_without_device():
    sim = RaceSimulator(RaceConfig(10, 20.0, 0.5, 0.01, 0.01, 0.001, {}, 2, 0.3, K.TIRE_COMPOUNDS, {}))
    assert sim.run_monte_carlo(0, {'A': [1.0]}, {}, {}, {}) == {}
    assert sim.run_monte_carlo(10, {}, {}, {}, {}) == {}
    with pytest.raises(ValueError):
        sim.run_monte_carlo(10, {str(i): [1 / 33] * 33 for i in range(33)}, {}, {}, {}, seed=1)
    with pytest.raises(ValueError):
        sim.run_monte_carlo(10, {'A': [1.0]}, {}, {}, {}, seed=1, track_condition='snow')
    if N.lib().mcgp_device_count() == 0:
        with pytest.raises(N.McgpError) as e:
            sim.run_monte_carlo(10, {'A': [0.5, 0.5], 'B': [0.5, 0.5]}, {}, {}, {}, seed=1)
        assert e.value.code == -2 and 'no HIP device' in str(e.value)


def test_c_abi_rejects_bad_arguments():
    L = N.lib()
    case = O.load_case('N10')
    p = _Problem(RaceConfig(**case['config']), list(case['grid_probs']), case['base_pace'], case['tire_deg'],
                 case['driver_variance'], case['driver_dnf_rates'], 'dry', DEFAULT_SET_POP)
    g = np.full((10, 10), 0.1)
    h = np.zeros(100, np.uint64)
    gp, hp = g.ctypes.data_as(C.POINTER(C.c_double)), h.ctypes.data_as(C.POINTER(C.c_uint64))
    assert L.mcgp_run(None, C.byref(p.drv), gp, 10, 5, 0, 1, 0, hp, None) == -1
    assert L.mcgp_run(C.byref(p.cfg), C.byref(p.drv), gp, 0, 5, 0, 1, 0, hp, None) == -1
    assert L.mcgp_run(C.byref(p.cfg), C.byref(p.drv), gp, 33, 5, 0, 1, 0, hp, None) == -1
    assert L.mcgp_run(C.byref(p.cfg), C.byref(p.drv), None, 10, 5, 0, 1, 0, hp, None) == -1
    assert b'NULL' in L.mcgp_last_error()
    bad = N.McgpConfig.from_buffer_copy(bytes(p.cfg))
    bad.total_laps = 5000
    assert L.mcgp_run(C.byref(bad), C.byref(p.drv), gp, 10, 5, 0, 1, 0, hp, None) == -1
    g[3, 4] = -0.5
    assert L.mcgp_run(C.byref(p.cfg), C.byref(p.drv), gp, 10, 5, 0, 1, 0, hp, None) == -1
    grid = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 8], np.uint8)    # not a permutation
    o = np.zeros(10, np.uint8)
    assert L.mcgp_simulate_race(C.byref(p.cfg), C.byref(p.drv), grid.ctypes.data_as(C.POINTER(C.c_uint8)), 10, 0, 1,
                                0, o.ctypes.data_as(C.POINTER(C.c_uint8))) == -1
    assert h.sum() == 0


def test_seed_resolution():
    import random
    assert RaceSimulator._resolve_seed(42) == 42
    assert RaceSimulator._resolve_seed(-42) == 42
    assert RaceSimulator._resolve_seed(2 ** 70 + 5) == 5
    random.seed(123)
    a = RaceSimulator._resolve_seed(None)
    random.seed(123)
    assert RaceSimulator._resolve_seed(None) == a       # a globally seeded run stays reproducible (Q20)


def test_one_hip_runtime_whichever_is_loaded_first():
    """libmcgp_hip.so (NEEDED libamdhip64.so.7, RUNPATH /opt/rocm) loaded before torch (which asks for
    its bundled copy as `libamdhip64.so`) used to map TWO HIP runtimes; _native._bind_hip_runtime maps torch's copy
    first.  Checked in fresh interpreters, both orders (no GPU needed: only the mappings are inspected); with the
    binding switched off the library refuses to coexist silently -- assert_single_hip_runtime raises."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from monte_carlo_gp_amd import _native as N\n"
            "order = sys.argv[1]\n"
            "if order == 'torch_first':\n    import torch\n    N.lib()\n"
            "else:\n    N.lib()\n    import torch\n"
            "try:\n    N.assert_single_hip_runtime(); ok = 1\n"
            "except N.McgpError: ok = 0\n"
            "print(len(N.hip_runtimes_mapped()), ok)\n") % (str(__import__('os').path.dirname(N._PKG)),)
    import os
    env = {k: v for k, v in os.environ.items() if k != 'MCGP_HIP_RUNTIME'}
    for order in ('lib_first', 'torch_first'):
        out = subprocess.run([sys.executable, '-c', code, order], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split() == ['1', '1'], (order, out.stdout)
    if os.path.exists('/opt/rocm/lib/libamdhip64.so'):
        out = subprocess.run([sys.executable, '-c', code, 'lib_first'], env=dict(env, MCGP_HIP_RUNTIME='system'),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split() == ['2', '0'], out.stdout          # the failure mode, detected


def test_every_hashed_source_is_a_make_prerequisite():
    """an object that does not depend on a header the identity hash covers would stay stale under a fresh
    hash.  Every file csrc/source_hash.py hashes must be a prerequisite of the object(s) it is compiled into: every
    header of every object, mcgp_hip.hip / source_hash.py of the object that carries the hash."""
    import subprocess
    csrc = os.path.join(O.ROOT, 'monte_carlo_gp_amd', 'csrc')
    sys.path.insert(0, csrc)
    try:
        import source_hash
        hashed = {os.path.basename(f) for f in source_hash.sources()}
    finally:
        sys.path.remove(csrc)
    db = subprocess.run(['make', '-pn', '-C', csrc, 'build/reg_inst_20.o', 'build/mcgp_hip.o'], capture_output=True,
                        text=True).stdout
    deps = {}
    for line in db.splitlines():
        for target in ('build/reg_inst_20.o', 'build/mcgp_hip.o'):
            if line.startswith(target + ':'):
                deps[target] = {os.path.basename(x) for x in line.split(':', 1)[1].split()}
    headers = {f for f in hashed if f.endswith('.h')}
    assert headers and 'sort_networks.h' in headers and 'normal53_table.h' in headers
    assert headers | {'reg_inst.hip', 'Makefile'} <= deps['build/reg_inst_20.o'], headers - deps['build/reg_inst_20.o']
    assert hashed <= deps['build/mcgp_hip.o'], hashed - deps['build/mcgp_hip.o']


def test_a_stale_library_is_refused(tmp_path):
    """the binary carries the hash of the sources it was compiled from (mcgp_build_hash(), and the
    marker MCGP_BUILD_HASH=... in the file) and the binding checks CONTENTS, not time stamps.  A library built from
    other sources -- here: a copy of the real one with another hash patched in, and the newest mtime of all -- is
    refused before anything is launched: with MCGP_NO_BUILD=1 by build(), and, if it is mapped behind the binding's
    back, by lib() itself."""
    import os
    import shutil
    import subprocess
    import sys
    N.lib()                                                  # the real library exists and is current
    assert N.file_build_hash(N.LIB_PATH) == N.source_hash() == N.build_hash()
    stale = tmp_path / 'libmcgp_hip.so'
    shutil.copy(N.LIB_PATH, stale)
    blob = stale.read_bytes()
    marker = b'MCGP_BUILD_HASH=' + N.source_hash().encode()
    assert blob.count(marker) >= 1
    stale.write_bytes(blob.replace(marker, b'MCGP_BUILD_HASH=' + b'0123456789abcdef'))
    os.utime(stale)                                          # newer than every source: an mtime test would accept it
    assert N.file_build_hash(str(stale)) == '0123456789abcdef'
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from monte_carlo_gp_amd import _native as N\n"
            "N.LIB_PATH = sys.argv[1]\n"
            "mode = sys.argv[2]\n"
            "if mode == 'behind':\n    N._stale = lambda: False\n"      # pretend the file check passed
            "try:\n    N.lib(); print('loaded')\n"
            "except N.McgpError as e:\n    print('refused:', e)\n") % (os.path.dirname(N._PKG),)
    env = dict({k: v for k, v in os.environ.items() if k != 'MCGP_LIB'}, MCGP_NO_BUILD='1')
    for mode in ('file', 'behind'):
        out = subprocess.run([sys.executable, '-c', code, str(stale), mode], env=env, capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.startswith('refused:') and '0123456789abcdef' in out.stdout, (mode, out.stdout)


def test_device_argument_of_the_drop_in_class():
    cfg = RaceConfig(**O.load_case('S60')['config'])
    assert RaceSimulator(cfg).devices == [0] and RaceSimulator(cfg, device=3).devices == [3]
    assert RaceSimulator(cfg, device=[1, 0, 1]).devices == [1, 0, 1] and RaceSimulator(cfg, device=(2,)).device == 2
    with pytest.raises(ValueError):
        RaceSimulator(cfg, device=[])
    with pytest.raises(ValueError):
        RaceSimulator(cfg, device='every')
    if N.lib().mcgp_device_count() == 0:
        with pytest.raises(N.McgpError):
            RaceSimulator(cfg, device='all')                # no CPU path, no silent empty device list


def test_generated_sort_networks_sort_every_zero_one_input():
    """csrc/sort_networks.h (tools/gen_sort_networks.py): every comparator list, parsed from the header, sorts all 2^n
    zero-one inputs (zero-one principle: it sorts everything), is smaller than merge exchange for its size, is written in
    layers of disjoint comparators, and equals what the generator builds today."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('gen_sort_networks', os.path.join(root, 'tools', 'gen_sort_networks.py'))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    text = open(os.path.join(root, 'monte_carlo_gp_amd', 'csrc', 'sort_networks.h')).read()
    found = {}
    for m in re.finditer(r'kSortNetwork(\d+)\[(\d+)\]\[3\] = \{(.*?)\};', text, re.S):
        triples = [tuple(int(x) for x in t) for t in re.findall(r'\{(\d+), (\d+), (\d+)\}', m.group(3))]
        assert len(triples) == int(m.group(2))
        found[int(m.group(1))] = triples
    assert sorted(found) == sorted(G.SIZES) and 20 in found and len(found[20]) == 93
    for n, triples in found.items():
        net = [(a, b) for a, b, _ in triples]
        assert all(0 <= a < n and 0 <= b < n and a != b for a, b in net)
        assert len(net) < G.merge_exchange_size(n)
        layers = [s for _, _, s in triples]
        assert layers == sorted(layers)
        for s in set(layers):
            wires = [w for a, b, t in triples if t == s for w in (a, b)]
            assert len(wires) == len(set(wires)), (n, s)
        assert G.sorts_all_zero_one_inputs(n, net), n
        built, built_layers = G.build(n)
        assert built == net and built_layers == layers
    # the checker itself: a network with one comparator missing is caught
    assert not G.sorts_all_zero_one_inputs(10, [(a, b) for a, b, _ in found[10]][:-1])
