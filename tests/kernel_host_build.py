"""Host DEBUGGING build of the kernel sources (tools/emu): compile, load, run.  Test infrastructure only --
the product (monte_carlo_gp_amd/) never imports this and has no CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMU_DIR = os.path.join(ROOT, 'tools', 'emu')
LIB = os.path.join(EMU_DIR, 'libmcgp_emu.so')
CSRC = os.path.join(ROOT, 'monte_carlo_gp_amd', 'csrc')

_libs = {}

# named diagnostic variants of the host build: extra -D flags of csrc/race_kernel_reg.hip.h
VARIANTS = {
    None: [],
    'grid_exact': ['-DMCGP_GRID_EXACT=1'],      # _sample_grid takes the exact (dividing) path for every draw
    # the reference-width build decides every event draw and overtake pass with its exact 53-bit code (the path of a draw
    # word equal to the leading word of its threshold, one in 2^32), and reads every deviate's row from device memory
    'wide_exact': ['-DMCGP_WIDE_EXACT=1'],
    # ... a draw word within 2^27 of the leading word of its threshold counts as a tie: a few per cent of the draws
    'wide_near_ties': ['-DMCGP_WIDE_TIE_SHIFT=27'],
}


def build(variant=None):
    LIB = os.path.join(EMU_DIR, 'libmcgp_emu.so' if variant is None else f'libmcgp_emu_{variant}.so')
    tag = '' if variant is None else '_' + variant
    srcs = [os.path.join(EMU_DIR, f) for f in ('emu_kernel.cpp', 'race_isa_host.h', 'hip/hip_runtime.h')]
    srcs += [os.path.join(CSRC, f) for f in ('race_kernel_reg.hip.h', 'race_common.hip.h', 'params_build.h', 'normal_table.h', 'normal53_table.h', 'sort_networks.h')]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
        # four translation units (field sizes n % 4 == k) compiled side by side, then linked
        flags = ['-O1', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-I' + EMU_DIR, '-DEMU_PARTS=4']
        flags += VARIANTS[variant]
        objs = [os.path.join(EMU_DIR, f'emu_part{k}{tag}.o') for k in range(4)]
        procs = [subprocess.Popen(['g++'] + flags + [f'-DEMU_PART={k}', '-c', '-o', objs[k],
                                                     os.path.join(EMU_DIR, 'emu_kernel.cpp')]) for k in range(4)]
        for pr in procs:
            if pr.wait() != 0:
                raise RuntimeError('host build of the kernel sources failed')
        subprocess.check_call(['g++', '-shared', '-o', LIB] + objs)
    return LIB


def lib(variant=None):
    if variant not in _libs:
        L = C.CDLL(build(variant))
        L.emu_run.restype = C.c_int
        _libs[variant] = L
    return _libs[variant]


def run(case, n_sims, seed, sim_offset=0, set_pop=None, fixed_grid=None, variant=None, deviates=32):
    """(hist, orders) of the kernel source executed on the host for a golden-case dict."""
    from monte_carlo_gp_amd import RaceConfig
    from monte_carlo_gp_amd.simulation import RaceSimulator, _Problem, _dptr
    import oracle_py as O
    drivers = list(case['grid_probs'])
    p = _Problem(RaceConfig(**case['config']), drivers, case['base_pace'], case['tire_deg'], case['driver_variance'],
                 case['driver_dnf_rates'], case['track_condition'], set_pop or O.load_cases()['set_pop'], deviates)
    g = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
    n = p.n
    hist = np.zeros((n, n), np.uint64)
    orders = np.zeros((n_sims, n), np.uint8)
    err = C.c_char_p()
    fg = None
    if fixed_grid is not None:
        fg = np.ascontiguousarray(fixed_grid, np.uint8).ctypes.data_as(C.c_void_p)
    rc = lib(variant).emu_run(C.byref(p.cfg), C.byref(p.drv), _dptr(g), C.c_uint32(n), C.c_uint64(n_sims),
                       C.c_uint64(sim_offset), C.c_uint64(seed), hist.ctypes.data_as(C.c_void_p),
                       orders.ctypes.data_as(C.c_void_p), fg, C.byref(err))
    if rc == -100:
        raise NotServed(err.value.decode())
    assert rc == 0, err.value
    return hist.astype(np.int64), orders


class NotServed(Exception):
    """The register kernel hands this problem to the generic kernel (csrc: reg_kernel_serves)."""
