"""ctypes driver for the CPU oracle (oracle/libmcgp_oracle.so).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only.  The product package never imports it.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
LIB_PATH = os.path.join(ORACLE_DIR, 'libmcgp_oracle.so')

COMPOUNDS = ['SOFT', 'MEDIUM', 'HARD', 'INTERMEDIATE', 'WET']
COMPOUND_ID = {c: i for i, c in enumerate(COMPOUNDS)}
TRACK_ID = {'dry': 0, 'damp': 1, 'wet': 2}
RNG_MT, RNG_PHILOX, RNG_PHILOX53 = 0, 1, 2


class OrcConfig(C.Structure):
    _fields_ = [
        ('total_laps', C.c_int32), ('track_condition', C.c_int32),
        ('pit_loss', C.c_double), ('overtake_delta', C.c_double),
        ('sc_probability', C.c_double), ('vsc_probability', C.c_double),
        ('red_flag_probability', C.c_double), ('drs_delta', C.c_double),
        ('dirty_air_threshold', C.c_double), ('dirty_air_penalty', C.c_double),
        ('comp_pace_delta', C.c_double * 5), ('comp_deg_rate', C.c_double * 5),
        ('comp_optimal_laps', C.c_int32 * 5),
        ('pop_soft_hard', C.c_int32), ('pop_medium_hard', C.c_int32),
    ]


class OrcDrivers(C.Structure):
    _fields_ = [(k, C.POINTER(C.c_double)) for k in
                ('base_pace', 'tire_deg', 'tire_deg_pit', 'variance', 'team_dnf', 'lap_dnf')]


class OrcTrace(C.Structure):
    _fields_ = [('n_trace', C.c_int64),
                ('cum', C.POINTER(C.c_double)), ('tbl', C.POINTER(C.c_double)), ('last', C.POINTER(C.c_double)),
                ('age', C.POINTER(C.c_int16)), ('dnf_lap', C.POINTER(C.c_int16)),
                ('comp', C.POINTER(C.c_uint8)), ('used', C.POINTER(C.c_uint8)),
                ('dnf', C.POINTER(C.c_uint8)), ('drs', C.POINTER(C.c_uint8))]


def build(force=False):
    """Compile the oracle if the shared object is missing or stale."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ('mcgp_oracle.c', 'mcgp_oracle.h', 'normal_table.h', 'normal53_table.h', 'frontend_exp.h', 'elo_update.h', 'Makefile')]
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)):
        subprocess.check_call(['make', '-C', ORACLE_DIR, '-s', '-B'], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_mt_new.restype = C.c_void_p
        L.orc_mt_free.argtypes = [C.c_void_p]
        L.orc_mt_seed.argtypes = [C.c_void_p, C.c_uint32]
        for f in ('orc_mt_py_random', 'orc_mt_np_sample'):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_mt_np_normal.restype = C.c_double
        L.orc_mt_np_normal.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_mt_np_choice.restype = C.c_int
        L.orc_mt_np_choice.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_normal_from_u32.restype = C.c_float
        L.orc_normal_from_u32.argtypes = [C.c_uint32]
        L.orc_phi_inverse_tail.restype = C.c_double
        L.orc_phi_inverse_tail.argtypes = [C.c_uint64, C.c_float]
        L.orc_normal53_tail.restype = C.c_double
        L.orc_normal53_tail.argtypes = [C.c_uint64]
        L.orc_run.restype = C.c_int
        L.orc_run.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcDrivers), C.POINTER(C.c_double), C.c_int32,
                              C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p,
                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                              C.POINTER(OrcTrace)]
        L.orc_simulate_race.restype = C.c_int
        L.orc_simulate_race.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcDrivers), C.POINTER(C.c_uint8),
                                        C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p,
                                        C.POINTER(C.c_uint8)]
        L.orc_sample_grid_mt.restype = C.c_int
        L.orc_sample_grid_mt.argtypes = [C.POINTER(C.c_double), C.c_int32, C.c_void_p, C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


class MTState:
    """The two global Mersenne-Twister streams of the reference process."""

    def __init__(self, seed=None):
        self.h = lib().orc_mt_new()
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        lib().orc_mt_seed(self.h, seed)

    def __del__(self):
        try:
            lib().orc_mt_free(self.h)
        except Exception:
            pass


_cases = None


def load_cases():
    global _cases
    if _cases is None:
        with open(os.path.join(GOLDEN_DIR, 'cases.json')) as f:
            _cases = json.load(f)
    return _cases


def load_case(name):
    return load_cases()['cases'][name]


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Problem:
    """A case dict (the arguments of RaceSimulator.run_monte_carlo) resolved to dense arrays.

    Resolves the reference's dict defaults exactly where the reference applies
    them (src/simulation.py:190-193,202-204,286,294-296,317-325,454-458,514).
    """

    def __init__(self, case, set_pop=None):
        cfg = case['config']
        self.drivers = list(case['grid_probs'].keys())
        n = self.n = len(self.drivers)
        self.total_laps = cfg['total_laps']
        set_pop = set_pop or load_cases()['set_pop']
        c = self.cfg = OrcConfig()
        c.total_laps = cfg['total_laps']
        c.track_condition = TRACK_ID[case.get('track_condition', 'dry')]
        for k in ('pit_loss', 'overtake_delta', 'sc_probability', 'vsc_probability', 'red_flag_probability',
                  'drs_delta'):
            setattr(c, k, float(cfg[k]))
        c.dirty_air_threshold = float(cfg.get('dirty_air_threshold', 2.0))
        c.dirty_air_penalty = float(cfg.get('dirty_air_penalty', 0.5))
        for name, i in COMPOUND_ID.items():
            info = cfg['tire_compounds'].get(name, {})
            c.comp_pace_delta[i] = float(info.get('pace_delta', 0))
            c.comp_deg_rate[i] = float(info.get('deg_rate', 0.05))
            c.comp_optimal_laps[i] = int(info.get('optimal_laps', 30))
        c.pop_soft_hard = COMPOUND_ID[set_pop['SOFT_HARD']]
        c.pop_medium_hard = COMPOUND_ID[set_pop['MEDIUM_HARD']]

        ddr = case.get('driver_dnf_rates') or {}
        team_rate = [cfg['dnf_rates'].get(cfg['driver_teams'].get(d, 'Unknown'), 0.002) for d in self.drivers]
        self.arr = dict(
            base_pace=np.array([case['base_pace'].get(d, 90.0) for d in self.drivers], np.float64),
            tire_deg=np.array([case['tire_deg'].get(d, 0.05) for d in self.drivers], np.float64),
            tire_deg_pit=np.array([case['tire_deg'].get(d, 0.0) for d in self.drivers], np.float64),
            variance=np.array([case['driver_variance'].get(d, 0.15) for d in self.drivers], np.float64),
            team_dnf=np.array(team_rate, np.float64),
            lap_dnf=np.array([ddr.get(d, team_rate[i]) for i, d in enumerate(self.drivers)], np.float64),
        )
        self.drv = OrcDrivers(**{k: _dptr(v) for k, v in self.arr.items()})
        self.grid_probs = np.ascontiguousarray(
            np.array([case['grid_probs'][d] for d in self.drivers], np.float64))
        assert self.grid_probs.shape == (n, n)

    def run(self, n_sims, rng=RNG_PHILOX, seed=0, sim_offset=0, mt=None, want_orders=False,
            want_grids=False, n_trace=0):
        n, L = self.n, self.total_laps
        hist = np.zeros((n, n), np.uint64)
        orders = np.zeros((n_sims, n), np.uint8) if want_orders else None
        grids = np.zeros((n_sims, n), np.uint8) if want_grids else None
        trace = None
        tr = None
        if n_trace:
            trace = dict(cum=np.zeros((n_trace, L, n)), tbl=np.zeros((n_trace, L, n)),
                         last=np.zeros((n_trace, L, n)),
                         age=np.zeros((n_trace, L, n), np.int16), dnf_lap=np.zeros((n_trace, L, n), np.int16),
                         comp=np.zeros((n_trace, L, n), np.uint8), used=np.zeros((n_trace, L, n), np.uint8),
                         dnf=np.zeros((n_trace, L, n), np.uint8), drs=np.zeros((n_trace, L, n), np.uint8))
            tr = OrcTrace(n_trace=n_trace)
            for k, v in trace.items():
                setattr(tr, k, v.ctypes.data_as(dict(OrcTrace._fields_)[k]))
        if rng == RNG_MT:
            if mt is None:
                mt = MTState(seed)
            mt_h = mt.h
        else:
            mt_h = None
        rc = lib().orc_run(
            C.byref(self.cfg), C.byref(self.drv), _dptr(self.grid_probs), n, n_sims, sim_offset, seed, rng, mt_h,
            hist.ctypes.data_as(C.POINTER(C.c_uint64)),
            orders.ctypes.data_as(C.POINTER(C.c_uint8)) if want_orders else None,
            grids.ctypes.data_as(C.POINTER(C.c_uint8)) if want_grids else None,
            C.byref(tr) if tr is not None else None)
        if rc != 0:
            raise ValueError(f'orc_run failed: {rc}')
        out = dict(hist=hist.astype(np.int64))
        if want_orders:
            out['orders'] = orders
        if want_grids:
            out['grids'] = grids
        if trace:
            out['trace'] = trace
        return out

    def simulate_race(self, grid, rng=RNG_PHILOX, seed=0, sim_id=0, mt=None):
        g = np.ascontiguousarray(np.array(grid, np.uint8))
        order = np.zeros(self.n, np.uint8)
        rc = lib().orc_simulate_race(C.byref(self.cfg), C.byref(self.drv), g.ctypes.data_as(C.POINTER(C.c_uint8)),
                                     self.n, sim_id, seed, rng, mt.h if mt is not None else None,
                                     order.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            raise ValueError(f'orc_simulate_race failed: {rc}')
        return order


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)
