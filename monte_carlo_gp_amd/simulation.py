"""Monte Carlo race simulation on MI355X -- host side of the drop-in boundary.

Mirrors the reference's interface for the hot path (reference src/simulation.py):

    RaceConfig                                   :37-52   same fields, same defaults
    RaceSimulator(config)                        :55-57
    RaceSimulator.run_monte_carlo(...)           :59-100  same arguments, same result shape
    RaceSimulator.simulate_race(grid, ...)       :147-242 one race, list of (driver, position)

The per-lap loop itself runs in hand-written HIP (csrc/race_kernel_reg.hip.h) behind
the C ABI of include/mcgp.h; this module only resolves the reference's dict
defaults into dense arrays, calls the library through ctypes and reshapes the
integer histogram into the reference's `dict[driver][position] -> probability`.
There is no CPU path: without the HIP library or a GPU the calls raise.

Randomness: the reference seeds two global Mersenne-Twister streams
(:76-78); here every draw is a pure function of (seed, simulation id, lap,
purpose, index) under Philox4x32-10, so results are reproducible for a given
seed on any number of GPUs.  `seed=None` draws a 64-bit seed from Python's
global `random`, which keeps a globally seeded backtest reproducible the way
reference src/validation.py:172-174 relies on.
"""
from __future__ import annotations

import ctypes as C
import random
from dataclasses import dataclass, field

import numpy as np

from . import _native as N


@dataclass
class CarState:
    """The reference's per-car record (:9-34), kept for callers that import it: same fields, same defaults, same
    __post_init__.  Nothing here computes on it -- on the device a car is a binary64 `cumulative_time` and one packed
    word (grid slot, tyre age or retirement lap, driver, compound, dirty-air / DRS / retired flags, dry compounds used:
    csrc/race_kernel_reg.hip.h) in registers, and its `last_lap_time` a row of LDS; `position`, `pit_stops`,
    `laps_completed`, `team` and `fuel_load` are inert or derivable in the reference's loop."""
    driver: str
    team: str
    position: int
    lap: int
    tire_compound: str
    tire_age: int
    fuel_load: float
    time_behind_leader: float
    pit_stops: int
    cumulative_time: float = 0.0
    drs_enabled: bool = False
    dnf: bool = False
    used_compounds: set = field(default_factory=set)
    laps_completed: int = 0
    last_lap_time: float = 0.0

    def __post_init__(self):
        self.used_compounds.add(self.tire_compound)          # the starting compound counts as used (:31-34)


@dataclass
class RaceConfig:
    """Same fields and defaults as the reference's RaceConfig (:37-52)."""
    total_laps: int
    pit_loss: float
    overtake_delta: float
    sc_probability: float
    vsc_probability: float
    red_flag_probability: float
    dnf_rates: dict
    drs_zones: int
    drs_delta: float
    tire_compounds: dict
    driver_teams: dict
    dirty_air_threshold: float = 2.0
    dirty_air_penalty: float = 0.5


# `available.pop()` at reference :486,488 picks from a two-string set; CPython's answer
# depends on PYTHONHASHSEED.  These are the outcomes under PYTHONHASHSEED=0 (the setting the
# golden fixtures were made with, tests/golden/cases.json "set_pop").
DEFAULT_SET_POP = {'SOFT_HARD': 'HARD', 'MEDIUM_HARD': 'MEDIUM'}


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _Problem:
    """run_monte_carlo's arguments resolved to the dense tables of include/mcgp.h."""

    def __init__(self, config: RaceConfig, drivers, base_pace, tire_deg, driver_variance,
                 driver_dnf_rates, track_condition, set_pop, deviates=32):
        if track_condition not in N.TRACK_ID:
            raise ValueError(f"track_condition must be 'dry', 'damp' or 'wet', got {track_condition!r}")
        self.drivers = [str(d) for d in drivers]
        n = self.n = len(self.drivers)
        c = self.cfg = N.McgpConfig()
        c.total_laps = int(config.total_laps)
        c.track_condition = N.TRACK_ID[track_condition]
        c.pit_loss = float(config.pit_loss)
        c.overtake_delta = float(config.overtake_delta)
        c.sc_probability = float(config.sc_probability)
        c.vsc_probability = float(config.vsc_probability)
        c.red_flag_probability = float(config.red_flag_probability)
        c.drs_delta = float(config.drs_delta)
        c.dirty_air_threshold = float(config.dirty_air_threshold)
        c.dirty_air_penalty = float(config.dirty_air_penalty)
        for name, i in N.COMPOUND_ID.items():
            info = config.tire_compounds.get(name, {})        # reference :317,454
            c.comp_pace_delta[i] = float(info.get('pace_delta', 0))
            c.comp_deg_rate[i] = float(info.get('deg_rate', 0.05))
            c.comp_optimal_laps[i] = int(info.get('optimal_laps', 30))
        c.pop_soft_hard = N.COMPOUND_ID[set_pop['SOFT_HARD']]
        c.pop_medium_hard = N.COMPOUND_ID[set_pop['MEDIUM_HARD']]
        if deviates not in (32, 53):
            raise ValueError(f'deviates must be 32 or 53, got {deviates!r}')
        c.deviates = 1 if deviates == 53 else 0

        driver_dnf_rates = driver_dnf_rates or {}
        team_rate = [config.dnf_rates.get(config.driver_teams.get(d, 'Unknown'), 0.002)   # :263,286
                     for d in self.drivers]
        f64 = np.float64
        self.arrays = dict(
            base_pace=np.array([base_pace.get(d, 90.0) for d in self.drivers], f64),           # :202
            tire_deg=np.array([tire_deg.get(d, 0.05) for d in self.drivers], f64),             # :203
            tire_deg_pit=np.array([tire_deg.get(d, 0.0) for d in self.drivers], f64),          # :458
            variance=np.array([driver_variance.get(d, 0.15) for d in self.drivers], f64),      # :204
            team_dnf=np.array(team_rate, f64),
            lap_dnf=np.array([driver_dnf_rates.get(d, team_rate[i])                           # :190-193
                              for i, d in enumerate(self.drivers)], f64),
        )
        self.drv = N.McgpDrivers(**{k: _dptr(v) for k, v in self.arrays.items()})


class RaceSimulator:
    """Drop-in for the reference's RaceSimulator (:55-560) with the race loop on the GPU."""

    def __init__(self, config: RaceConfig, device=0, set_pop: dict | None = None, deviates: int = 32):
        """`deviates`: 32 (default) -- uniforms w / 2^32 and normals from a binary32 cubic table, the fast path -- or 53:
        the reference's width, 53-bit uniforms and binary64 normals (reference :137,194,302,330,524), every draw keeping
        the 32-bit mode's word as its leading bits; a priced option (include/mcgp.h: mcgp_config.deviates), every field
        size.

        `device`: a HIP device index (default 0), a list of indices, or 'all' (every visible device).  With more
        than one device a run_monte_carlo call is split by simulation id into contiguous shards, one host thread per
        device over the same C entry point, and the integer histograms are added on the host -- the whole node from the
        plain single-process call the reference's caller makes (reference src/predictor.py:264,283-291), with results
        identical to a one-device run (every draw is a function of the global simulation id).  An index may be
        listed more than once (shards then queue on that device).  The torchrun + RCCL path (distributed.py) is
        separate and unchanged."""
        self.config = config
        self.deviates = int(deviates)
        if isinstance(device, str):
            if device != 'all':
                raise ValueError(f"device must be an index, a list of indices or 'all', got {device!r}")
            count = N.lib().mcgp_device_count()
            if count < 1:
                raise N.McgpError(-2, 'no HIP device visible (this library has no CPU path)')
            self.devices = list(range(count))
        elif isinstance(device, (list, tuple)):
            if not device:
                raise ValueError('device list is empty')
            self.devices = [int(d) for d in device]
        else:
            self.devices = [int(device)]
        self.device = self.devices[0]        # single-device entry points (simulate_race, the front end) use the first
        self.set_pop = dict(set_pop or DEFAULT_SET_POP)
        self.last_histogram = None      # np.int64 [n, n], counts[driver][position-1] of the last run
        self.last_drivers = None
        self._race_inputs = None

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _grid_matrix(grid_probs, drivers):
        n = len(drivers)
        g = np.zeros((n, n), np.float64)
        for i, d in enumerate(drivers):
            row = grid_probs[d]
            m = min(len(row), n)             # `pos < len(grid_probs.get(d, []))` else 0, reference :120
            g[i, :m] = np.asarray(row[:m], np.float64)
        return np.ascontiguousarray(g)

    @staticmethod
    def _resolve_seed(seed):
        if seed is None:
            return random.getrandbits(64)
        seed = int(seed)
        if seed < 0:
            seed = -seed                     # random.seed() uses abs(seed)
        return seed & 0xFFFFFFFFFFFFFFFF

    def _problem(self, drivers, base_pace, tire_deg, driver_variance, driver_dnf_rates, track_condition):
        n = len(drivers)
        if n < 1 or n > N.MAX_CARS:
            raise ValueError(f'number of drivers must be in [1, {N.MAX_CARS}], got {n}')
        return _Problem(self.config, drivers, base_pace, tire_deg, driver_variance, driver_dnf_rates,
                        track_condition, self.set_pop, self.deviates)

    # ------------------------------------------------------------------ reference surface
    def run_monte_carlo(
        self,
        n_simulations: int,
        grid_probs: dict,
        base_pace: dict,
        tire_deg: dict,
        driver_variance: dict,
        driver_dnf_rates: dict | None = None,
        seed: int | None = None,
        track_condition: str = 'dry',
        sim_offset: int = 0,
        return_orders: bool = False,
    ):
        """Run n simulations and return position probability distributions (reference :59-100).

        Returns {driver: {position (1-based): probability}} holding only non-zero cells,
        like the reference.  Extra keyword arguments (not in the reference):
        sim_offset -- first global simulation id (for sharding a run over ranks);
        return_orders -- also return the [n_sims, n] finishing orders (driver index per position).
        """
        drivers = [str(d) for d in grid_probs.keys()]
        if not drivers or n_simulations <= 0:
            self.last_histogram, self.last_drivers = np.zeros((len(drivers),) * 2, np.int64), drivers
            return ({}, np.zeros((0, len(drivers)), np.uint8)) if return_orders else {}
        prob = self._problem(drivers, base_pace, tire_deg, driver_variance, driver_dnf_rates, track_condition)
        n = prob.n
        g = self._grid_matrix({str(k): v for k, v in grid_probs.items()}, drivers)
        orders = np.zeros((n_simulations, n), np.uint8) if return_orders else None
        seed64 = self._resolve_seed(seed)
        lib = N.lib()

        def run_shard(device, offset, count):
            h = np.zeros((n, n), np.uint64)
            o = orders[offset:offset + count] if return_orders else None       # contiguous rows of the caller's buffer
            rc = lib.mcgp_run(C.byref(prob.cfg), C.byref(prob.drv), _dptr(g), n, int(count), int(sim_offset) + int(offset),
                              seed64, device, h.ctypes.data_as(C.POINTER(C.c_uint64)),
                              o.ctypes.data_as(C.POINTER(C.c_uint8)) if return_orders else None)
            # (mcgp_last_error is thread-local: read it on the thread that made the call)
            return h, rc, (lib.mcgp_last_error().decode('utf-8', 'replace') if rc != 0 else '')

        if len(self.devices) == 1:
            parts = [run_shard(self.devices[0], 0, int(n_simulations))]
        else:
            from concurrent.futures import ThreadPoolExecutor
            from .distributed import shard_range
            world = len(self.devices)
            shards = [shard_range(int(n_simulations), k, world) for k in range(world)]
            with ThreadPoolExecutor(world) as ex:          # ctypes releases the GIL for the duration of the call
                parts = list(ex.map(lambda a: run_shard(a[0], *a[1]), zip(self.devices, shards)))
        for _, rc, msg in parts:
            if rc != 0:
                raise N.McgpError(rc, msg)
        hist = np.sum([h for h, _, _ in parts], axis=0, dtype=np.uint64)
        self.last_histogram = hist.astype(np.int64)
        self.last_drivers = drivers
        result = histogram_to_probs(self.last_histogram, drivers, n_simulations)
        return (result, orders) if return_orders else result

    def simulate_race(
        self,
        grid: list,
        base_pace: dict,
        tire_deg: dict,
        driver_variance: dict,
        driver_dnf_rates: dict | None = None,
        track_condition: str = 'dry',
        seed: int | None = None,
        sim_id: int = 0,
    ):
        """Simulate a single race from a fixed grid; returns [(driver, position)] (reference :147-242)."""
        drivers = [str(d) for d in grid]
        if not drivers:
            return []
        prob = self._problem(drivers, base_pace, tire_deg, driver_variance, driver_dnf_rates, track_condition)
        n = prob.n
        g = np.arange(n, dtype=np.uint8)           # driver index == grid slot here
        order = np.zeros(n, np.uint8)
        N.check(N.lib().mcgp_simulate_race(
            C.byref(prob.cfg), C.byref(prob.drv), g.ctypes.data_as(C.POINTER(C.c_uint8)), n, int(sim_id),
            self._resolve_seed(seed), self.device, order.ctypes.data_as(C.POINTER(C.c_uint8))))
        return [(drivers[int(d)], p + 1) for p, d in enumerate(order)]

    # ------------------------------------------------------------------ device grid-probability front end
    @staticmethod
    def front_end_arrays(drivers, quali_ratings, quali_features=None, penalties=None, initial_rating=1500.0):
        """Dense inputs of the device front end with the reference's .get() defaults resolved
        (src/elo.py:131 initial rating; src/predictor.py:335,352-354 features default 0; :386-390 penalties)."""
        from .predictor import _penalty_value
        quali_features, penalties = quali_features or {}, penalties or {}
        f = lambda key: np.array([quali_features.get(d, {}).get(key, 0) for d in drivers], np.float64)
        rating = np.array([quali_ratings.get(d, initial_rating) for d in drivers], np.float64)
        pen = np.array([int(_penalty_value(penalties.get(d, 0))) for d in drivers], np.int32)
        return rating, f('teammate_delta'), f('form_score'), f('circuit_affinity'), pen

    def grid_probs_on_device(self, drivers, quali_ratings, quali_features=None, penalties=None):
        """{driver: [P(grid slot)]} computed by the device front end (include/mcgp.h: mcgp_grid_probs)."""
        drivers = [str(d) for d in drivers]
        n = len(drivers)
        r, td, fs, ca, pen = self.front_end_arrays(drivers, quali_ratings, quali_features, penalties)
        out = np.zeros((n, n), np.float64)
        N.check(N.lib().mcgp_grid_probs(_dptr(r), _dptr(td), _dptr(fs), _dptr(ca),
                                        pen.ctypes.data_as(C.POINTER(C.c_int32)), n, self.device, _dptr(out)))
        return {d: [float(x) for x in out[i]] for i, d in enumerate(drivers)}

    def run_from_ratings(self, n_simulations, drivers, quali_ratings, quali_features, penalties, base_pace, tire_deg,
                         driver_variance, driver_dnf_rates=None, seed=None, track_condition='dry', sim_offset=0):
        """run_monte_carlo with the grid matrix built ON THE DEVICE from the Elo quali ratings and features and
        handed to the race kernel without a host round trip (mcgp_run_from_ratings).
        Returns (position probabilities as run_monte_carlo does, {driver: [P(grid slot)]})."""
        drivers = [str(d) for d in drivers]
        prob = self._problem(drivers, base_pace, tire_deg, driver_variance, driver_dnf_rates, track_condition)
        n = prob.n
        r, td, fs, ca, pen = self.front_end_arrays(drivers, quali_ratings, quali_features, penalties)
        hist = np.zeros((n, n), np.uint64)
        grid = np.zeros((n, n), np.float64)
        N.check(N.lib().mcgp_run_from_ratings(
            C.byref(prob.cfg), C.byref(prob.drv), _dptr(r), _dptr(td), _dptr(fs), _dptr(ca),
            pen.ctypes.data_as(C.POINTER(C.c_int32)), n, int(n_simulations), int(sim_offset), self._resolve_seed(seed),
            self.device, hist.ctypes.data_as(C.POINTER(C.c_uint64)), _dptr(grid)))
        self.last_histogram = hist.astype(np.int64)
        self.last_drivers = drivers
        return (histogram_to_probs(self.last_histogram, drivers, n_simulations),
                {d: [float(x) for x in grid[i]] for i, d in enumerate(drivers)})

    # ------------------------------------------------------------------ north-star alias
    def set_race_inputs(self, base_pace=None, tire_deg=None, driver_variance=None, driver_dnf_rates=None,
                        track_condition='dry'):
        """Per-race inputs used by run_simulations(); missing dicts fall back to the reference defaults."""
        self._race_inputs = dict(base_pace=base_pace or {}, tire_deg=tire_deg or {},
                                 driver_variance=driver_variance or {}, driver_dnf_rates=driver_dnf_rates,
                                 track_condition=track_condition)
        return self

    def run_simulations(self, grid: dict, n_sims: int, seed: int | None = None):
        """run_simulations(grid, n_sims, seed): thin wrapper over run_monte_carlo (BASELINE north star)."""
        ri = self._race_inputs or dict(base_pace={}, tire_deg={}, driver_variance={}, driver_dnf_rates=None,
                                       track_condition='dry')
        return self.run_monte_carlo(n_sims, grid, ri['base_pace'], ri['tire_deg'], ri['driver_variance'],
                                    ri['driver_dnf_rates'], seed=seed, track_condition=ri['track_condition'])


def run_monte_carlo_batch(problems, n_simulations, device=0, set_pop=None):
    """Several races in ONE launch (include/mcgp.h: mcgp_run_batch): what a backtest of the reference does race after
    race (reference src/validation.py:179-185, 10 000 simulations each, src/predictor.py:284) -- at that size a launch
    is as long as one race of one lane, and a season fits beside itself on the device.

    `problems`: a list of dicts with the keys `config` (RaceConfig) and run_monte_carlo's arguments `grid_probs`,
    `base_pace`, `tire_deg`, `driver_variance`, and optionally `driver_dnf_rates`, `seed`, `track_condition`,
    `sim_offset`, `deviates` (32 or 53).  Returns a list of (probabilities as run_monte_carlo returns them, integer
    histogram [n, n]), one per problem and bit-identical to running it alone.  Races of different field sizes go into
    one launch per size; a problem the shared launch does not take (deviates = 53, or one only the generic kernel
    serves) runs by itself inside the same call -- no problem can make another one fail."""
    set_pop = dict(set_pop or DEFAULT_SET_POP)
    lib = N.lib()
    n_simulations = int(n_simulations)
    prepared = []
    for pr in problems:
        drivers = [str(d) for d in pr['grid_probs'].keys()]
        if not (1 <= len(drivers) <= N.MAX_CARS):
            raise ValueError(f'number of drivers must be in [1, {N.MAX_CARS}], got {len(drivers)}')
        prob = _Problem(pr['config'], drivers, pr['base_pace'], pr['tire_deg'], pr['driver_variance'],
                        pr.get('driver_dnf_rates'), pr.get('track_condition', 'dry'), set_pop, pr.get('deviates', 32))
        g = RaceSimulator._grid_matrix({str(k): v for k, v in pr['grid_probs'].items()}, drivers)
        prepared.append((prob, g, RaceSimulator._resolve_seed(pr.get('seed')), int(pr.get('sim_offset', 0))))
    out = [None] * len(prepared)
    if n_simulations <= 0:
        return [({}, np.zeros((p.n, p.n), np.int64)) for p, _, _, _ in prepared]
    by_n = {}
    for i, item in enumerate(prepared):
        by_n.setdefault(item[0].n, []).append(i)
    for n, idx in by_n.items():
        k = len(idx)
        cfgs = (N.McgpConfig * k)(*[prepared[i][0].cfg for i in idx])
        drvs = (N.McgpDrivers * k)(*[prepared[i][0].drv for i in idx])
        grids = (C.POINTER(C.c_double) * k)(*[_dptr(prepared[i][1]) for i in idx])
        seeds = (C.c_uint64 * k)(*[prepared[i][2] for i in idx])
        offsets = (C.c_uint64 * k)(*[prepared[i][3] for i in idx])
        hist = np.zeros((k, n, n), np.uint64)
        N.check(lib.mcgp_run_batch(k, cfgs, drvs, grids, n, n_simulations, offsets, seeds, int(device),
                                   hist.ctypes.data_as(C.POINTER(C.c_uint64))))
        for j, i in enumerate(idx):
            h = hist[j].astype(np.int64)
            out[i] = (histogram_to_probs(h, prepared[i][0].drivers, n_simulations), h)
    return out


def histogram_to_probs(hist, drivers, n_simulations):
    """counts[driver][position-1] -> {driver: {position: count / n}} with zero cells omitted (:97-100)."""
    out = {}
    for i, d in enumerate(drivers):
        row = hist[i]
        nz = np.nonzero(row)[0]
        out[d] = {int(p) + 1: int(row[p]) / n_simulations for p in nz}
    return out
