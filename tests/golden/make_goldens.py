#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run only in the build container (the reference lives at /root/reference and
never travels to the GPU box):

    PYTHONHASHSEED=0 PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

PYTHONHASHSEED=0 matters: `_handle_pit_stops` calls `set.pop()` on a
two-string set (reference src/simulation.py:486,488), whose result depends on
the string hash seed.  The outcomes observed under hashseed 0
are recorded in cases.json["set_pop"] and fed to the oracle / the HIP path as
an explicit rule.

What is written (all data, no reference source text):
  cases.json            inputs of every case (config, per-driver dicts,
                        grid_probs) exactly as they were handed to
                        RaceSimulator.run_monte_carlo, plus meta
  <case>.npz            hist   [n, n] int64   counts[driver][position-1]
                        orders [m, n] uint8   driver index at finishing pos p
                        grids  [m, n] uint8   driver index on grid slot p
                        trace_* per-lap state of the first T sims (after every
                        _update_positions call), cars in driver-index order
  mt_streams.npz        G4: raw draws of the two Mersenne-Twister streams
  sample_grid.npz       G5: _sample_grid outputs
  misc.json             G6: _predict_quali matrix inputs/outputs,
                        _create_race_config fields, brier_score values
  weekend.json          predict_weekend's glue: the arguments it hands to
                        run_monte_carlo for two offline weekends
  fuzz_cases.json,      72 random configurations over the whole input space (field
  fuzz.npz              size 2..24, rain, missing dict entries, event storms, ...) and the
                        reference's finishing orders / sampled grids for 40 runs of each
  ref_stat_<case>.npz   big-N reference histograms (several seeds) for the
                        statistical link HIP-Philox ~ reference-MT
"""
import json
import os
import random
import sys
import time
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

assert os.environ.get('PYTHONHASHSEED') == '0', 'run with PYTHONHASHSEED=0'
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
# predictor.py / validation.py import fastf1 at module scope only for the
# data loader; an inert module object is enough to import them.
sys.modules.setdefault('fastf1', types.ModuleType('fastf1'))

from src.simulation import RaceSimulator, RaceConfig  # noqa: E402
from src.config import DRIVER_TEAMS, TIRE_COMPOUNDS, DEFAULT_DNF_RATES, CIRCUITS  # noqa: E402
from src.predictor import F1Predictor  # noqa: E402
from src.elo import F1EloSystem  # noqa: E402
from src.validation import brier_score, podium_accuracy, calibration_analysis  # noqa: E402

COMPOUND_ID = {'SOFT': 0, 'MEDIUM': 1, 'HARD': 2, 'INTERMEDIATE': 3, 'WET': 4}


def make_predictor(ratings):
    p = F1Predictor.__new__(F1Predictor)
    p.elo_system = F1EloSystem()
    p.elo_system.ratings = {d: {'quali': r, 'race': r} for d, r in ratings.items()}
    return p


def elo_grid(drivers):
    """quali = race = 1700 - 20 i, empty feature dicts."""
    p = make_predictor({d: 1700.0 - 20.0 * i for i, d in enumerate(drivers)})
    g = p._predict_quali(drivers, {})
    return {d: [float(x) for x in g[d]] for d in drivers}


def config_dict(laps, pit_loss, overtake_delta, sc=0.01, vsc=0.015, red=0.002,
                tire_compounds=None, dnf_rates=None, driver_teams=None, drs_delta=0.3,
                drs_zones=2, dirty_thr=2.0, dirty_pen=0.5):
    return dict(
        total_laps=laps, pit_loss=pit_loss, overtake_delta=overtake_delta,
        sc_probability=sc, vsc_probability=vsc, red_flag_probability=red,
        dnf_rates=dict(dnf_rates if dnf_rates is not None else DEFAULT_DNF_RATES),
        drs_zones=drs_zones, drs_delta=drs_delta,
        tire_compounds={k: dict(v) for k, v in (tire_compounds or TIRE_COMPOUNDS).items()},
        driver_teams=dict(driver_teams if driver_teams is not None else DRIVER_TEAMS),
        dirty_air_threshold=dirty_thr, dirty_air_penalty=dirty_pen,
    )


def canonical_case(name, laps, circuit, n_sims, n_trace, seed=42):
    drivers = list(DRIVER_TEAMS.keys())
    c = CIRCUITS[circuit]
    return dict(
        name=name, n_sims=n_sims, n_orders=min(n_sims, 256), n_trace=n_trace, seed=seed,
        track_condition='dry',
        config=config_dict(laps, c['pit_loss'], c['overtake_delta'], drs_zones=c['drs_zones']),
        grid_probs=elo_grid(drivers),
        base_pace={d: 90.0 + 0.1 * i for i, d in enumerate(drivers)},
        tire_deg={d: 0.05 for d in drivers},
        driver_variance={d: 0.15 * 1.2 for d in drivers},
        driver_dnf_rates={d: 0.05 / laps for d in drivers},
    )


def build_cases():
    cases = []
    # S60 / S78: the two bench configurations of SURVEY.md 8(d).
    cases.append(canonical_case('S60', 60, 'Bahrain', 2000, 24))
    cases.append(canonical_case('S78', 78, 'Monaco', 2000, 16))
    # S50: Q13-sensitive race length (MEDIUM starters first pit with 20 < remaining <= 30).
    cases.append(canonical_case('S50', 50, 'Saudi Arabia', 2000, 24))

    # EVT: event storm (SC / VSC / red flag on most laps), short race, strong DNF rates
    # so that lap-1 double retirements (Q17) and empty-field corner cases show up.
    drivers = list(DRIVER_TEAMS.keys())
    evt = canonical_case('EVT', 34, 'Italy', 1500, 48, seed=7)
    evt['config'].update(sc_probability=0.12, vsc_probability=0.15, red_flag_probability=0.06)
    evt['config']['dnf_rates'] = {t: r * 12 for t, r in DEFAULT_DNF_RATES.items()}
    evt['driver_dnf_rates'] = None  # team rates on every lap (reference :190-193)
    evt['driver_variance'] = {d: 0.15 for d in drivers}
    cases.append(evt)

    # HET: heterogeneous per-driver inputs, a 21-car field with an unknown driver,
    # missing dict entries (exercises every .get default of the reference),
    # high / low degradation branches of the pit logic (Q12), dynamic compound table,
    # one-hot actual grid with penalties piled on the last slot (Q18).
    rng = random.Random(1234)
    drv = list(DRIVER_TEAMS.keys()) + ['XXX']
    n = len(drv)
    p = make_predictor({})
    onehot = {}
    order = drv[:]
    rng.shuffle(order)
    for slot, d in enumerate(order):
        v = [0.0] * n
        v[slot] = 1.0
        onehot[d] = v
    onehot = {d: onehot[d] for d in drv}
    onehot = p._adjust_for_penalties(onehot, {order[2]: 'engine', order[5]: 'full_pu',
                                              order[9]: 5, order[11]: 'pitlane_start', order[15]: 10})
    onehot = {d: [float(x) for x in onehot[d]] for d in drv}
    tc = {k: dict(v) for k, v in TIRE_COMPOUNDS.items()}
    tc['SOFT']['pace_delta'] = -0.63
    tc['HARD']['pace_delta'] = 0.41
    het = dict(
        name='HET', n_sims=1500, n_orders=256, n_trace=24, seed=2024, track_condition='dry',
        config=config_dict(57, 21.0, 0.6, tire_compounds=tc),
        grid_probs=onehot,
        base_pace={d: 88.0 + 2.5 * rng.random() for d in drv if d not in ('HUL',)},
        tire_deg={d: rng.choice([0.01, 0.015, 0.03, 0.05, 0.06, 0.09, 0.15]) for d in drv if d not in ('BOR', 'XXX')},
        driver_variance={d: 0.05 + 0.25 * rng.random() for d in drv if d not in ('OCO',)},
        driver_dnf_rates={d: (0.02 + 0.1 * rng.random()) / 57 for d in drv if d not in ('BEA', 'VER')},
    )
    cases.append(het)

    # DMP / WET: rain conditions (INTERMEDIATE / WET start tyres, no 2-compound rule).
    dmp = canonical_case('DMP', 57, 'Bahrain', 1000, 8, seed=11)
    dmp['track_condition'] = 'damp'
    dmp['config'].update(sc_probability=0.04, vsc_probability=0.04, red_flag_probability=0.02)
    cases.append(dmp)
    wet = canonical_case('WET', 44, 'Belgium', 1000, 8, seed=12)
    wet['track_condition'] = 'wet'
    wet['config'].update(sc_probability=0.04, vsc_probability=0.04, red_flag_probability=0.02)
    cases.append(wet)

    # N10: small field (10 cars), long race so backmarkers get lapped-distance gaps
    # and a retired car can sit between runners in the overtake sort (Q15).
    d10 = list(DRIVER_TEAMS.keys())[::2]
    n10 = dict(
        name='N10', n_sims=1500, n_orders=256, n_trace=16, seed=5, track_condition='dry',
        config=config_dict(72, 20.0, 0.5),
        grid_probs=elo_grid(d10),
        base_pace={d: 80.0 + 0.6 * i for i, d in enumerate(d10)},
        tire_deg={d: 0.04 + 0.004 * i for i, d in enumerate(d10)},
        driver_variance={d: 0.3 for d in d10},
        driver_dnf_rates={d: 0.6 / 72 for d in d10},
    )
    cases.append(n10)
    return cases


class TracingSimulator(RaceSimulator):
    """Reference simulator + observers.  Only wraps calls; draws no randomness."""

    def __init__(self, config, drivers, n_trace, n_orders):
        super().__init__(config)
        self.drivers = drivers
        self.index = {d: i for i, d in enumerate(drivers)}
        self.n_trace = n_trace
        self.n_orders = n_orders
        self.sim = -1
        self.grids, self.orders, self.traces = [], [], []
        self._cur = None

    def simulate_race(self, grid, *a, **k):
        self.sim += 1
        if self.sim < self.n_orders:
            self.grids.append([self.index[str(d)] for d in grid])
        self._cur = [] if self.sim < self.n_trace else None
        res = super().simulate_race(grid, *a, **k)
        if self.sim < self.n_orders:
            self.orders.append([self.index[str(d)] for d, _ in res])
        if self._cur is not None:
            self.traces.append(self._cur)
        self._cur = None
        return res

    def _update_positions(self, cars, lap=3, drs_disabled=False):
        cars = super()._update_positions(cars, lap=lap, drs_disabled=drs_disabled)
        if self._cur is not None:
            n = len(self.drivers)
            snap = dict(cum=np.zeros(n), tbl=np.zeros(n), last=np.zeros(n), age=np.zeros(n, np.int16),
                        comp=np.zeros(n, np.uint8), used=np.zeros(n, np.uint8), dnf=np.zeros(n, np.uint8),
                        drs=np.zeros(n, np.uint8), dnf_lap=np.zeros(n, np.int16))
            for c in cars:
                i = self.index[str(c.driver)]
                snap['cum'][i] = c.cumulative_time
                snap['tbl'][i] = c.time_behind_leader
                snap['last'][i] = c.last_lap_time
                snap['age'][i] = c.tire_age
                snap['comp'][i] = COMPOUND_ID[c.tire_compound]
                snap['used'][i] = sum(1 << COMPOUND_ID[x] for x in c.used_compounds)
                snap['dnf'][i] = c.dnf
                snap['drs'][i] = c.drs_enabled
                snap['dnf_lap'][i] = c.lap if c.dnf else 0
            self._cur.append(snap)
        return cars


def run_case(case):
    cfg = RaceConfig(**case['config'])
    drivers = list(case['grid_probs'].keys())
    n = len(drivers)
    sim = TracingSimulator(cfg, drivers, case['n_trace'], case['n_orders'])
    t0 = time.time()
    probs = sim.run_monte_carlo(
        case['n_sims'], case['grid_probs'], case['base_pace'], case['tire_deg'],
        case['driver_variance'], case['driver_dnf_rates'], seed=case['seed'],
        track_condition=case['track_condition'])
    dt = time.time() - t0
    hist = np.zeros((n, n), np.int64)
    for d, dist in probs.items():
        for pos, p in dist.items():
            c = p * case['n_sims']
            assert abs(c - round(c)) < 1e-6
            hist[sim.index[str(d)], pos - 1] = int(round(c))
    assert hist.sum() == n * case['n_sims']
    L = case['config']['total_laps']
    out = dict(hist=hist, orders=np.array(sim.orders, np.uint8), grids=np.array(sim.grids, np.uint8))
    for key in ('cum', 'tbl', 'last', 'age', 'comp', 'used', 'dnf', 'drs', 'dnf_lap'):
        arr = np.array([[snap[key] for snap in tr] for tr in sim.traces])
        assert arr.shape == (case['n_trace'], L, n), arr.shape
        out['trace_' + key] = arr
    np.savez_compressed(os.path.join(HERE, case['name'] + '.npz'), **out)
    print(f"{case['name']}: {case['n_sims']} sims in {dt:.1f}s ({case['n_sims']/dt:.0f}/s); "
          f"wins {[int(x) for x in hist[:6, 0]]}", flush=True)
    return dt


def mt_streams():
    """G4: pins the RNG replay (CPython `random`, numpy legacy RandomState)."""
    out = {}
    for seed in (42, 0, 2 ** 32 - 1):
        random.seed(seed)
        out[f'py_random_{seed}'] = np.array([random.random() for _ in range(4096)])
        np.random.seed(seed)
        out[f'np_sample_{seed}'] = np.array([np.random.random_sample() for _ in range(1024)])
        np.random.seed(seed)
        out[f'np_normal_{seed}'] = np.array([np.random.normal(0, 1.0) for _ in range(1025)])
        # interleaved choice / normal exactly as the simulator issues them: the
        # gauss cache must survive the choice() calls.
        np.random.seed(seed)
        mix = []
        p = np.array([0.1, 0.0, 0.25, 0.05, 0.6])
        for i in range(600):
            mix.append(float(np.random.choice(5, p=p)))
            mix.append(np.random.normal(0, 0.18))
            if i % 3 == 0:
                mix.append(np.random.normal(0, 1.5))
        out[f'np_mix_{seed}'] = np.array(mix)
    np.savez_compressed(os.path.join(HERE, 'mt_streams.npz'), **out)


def sample_grids(cases):
    """G5: 1000 `_sample_grid` results for the Elo grid and the one-hot+penalty grid."""
    out = {}
    for case in cases:
        if case['name'] not in ('S60', 'HET', 'N10'):
            continue
        sim = RaceSimulator(RaceConfig(**case['config']))
        drivers = list(case['grid_probs'].keys())
        idx = {d: i for i, d in enumerate(drivers)}
        np.random.seed(99)
        random.seed(99)
        out[case['name']] = np.array(
            [[idx[str(d)] for d in sim._sample_grid(case['grid_probs'])] for _ in range(1000)], np.uint8)
    np.savez_compressed(os.path.join(HERE, 'sample_grid.npz'), **out)


def misc():
    """G6: host-side helpers either side of the path ("next" rows of SURVEY 8f)."""
    drivers = list(DRIVER_TEAMS.keys())
    ratings = {d: 1700.0 - 20.0 * i for i, d in enumerate(drivers)}
    p = make_predictor(ratings)
    feats = {d: {'teammate_delta': 0.3 * ((i % 3) - 1), 'form_score': 0.1 * (i % 5) - 0.2,
                 'circuit_affinity': 0.05 * (i % 4)} for i, d in enumerate(drivers)}
    out = dict(
        drivers=drivers, ratings=ratings, features=feats,
        pole_probs=p.elo_system.predict_quali_probs(drivers),
        quali_plain=p._predict_quali(drivers, {}),
        quali_feat=p._predict_quali(drivers, feats),
        penalties={'VER': 'engine', 'NOR': 5, 'HAM': 'full_pu', 'ALB': 'gearbox'},
    )
    out['quali_penalised'] = p._adjust_for_penalties(out['quali_plain'], out['penalties'])
    out['apply_grid_penalties'] = p.apply_grid_penalties(
        {d: i + 1 for i, d in enumerate(drivers)}, out['penalties'])
    rc = p._create_race_config(p._get_circuit_info('Bahrain Grand Prix'))
    out['race_config_bahrain'] = [rc.total_laps, rc.pit_loss, rc.overtake_delta, rc.sc_probability,
                                  rc.vsc_probability, rc.red_flag_probability, rc.drs_delta]
    rc2 = p._create_race_config(p._get_circuit_info('Nowhere GP'))
    out['race_config_fallback'] = [rc2.total_laps, rc2.pit_loss, rc2.overtake_delta, rc2.drs_zones]
    uni = {d: 1.0 / 20 for d in drivers}
    out['brier_uniform'] = float(brier_score([uni], ['VER']))
    preds = [out['pole_probs'], uni, {d: (0.9 if d == 'NOR' else 0.1 / 19) for d in drivers}]
    acts = ['VER', None, 'NOR']
    out['brier_inputs'] = dict(preds=preds, actuals=acts)
    out['brier_mixed'] = float(brier_score(preds, acts))
    pod_pred = [{'podium_probabilities': {d: 1.0 / (i + 1) for i, d in enumerate(drivers)}},
                {'podium_probabilities': {d: float(i) for i, d in enumerate(drivers)}}]
    pod_act = [{'podium': ['VER', 'NOR', 'HAM']}, {'podium': ['BEA', 'OCO', 'VER']}]
    out['podium_inputs'] = dict(preds=pod_pred, actuals=pod_act)
    out['podium_accuracy'] = float(podium_accuracy(pod_pred, pod_act))
    # Elo update known answers
    e = F1EloSystem()
    e.set_recency_weight(0, 5, 24)
    e.update_quali_ratings([(d, 80.0 + 0.1 * ((i * 7) % 20)) for i, d in enumerate(drivers)])
    e.set_recency_weight(1)
    e.update_race_ratings([(d, ((i * 3) % 20) + 1) for i, d in enumerate(drivers)])
    out['elo_after'] = e.ratings
    # calibration_analysis (reference src/validation.py:133-158, sklearn's calibration_curve underneath): a season-sized
    # sample (10 bins), one with a race without winner and one without probabilities, and one too small for 10 bins
    rs = np.random.RandomState(7)

    def season(n_races, n_drivers):
        preds, acts = [], []
        for _ in range(n_races):
            w = rs.gamma(0.6, size=n_drivers)
            w = w / w.sum()
            ds = drivers[:n_drivers]
            preds.append({'win_probabilities': {d: float(x) for d, x in zip(ds, w)}})
            acts.append({'winner': ds[int(rs.choice(n_drivers, p=w))]})
        return preds, acts
    cal = {}
    for name, (n_races, n_drivers) in {'season': (24, 20), 'small': (3, 5), 'tiny': (1, 4)}.items():
        preds, acts = season(n_races, n_drivers)
        if name == 'season':
            acts[3] = {'winner': None}
            preds[7] = {'win_probabilities': {}}
            preds[11] = {}
        cal[name] = dict(preds=preds, actuals=acts, result=calibration_analysis(preds, acts))
    cal['empty'] = dict(preds=[], actuals=[], result=calibration_analysis([], []))
    out['calibration'] = cal
    # Elo updates from a complete ordering, for the backtest sweep's known-pairs update (reference src/elo.py:45-122)
    e2 = F1EloSystem()
    for i, d in enumerate(drivers):
        e2.ratings[d] = {'quali': 1700.0 - 20.0 * i, 'race': 1650.0 - 15.0 * i}
    e2.set_recency_weight(0, 9, 24)
    order = [drivers[(i * 7) % 20] for i in range(20)]
    e2.update_quali_ratings([(d, 80.0 + 0.05 * k) for k, d in enumerate(order)])      # fastest first
    e2.update_race_ratings([(d, k + 1) for k, d in enumerate(order)])
    out['elo_full_order'] = dict(order=order, race_index=9, total=24, after=e2.ratings)
    out['circuits'] = CIRCUITS
    out['driver_teams'] = DRIVER_TEAMS
    out['dnf_rates'] = DEFAULT_DNF_RATES
    out['tire_compounds'] = TIRE_COMPOUNDS
    with open(os.path.join(HERE, 'misc.json'), 'w') as f:
        json.dump(json.loads(json.dumps(out, default=float)), f, indent=0)


def weekend():
    """predict_weekend's numeric glue (reference src/predictor.py:186-319), captured offline.

    The reference predictor is driven with stub data sources (no FastF1): a synthetic practice
    DataFrame, canned feature dicts and weather.  RaceSimulator.run_monte_carlo is replaced by a
    recorder, so the fixture holds exactly what the orchestrator hands to the hot path, plus the
    practice-derived dicts (base_pace, tire_deg, tyre table) that an offline race fixture supplies.
    """
    import pandas as pd
    import src.predictor as predictor_mod
    rng = random.Random(7)
    drivers = list(DRIVER_TEAMS.keys())
    out = {}
    for label, rain, point, actual in (('dry_fp2', False, 'fp2', False), ('damp_quali', True, 'quali', True)):
        rows = []
        for i, d in enumerate(drivers):
            for lap in range(1, 13):
                comp = ['SOFT', 'MEDIUM', 'HARD'][(lap + i) % 3]
                t = 91.0 + 0.12 * i + 0.04 * lap + {'SOFT': -0.7, 'MEDIUM': 0.0, 'HARD': 0.5}[comp] + rng.gauss(0, 0.15)
                rows.append(dict(Driver=d, LapTime=t, LapNumber=lap, Compound=comp))
        fp = pd.DataFrame(rows)
        race_feats = {d: dict(clutch_factor=rng.uniform(-1, 1), dnf_probability=rng.uniform(0.02, 0.12),
                              team_trend=rng.uniform(-0.5, 0.5), wet_performance=rng.uniform(-1, 1)) for d in drivers}
        quali_feats = {d: dict(teammate_delta=rng.uniform(-0.4, 0.4), form_score=rng.uniform(-1, 1),
                               circuit_affinity=rng.uniform(-1, 1)) for d in drivers}
        weather = dict(rainfall=rain, track_temp=31.5)

        class Loader:
            def load_season_data(self, season):
                raise RuntimeError('offline')

            def load_session(self, season, race, session):
                return fp

            def get_weather(self, season, race, session):
                return weather

        class Features:
            def load_historical_data(self, seasons):
                pass

            def calculate_quali_features(self, d, race):
                return quali_feats[d]

            def calculate_race_features(self, d, race, w):
                return race_feats[d]

        p = F1Predictor.__new__(F1Predictor)
        p.data_loader, p.feature_engine = Loader(), Features()
        p.elo_system = F1EloSystem()
        p.elo_system.ratings = {d: {'quali': 1650.0 - 13.0 * i, 'race': 1600.0} for i, d in enumerate(drivers)}
        p._processed_seasons, p._features_loaded = set(), False
        captured = {}

        class Recorder:
            def __init__(self, config):
                captured['config'] = {k: getattr(config, k) for k in (
                    'total_laps', 'pit_loss', 'overtake_delta', 'sc_probability', 'vsc_probability',
                    'red_flag_probability', 'dnf_rates', 'drs_zones', 'drs_delta', 'tire_compounds',
                    'driver_teams', 'dirty_air_threshold', 'dirty_air_penalty')}

            def run_monte_carlo(self, **kw):
                captured['call'] = kw
                return {d: {i + 1: 1.0 if i == j else 0.0 for i in range(3)} for j, d in enumerate(drivers)}

        saved = predictor_mod.RaceSimulator
        predictor_mod.RaceSimulator = Recorder
        try:
            grid = {d: ((i * 7) % 20) + 1 for i, d in enumerate(drivers)} if actual else None
            res = p.predict_weekend(2024, 'Monaco Grand Prix' if rain else 'Bahrain Grand Prix',
                                    grid_penalties={'NOR': 'gearbox', 'HAM': 3}, prediction_point=point,
                                    actual_grid=grid)
        finally:
            predictor_mod.RaceSimulator = saved
        pace = p._extract_race_pace(fp)
        out[label] = dict(
            race='Monaco Grand Prix' if rain else 'Bahrain Grand Prix', prediction_point=point,
            grid_penalties={'NOR': 'gearbox', 'HAM': 3}, actual_grid=grid, weather=weather,
            drivers=drivers, quali_ratings={d: 1650.0 - 13.0 * i for i, d in enumerate(drivers)},
            race_features=race_feats, quali_features=quali_feats,
            practice=dict(base_pace=pace, tire_deg=p._extract_tire_deg(fp),
                          tire_compounds=p._extract_tire_compound_deltas(fp)),
            captured=captured,
            result={k: res[k] for k in ('pole_probabilities', 'win_probabilities', 'podium_probabilities',
                                        'prediction_point', 'confidence', 'grid_is_actual')},
        )
    with open(os.path.join(HERE, 'weekend.json'), 'w') as f:
        json.dump(json.loads(json.dumps(out, default=float)), f, indent=0)


def fuzz_cases(count=72, seed=20240601):
    """Random configurations across the whole input space of run_monte_carlo (small runs)."""
    rng = random.Random(seed)
    teams = list(DEFAULT_DNF_RATES.keys())
    cases = []
    for k in range(count):
        n = rng.choice([2, 3, 5, 8, 10, 12, 15, 18, 19, 20, 20, 20, 21, 22, 24])
        drivers = [f'D{i:02d}' for i in range(n)]
        laps = rng.choice([6, 9, 14, 22, 31, 44, 57, 66, 78, 90])
        tc = {}
        for name, info in TIRE_COMPOUNDS.items():
            if rng.random() < 0.1:
                continue                                    # missing compound: .get(compound, {}) defaults
            d = {}
            if rng.random() < 0.9:
                d['pace_delta'] = round(info['pace_delta'] + rng.uniform(-0.4, 0.4), 3)
            if rng.random() < 0.9:
                d['deg_rate'] = rng.choice([info['deg_rate'], 0.02, 0.05, 0.1, 0.0])
            if rng.random() < 0.9:
                d['optimal_laps'] = rng.choice([info['optimal_laps'], 3, 8, 12, 20, 33])
            tc[name] = d
        heavy = rng.random() < 0.35
        cfg = config_dict(
            laps, rng.choice([18.0, 21.0, 24.5, 30.0]), rng.choice([0.2, 0.4, 0.6, 0.9, 1.5]),
            sc=rng.choice([0.0, 0.01, 0.05, 0.3]) if heavy else 0.01,
            vsc=rng.choice([0.0, 0.015, 0.08, 0.3]) if heavy else 0.015,
            red=rng.choice([0.0, 0.002, 0.03, 0.2]) if heavy else 0.002,
            tire_compounds=tc,
            dnf_rates={t: DEFAULT_DNF_RATES[t] * rng.choice([1, 1, 5, 40]) for t in teams if rng.random() < 0.9},
            driver_teams={d: rng.choice(teams) for d in drivers if rng.random() < 0.9},
            drs_delta=rng.choice([0.0, 0.3, 0.55]), drs_zones=rng.randint(1, 4),
            dirty_thr=rng.choice([2.0, 0.8, 3.5]), dirty_pen=rng.choice([0.5, 0.2, 1.0]))
        style = rng.choice(['random', 'onehot', 'sparse', 'elo'])
        if style == 'elo':
            grid = elo_grid(drivers)
        else:
            grid = {}
            perm = drivers[:]
            rng.shuffle(perm)
            for d in drivers:
                if style == 'onehot':
                    row = [1.0 if perm[s] == d else 0.0 for s in range(n)]
                elif style == 'sparse':
                    row = [rng.random() if rng.random() < 0.35 else 0.0 for _ in range(n)]
                else:
                    row = [rng.random() ** 3 for _ in range(n)]
                grid[d] = row
            if style == 'sparse' and n > 2:                 # an all-zero column: uniform fallback (Q18)
                col = rng.randrange(n)
                for d in drivers:
                    grid[d][col] = 0.0
        spread = rng.choice([0.02, 0.1, 0.4])
        cases.append(dict(
            name=f'F{k:02d}', n_sims=40, n_orders=40, n_trace=0, seed=rng.randrange(2 ** 32),
            track_condition=rng.choice(['dry', 'dry', 'dry', 'damp', 'wet']), config=cfg, grid_probs=grid,
            base_pace={d: 70.0 + 30.0 * rng.random() * 0 + 85.0 * 0 + 88.0 + spread * i + rng.uniform(-0.2, 0.2)
                       for i, d in enumerate(drivers) if rng.random() < 0.92},
            tire_deg={d: rng.choice([0.0, 0.01, 0.02, 0.03, 0.05, 0.050000000000000003, 0.07, 0.12, -0.01])
                      for d in drivers if rng.random() < 0.9},
            driver_variance={d: rng.choice([0.05, 0.15, 0.18, 0.3, 0.6]) for d in drivers if rng.random() < 0.9},
            driver_dnf_rates=None if rng.random() < 0.25 else
            {d: rng.choice([0.0, 0.0008, 0.004, 0.03, 0.2]) for d in drivers if rng.random() < 0.85},
        ))
    return cases


def extreme_cases():
    """Corner configurations: certain events, everybody retiring, every pair attempting, 1-2 lap races."""
    out = []

    def mk(name, laps, **over):
        c = canonical_case(name, laps, 'Bahrain', 30, 0, seed=over.pop('seed', 3))
        c['n_orders'] = 30
        for k, v in over.items():
            if k in c['config']:
                c['config'][k] = v
            else:
                c[k] = v
        return c
    out.append(mk('X_onelap', 1))
    out.append(mk('X_twolaps', 2))
    out.append(mk('X_always_sc', 12, sc_probability=1.0, red_flag_probability=0.0))
    out.append(mk('X_always_vsc', 12, sc_probability=0.0, red_flag_probability=0.0, vsc_probability=1.0))
    out.append(mk('X_always_red', 40, red_flag_probability=1.0))
    out.append(mk('X_all_attempt', 15, overtake_delta=-50.0, drs_delta=5.0))
    out.append(mk('X_never_attempt', 15, overtake_delta=1e9))
    allout = mk('X_all_out_lap1', 10, dnf_rates={t: 0.25 for t in DEFAULT_DNF_RATES})
    out.append(allout)
    out.append(mk('X_all_out_lap2', 10, driver_dnf_rates={d: 1.0 for d in DRIVER_TEAMS}))
    out.append(mk('X_half_out', 20, driver_dnf_rates={d: 0.5 for d in DRIVER_TEAMS}, sc_probability=0.5))
    out.append(mk('X_no_noise', 25, driver_variance={d: 0.0 for d in DRIVER_TEAMS},
                  base_pace={d: 90.0 for d in DRIVER_TEAMS}))
    out.append(mk('X_pit_every_lap', 30, tire_compounds={k: dict(v, optimal_laps=0) for k, v in TIRE_COMPOUNDS.items()}))
    return out


def fuzz():
    """Reference finishing orders for the random configurations: fuzz_cases.json + fuzz.npz."""
    cases = fuzz_cases() + extreme_cases()
    out = {}
    for c in cases:
        cfg = RaceConfig(**c['config'])
        drivers = list(c['grid_probs'].keys())
        sim = TracingSimulator(cfg, drivers, 0, c['n_orders'])
        sim.run_monte_carlo(c['n_sims'], c['grid_probs'], c['base_pace'], c['tire_deg'], c['driver_variance'],
                            c['driver_dnf_rates'], seed=c['seed'], track_condition=c['track_condition'])
        out[c['name'] + '_orders'] = np.array(sim.orders, np.uint8)
        out[c['name'] + '_grids'] = np.array(sim.grids, np.uint8)
    np.savez_compressed(os.path.join(HERE, 'fuzz.npz'), **out)
    with open(os.path.join(HERE, 'fuzz_cases.json'), 'w') as f:
        json.dump({c['name']: c for c in cases}, f, indent=0)
    print(f'fuzz: {len(cases)} configurations', flush=True)


def ref_stat(case, seeds, n_each):
    """Big-N reference histogram (one process per seed) for the statistical link."""
    import multiprocessing as mp
    with mp.Pool(min(len(seeds), int(os.environ.get('MCGP_STAT_PROCS', '8')))) as pool:
        hs = pool.starmap(_ref_stat_one, [(case, s, n_each) for s in seeds])
    np.savez_compressed(os.path.join(HERE, f"ref_stat_{case['name']}.npz"),
                        hist=np.sum(hs, axis=0), per_seed=np.array(hs), seeds=np.array(seeds),
                        n_each=n_each)
    print(f"ref_stat {case['name']}: {len(seeds)}x{n_each}", flush=True)


def _ref_stat_one(case, seed, n_each):
    cfg = RaceConfig(**case['config'])
    drivers = list(case['grid_probs'].keys())
    idx = {d: i for i, d in enumerate(drivers)}
    probs = RaceSimulator(cfg).run_monte_carlo(
        n_each, case['grid_probs'], case['base_pace'], case['tire_deg'], case['driver_variance'],
        case['driver_dnf_rates'], seed=seed, track_condition=case['track_condition'])
    h = np.zeros((len(drivers), len(drivers)), np.int64)
    for d, dist in probs.items():
        for pos, p in dist.items():
            h[idx[str(d)], pos - 1] = int(round(p * n_each))
    return h


def elo_season():
    """Three seasons of Elo updates through the reference's F1EloSystem (src/elo.py:13-122): two past seasons at
    K x 0.7 and K x 1.0, the current one with the race-within-season K; a rookie who appears mid-season, absentees,
    races with fewer classified finishers than qualifiers, exact lap-time ties, a shared finishing position, a
    one-entry event (no update).  Ratings of every driver after every event -> elo_season.json."""
    drivers = list(DRIVER_TEAMS.keys())
    rookie = 'ROO'
    rs = np.random.RandomState(2024)
    skill = {d: 0.04 * i for i, d in enumerate(drivers + [rookie])}
    e = F1EloSystem()
    events, after = [], []

    def snapshot():
        after.append({d: dict(r) for d, r in e.ratings.items()})

    def weekend(field):
        times = {d: 80.0 + skill[d] + float(rs.normal(0, 0.25)) for d in field}
        quali = sorted(times.items(), key=lambda kv: kv[1])
        if rs.rand() < 0.3 and len(quali) > 3:                     # an exact tie on lap time
            quali[2] = (quali[2][0], quali[1][1])
        order = sorted(field, key=lambda d: skill[d] + float(rs.normal(0, 0.5)))
        classified = order[:len(order) - int(rs.randint(0, 4))]
        race = [(d, i + 1) for i, d in enumerate(classified)]
        if rs.rand() < 0.15 and len(race) > 5:                     # two cars classified in the same position
            race[4] = (race[4][0], race[3][1])
        rs.shuffle(race)                                            # list order is not finishing order
        return quali, race

    plan = [(2, 6, 6), (1, 6, 6), (0, 24, 24)]                     # (years_ago, races run, total_races)
    for years_ago, n_races, total in plan:
        for idx in range(n_races):
            field = [d for d in drivers if rs.rand() > 0.04]
            if years_ago == 0 and idx >= 8:
                field.append(rookie)
            if years_ago == 0 and idx == 13:
                field = field[:1]                                   # a one-entry event: n < 2, no update
            quali, race = weekend(field)
            for kind, results, fn in (('quali', quali, e.update_quali_ratings), ('race', race, e.update_race_ratings)):
                e.set_recency_weight(years_ago, idx, total)
                fn(results)
                events.append(dict(kind=kind, years_ago=years_ago, race_index=idx, total_races=total, k=e.k,
                                   results=[[d, v] for d, v in results]))
                snapshot()
    out = dict(drivers=drivers + [rookie], initial=e.initial, base_k=e.base_k, events=events, after=after)
    with open(os.path.join(HERE, 'elo_season.json'), 'w') as f:
        json.dump(json.loads(json.dumps(out, default=float)), f, indent=0)
    print(f'elo_season: {len(events)} events', flush=True)


def main():
    what = sys.argv[1:] or ['cases', 'streams', 'grids', 'misc', 'weekend', 'fuzz', 'elo']
    cases = build_cases()
    if 'cases' in what:
        meta = dict(
            generated_with=dict(python=sys.version.split()[0], numpy=np.__version__, hashseed=0),
            set_pop=dict(SOFT_HARD=({'SOFT', 'MEDIUM', 'HARD'} - {'MEDIUM'}).pop(),
                         MEDIUM_HARD=({'SOFT', 'MEDIUM', 'HARD'} - {'SOFT'}).pop(),
                         SOFT_MEDIUM=({'SOFT', 'MEDIUM', 'HARD'} - {'HARD'}).pop()),
            compound_id=COMPOUND_ID,
            cases={c['name']: c for c in cases},
        )
        with open(os.path.join(HERE, 'cases.json'), 'w') as f:
            json.dump(meta, f, indent=0)
        for c in cases:
            run_case(c)
    if 'streams' in what:
        mt_streams()
    if 'grids' in what:
        sample_grids(cases)
    if 'misc' in what:
        misc()
    if 'weekend' in what:
        weekend()
    if 'fuzz' in what:
        fuzz()
    if 'elo' in what:
        elo_season()
    if 'stat' in what:
        by = {c['name']: c for c in cases}
        # 10^6 reference simulations each (round 4; rounds 1-3: 2x10^5 / 10^5): ~1.6 / 2.0 core-hours of the reference
        ref_stat(by['S60'], list(range(101, 117)), 62500)
        ref_stat(by['S78'], list(range(201, 217)), 62500)


if __name__ == '__main__':
    main()
