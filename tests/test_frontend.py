"""Grid-probability front end: reference src/elo.py:124-141 + src/predictor.py:321-407.

CPU: the oracle's restatement (oracle/mcgp_oracle.c: orc_grid_probs, with the front end's own exp) against the
matrices the REFERENCE produced (tests/golden/misc.json, made by tests/golden/make_goldens.py) -- tolerance
1e-14 relative, the price of not depending on a libm -- and against the host predictor (numpy, bit-identical to
the reference).  GPU: the HIP front end is bit-identical to the oracle's, and a run whose matrix was built on
the device equals the oracle run on that matrix."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle_py as O
from monte_carlo_gp_amd.elo import F1EloSystem
from monte_carlo_gp_amd.predictor import adjust_for_penalties, predict_quali
from monte_carlo_gp_amd.simulation import RaceSimulator

HERE = os.path.dirname(os.path.abspath(__file__))
REL_TOL = 1e-14


def _misc():
    with open(os.path.join(O.GOLDEN_DIR, 'misc.json')) as f:
        return json.load(f)


def _oracle_matrix(drivers, ratings, features, penalties):
    L = O.lib()
    L.orc_grid_probs.restype = C.c_int
    r, td, fs, ca, pen = RaceSimulator.front_end_arrays(drivers, ratings, features, penalties)
    n = len(drivers)
    out = np.zeros((n, n))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert L.orc_grid_probs(dp(r), dp(td), dp(fs), dp(ca), pen.ctypes.data_as(C.POINTER(C.c_int32)), n, dp(out)) == 0
    return out


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def test_exp_against_libm():
    L = O.lib()
    L.orc_fe_exp.restype = C.c_double
    L.orc_fe_exp.argtypes = [C.c_double]
    xs = np.concatenate([np.linspace(-40, 0, 100001), np.linspace(-700, 700, 20001), [-745.5, -1e-300, 0.0, 1e-17, 709.5]])
    worst = max(abs(L.orc_fe_exp(float(x)) / math.exp(x) - 1) for x in xs if 1e-300 < math.exp(min(x, 700)) and x < 709)
    assert worst < 4.5e-16, worst
    assert L.orc_fe_exp(0.0) == 1.0 and L.orc_fe_exp(-1e9) == 0.0 and math.isinf(L.orc_fe_exp(1e9))
    assert math.isnan(L.orc_fe_exp(float('nan')))


def test_two_copies_of_the_header_are_identical():
    a = open(os.path.join(HERE, '..', 'oracle', 'frontend_exp.h')).read()
    b = open(os.path.join(HERE, '..', 'monte_carlo_gp_amd', 'csrc', 'frontend_exp.h')).read()
    assert a == b


@pytest.mark.parametrize('key,with_features,with_penalties', [('quali_plain', False, False), ('quali_feat', True, False),
                                                               ('quali_penalised', None, True)])
def test_oracle_front_end_matches_reference_fixture(key, with_features, with_penalties):
    m = _misc()
    drivers = m['drivers']
    ref = np.array([m[key][d] for d in drivers])
    candidates = [m['features'], {}] if with_features is None else [m['features'] if with_features else {}]
    best = min(_rel(_oracle_matrix(drivers, m['ratings'], f, m['penalties'] if with_penalties else {}), ref)
               for f in candidates)
    assert best < REL_TOL, best
    assert np.allclose(ref.sum(axis=1), 1.0, atol=1e-12)


def test_oracle_front_end_matches_host_predictor_on_random_inputs():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 7, 20, 24, 32):
        drivers = [f'D{i:02d}' for i in range(n)]
        ratings = {d: float(1500 + 150 * rng.standard_normal()) for d in drivers}
        feats = {d: dict(teammate_delta=float(rng.choice([0.0, 0.3, -0.4, 3.0, -3.0])), form_score=float(rng.uniform(-1, 1)),
                         circuit_affinity=float(rng.uniform(-1, 1))) for d in drivers[::2]}
        pens = {drivers[0]: 'engine', drivers[-1]: 3, drivers[n // 2]: 40}
        elo = F1EloSystem()
        for d in drivers:
            elo.ratings[d] = {'quali': ratings[d], 'race': 1500.0}
        host = adjust_for_penalties(predict_quali(elo, drivers, feats), pens)
        got = _oracle_matrix(drivers, ratings, feats, pens)
        assert _rel(got, np.array([host[d] for d in drivers])) < REL_TOL, n


@pytest.mark.gpu
def test_device_front_end_is_bit_identical_to_the_oracle(require_gpu):
    m = _misc()
    drivers = m['drivers']
    sim = RaceSimulator(__import__('monte_carlo_gp_amd').RaceConfig(**O.load_case('S60')['config']))
    for feats, pens in (({}, {}), (m['features'], {}), (m['features'], m['penalties'])):
        got = sim.grid_probs_on_device(drivers, m['ratings'], feats, pens)
        ref = _oracle_matrix(drivers, m['ratings'], feats, pens)
        assert np.array_equal(np.array([got[d] for d in drivers]), ref)
    rng = np.random.default_rng(11)
    for n in (1, 2, 5, 21, 32):
        drv = [f'D{i:02d}' for i in range(n)]
        ratings = {d: float(1500 + 200 * rng.standard_normal()) for d in drv}
        feats = {d: dict(teammate_delta=float(rng.uniform(-3, 3)), form_score=float(rng.uniform(-1, 1))) for d in drv}
        pens = {drv[0]: 2, drv[-1]: 'pitlane_start'}
        got = sim.grid_probs_on_device(drv, ratings, feats, pens)
        assert np.array_equal(np.array([got[d] for d in drv]), _oracle_matrix(drv, ratings, feats, pens)), n


@pytest.mark.gpu
def test_run_from_ratings_equals_oracle_on_the_device_matrix(require_gpu):
    """The matrix never leaves the GPU between the front end and the race kernel; the histogram must be the
    oracle's for that matrix, and the predictor option gives the same pole probabilities as the host path to 1e-14."""
    from monte_carlo_gp_amd import RaceConfig
    from monte_carlo_gp_amd.cli import synthetic_fixture
    from monte_carlo_gp_amd.predictor import F1Predictor
    m = _misc()
    case = O.load_case('S60')
    drivers = list(case['grid_probs'])
    sim = RaceSimulator(RaceConfig(**case['config']), set_pop=O.load_cases()['set_pop'])
    ratings = {d: m['ratings'].get(d, 1500.0) for d in drivers}
    probs, grid = sim.run_from_ratings(5000, drivers, ratings, m['features'], {'NOR': 5}, case['base_pace'],
                                       case['tire_deg'], case['driver_variance'], case['driver_dnf_rates'], seed=9)
    ref_grid = _oracle_matrix(drivers, ratings, m['features'], {'NOR': 5})
    assert np.array_equal(np.array([grid[d] for d in drivers]), ref_grid)
    ref = O.Problem(dict(case, grid_probs={d: list(ref_grid[i]) for i, d in enumerate(drivers)})).run(
        5000, rng=O.RNG_PHILOX, seed=9)
    assert np.array_equal(sim.last_histogram, ref['hist'])
    fx = synthetic_fixture()
    a = F1Predictor(device_front_end=True).predict_weekend(2024, 'Bahrain', fx, n_simulations=20000, seed=3)
    b = F1Predictor(device_front_end=False).predict_weekend(2024, 'Bahrain', fx, n_simulations=20000, seed=3)
    for d in fx['drivers']:
        assert abs(a['pole_probabilities'][d] - b['pole_probabilities'][d]) <= REL_TOL * b['pole_probabilities'][d]
    assert a['win_probabilities'] == b['win_probabilities']        # same sampled grids: the matrices differ by ~1 ulp


@pytest.mark.gpu
def test_device_front_end_rejects_non_finite_inputs(require_gpu):
    """an infinite rating (inf - inf in the softmax) or a NaN / infinite feature would put a NaN matrix into
    the parameter block, which the race kernel samples uniform grids from without a word; mcgp_run rejects such a
    matrix, and so must the front-end entry points (MCGP_E_BAD_ARG, nothing launched)."""
    from monte_carlo_gp_amd import RaceConfig, _native as N
    case = O.load_case('S60')
    drivers = list(case['grid_probs'])
    sim = RaceSimulator(RaceConfig(**case['config']), set_pop=O.load_cases()['set_pop'])
    good = {d: 1600.0 - 10 * i for i, d in enumerate(drivers)}
    bad_inputs = [
        (dict(good, **{drivers[3]: float('inf')}), {}),
        (dict(good, **{drivers[0]: float('nan')}), {}),
        (good, {drivers[5]: dict(form_score=float('nan'))}),
        (good, {drivers[7]: dict(teammate_delta=float('-inf'))}),
        (good, {drivers[9]: dict(circuit_affinity=float('inf'))}),
    ]
    for ratings, feats in bad_inputs:
        with pytest.raises(N.McgpError) as e:
            sim.grid_probs_on_device(drivers, ratings, feats, {})
        assert e.value.code == -1
        with pytest.raises(N.McgpError) as e:
            sim.run_from_ratings(1000, drivers, ratings, feats, {}, case['base_pace'], case['tire_deg'],
                                 case['driver_variance'], case['driver_dnf_rates'], seed=1)
        assert e.value.code == -1
    # and the same call with finite inputs still runs
    probs, _ = sim.run_from_ratings(1000, drivers, good, {}, {}, case['base_pace'], case['tire_deg'],
                                    case['driver_variance'], case['driver_dnf_rates'], seed=1)
    assert abs(sum(probs[drivers[0]].values()) - 1.0) < 1e-12
