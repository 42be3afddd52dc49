"""Elo updates of a season ("next" row f4): reference F1EloSystem.update_quali_ratings / update_race_ratings
(src/elo.py:45-122) with the K of set_recency_weight (:13-38), event after event.

Fixture: tests/golden/elo_season.json, made by running the REFERENCE through three seasons (tests/golden/make_goldens.py
elo_season: rookies, absentees, lap-time ties, shared positions, a one-entry event) -- ratings of every driver after
every one of the 72 events.
CPU: the host path (monte_carlo_gp_amd/elo.py, numpy) is bit-identical to it; the oracle's restatement
(oracle/mcgp_oracle.c: orc_elo_season, with the library's own 10^x from elo_update.h) is required to agree to 2 ulp
of a rating (5e-13) -- the price of not depending on a libm; on this fixture all 3024 snapshot values come out equal.
GPU: mcgp_elo_season is bit-identical to the oracle, snapshot by snapshot."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle_py as O
from monte_carlo_gp_amd.elo import F1EloSystem

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 5e-13          # 2 ulp of a rating (~1500); measured on the fixture: 0 -- every snapshot equals the reference bit for bit


@pytest.fixture(scope='module')
def season():
    with open(os.path.join(HERE, 'golden', 'elo_season.json')) as f:
        return json.load(f)


def _oracle_season(n, kind, k, count, who, value, ratings, want_after=True):
    L = O.lib()
    L.orc_elo_season.restype = C.c_int
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    E = len(kind)
    after = np.zeros((E, 2, n)) if want_after else None
    r = np.array(ratings, np.float64)
    rc = L.orc_elo_season(n, E, kind.ctypes.data_as(C.POINTER(C.c_int32)), dp(k), count.ctypes.data_as(C.POINTER(C.c_uint32)),
                          who.ctypes.data_as(C.POINTER(C.c_uint8)), dp(value), dp(r), dp(after) if want_after else None)
    assert rc == 0
    return r, after


def _events(fx, with_k=True):
    evs = []
    for ev in fx['events']:
        e = dict(kind=ev['kind'], results=[(d, v) for d, v in ev['results']])
        if with_k:
            e['k'] = ev['k']
        else:
            e.update(years_ago=ev['years_ago'], race_index=ev['race_index'], total_races=ev['total_races'])
        evs.append(e)
    return evs


def test_host_path_is_bit_identical_to_the_reference(season):
    e = F1EloSystem()
    for ev, want in zip(season['events'], season['after']):
        e.set_recency_weight(ev['years_ago'], ev['race_index'], ev['total_races'])
        assert e.k == ev['k']
        res = [(d, v) for d, v in ev['results']]
        (e.update_quali_ratings if ev['kind'] == 'quali' else e.update_race_ratings)(res)
        assert e.ratings == want


def test_pow10_against_libm():
    L = O.lib()
    L.orc_elo_pow10.restype = C.c_double
    L.orc_elo_pow10.argtypes = [C.c_double]
    xs = np.concatenate([np.linspace(-10, 10, 200001), [0.0, 1.0, -1.0, 2.0, 3.0, 10.0, -10.0, 1e-17, -1e-300]])
    worst = max(abs(L.orc_elo_pow10(float(x)) / 10.0 ** float(x) - 1) for x in xs)
    assert worst < 4.5e-16, worst                                       # 2 ulp of libm's pow
    for x in (0.0, 1.0, -1.0, 2.0, 3.0, 10.0, -10.0):
        assert L.orc_elo_pow10(x) == 10.0 ** x


def test_oracle_season_against_the_reference(season):
    drivers = season['drivers']
    n = len(drivers)
    # K from the recency arguments (the mirror's own set_recency_weight) equals the reference's K per event
    arrs_k = F1EloSystem.season_arrays(_events(season, with_k=False), drivers, season['base_k'])
    arrs = F1EloSystem.season_arrays(_events(season), drivers, season['base_k'])
    assert all(np.array_equal(a, b) for a, b in zip(arrs, arrs_k))
    kind, k, count, who, value = arrs
    final, after = _oracle_season(n, kind, k, count, who, value, np.full((2, n), season['initial']))
    worst = 0.0
    for e, want in enumerate(season['after']):
        for i, d in enumerate(drivers):
            for row, key in enumerate(('quali', 'race')):
                ref = want[d][key] if d in want else season['initial']      # not seen yet: still at the initial rating
                worst = max(worst, abs(after[e, row, i] - ref))
    assert worst <= TOL, worst
    assert np.array_equal(final, after[-1])
    # a one-entry event and an empty one change nothing
    lone = [e for e, c in enumerate(count) if c < 2]
    assert lone
    for e in lone:
        assert np.array_equal(after[e], after[e - 1])


def test_oracle_rejects_malformed_events():
    L = O.lib()
    L.orc_elo_season.restype = C.c_int
    r = np.full((2, 3), 1500.0)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    kind = np.array([2], np.int32)
    k = np.array([32.0])
    count = np.array([2], np.uint32)
    who = np.array([[0, 1, 0]], np.uint8)
    value = np.array([[1.0, 2.0, 0.0]])
    args = lambda: (3, 1, kind.ctypes.data_as(C.POINTER(C.c_int32)), dp(k), count.ctypes.data_as(C.POINTER(C.c_uint32)),
                    who.ctypes.data_as(C.POINTER(C.c_uint8)), dp(value), dp(r), None)
    assert L.orc_elo_season(*args()) == -1
    kind[0] = 1
    count[0] = 4
    assert L.orc_elo_season(*args()) == -1
    count[0] = 2
    assert L.orc_elo_season(*args()) == 0


def test_the_two_copies_of_the_header_are_identical():
    a = open(os.path.join(HERE, '..', 'oracle', 'elo_update.h')).read()
    b = open(os.path.join(HERE, '..', 'monte_carlo_gp_amd', 'csrc', 'elo_update.h')).read()
    assert a == b


# ---------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_device_season_is_bit_identical_to_the_oracle(require_gpu, season):
    drivers = season['drivers']
    n = len(drivers)
    evs = _events(season)
    kind, k, count, who, value = F1EloSystem.season_arrays(evs, drivers, season['base_k'])
    _, after = _oracle_season(n, kind, k, count, who, value, np.full((2, n), season['initial']))
    e = F1EloSystem()
    snaps = e.update_season(evs, snapshots=True)
    assert len(snaps) == len(evs)
    order = list(e.ratings)                                              # drivers in order of first appearance
    assert sorted(order) == sorted(d for d in drivers if d in season['after'][-1])
    for ev_i, (snap, want) in enumerate(zip(snaps, season['after'])):
        assert set(snap) == set(want), ev_i                              # registered exactly when the reference does
        for d in snap:
            i = drivers.index(d)
            assert snap[d]['quali'] == after[ev_i, 0, i] and snap[d]['race'] == after[ev_i, 1, i]   # == oracle, bit for bit
            assert abs(snap[d]['quali'] - want[d]['quali']) <= TOL and abs(snap[d]['race'] - want[d]['race']) <= TOL
    assert e.ratings == snaps[-1]
    # the same season in two calls (ratings carried by the object) ends in the same place
    e2 = F1EloSystem()
    e2.update_season(evs[:31])
    e2.update_season(evs[31:])
    assert e2.ratings == e.ratings


@pytest.mark.gpu
def test_device_season_of_the_largest_field_and_many_events(require_gpu):
    """32 drivers (the ABI maximum), 600 events of random size, ties and shared positions: device == oracle exactly."""
    import ctypes as C
    from monte_carlo_gp_amd import _native as N
    rs = np.random.RandomState(5)
    n, E = 32, 600
    kind = rs.randint(0, 2, E).astype(np.int32)
    k = rs.choice([16.0, 22.4, 24.0, 32.0, 48.0], E)
    count = rs.randint(0, n + 1, E).astype(np.uint32)
    who = np.zeros((E, n), np.uint8)
    value = np.zeros((E, n))
    for e in range(E):
        who[e, :count[e]] = rs.permutation(n)[:count[e]]
        value[e, :count[e]] = np.round(rs.uniform(70, 100, count[e]), 1) if kind[e] == 0 else rs.randint(1, 21, count[e])
    start = np.vstack([1500 + 40 * rs.randn(n), np.full(n, 1500.0)])
    want, want_after = _oracle_season(n, kind, k, count, who, value, start)
    got = start.copy()
    got_after = np.zeros((E, 2, n))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    N.check(N.lib().mcgp_elo_season(n, E, kind.ctypes.data_as(C.POINTER(C.c_int32)), dp(k),
                                    count.ctypes.data_as(C.POINTER(C.c_uint32)), who.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    dp(value), dp(got), dp(got_after), 0))
    assert np.array_equal(got, want) and np.array_equal(got_after, want_after)
    # extreme rating gaps: the exponent clamp of expected_score (:42)
    far = np.array([[4000.0, 100.0, 1500.0], [1500.0] * 3])
    one = (np.zeros(1, np.int32), np.array([32.0]), np.array([3], np.uint32), np.array([[0, 1, 2]], np.uint8),
           np.array([[2.0, 1.0, 3.0]]))
    w, _ = _oracle_season(3, *one, far)
    g = far.copy()
    N.check(N.lib().mcgp_elo_season(3, 1, one[0].ctypes.data_as(C.POINTER(C.c_int32)), dp(one[1]),
                                    one[2].ctypes.data_as(C.POINTER(C.c_uint32)), one[3].ctypes.data_as(C.POINTER(C.c_uint8)),
                                    dp(one[4]), dp(g), None, 0))
    assert np.array_equal(g, w)
    ref = F1EloSystem()
    ref.ratings = {'a': {'quali': 4000.0, 'race': 1500.0}, 'b': {'quali': 100.0, 'race': 1500.0}, 'c': {'quali': 1500.0, 'race': 1500.0}}
    ref.update_quali_ratings([('a', 2.0), ('b', 1.0), ('c', 3.0)])
    assert max(abs(g[0, i] - ref.ratings[d]['quali']) for i, d in enumerate('abc')) <= TOL


@pytest.mark.gpu
def test_device_season_rejects_malformed_input(require_gpu):
    import ctypes as C
    from monte_carlo_gp_amd import _native as N
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))

    def call(n, kind, k, count, who, value, ratings):
        return N.lib().mcgp_elo_season(n, len(kind), kind.ctypes.data_as(C.POINTER(C.c_int32)), dp(k),
                                       count.ctypes.data_as(C.POINTER(C.c_uint32)), who.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       dp(value), dp(ratings), None, 0)
    good = dict(kind=np.array([0], np.int32), k=np.array([32.0]), count=np.array([2], np.uint32),
                who=np.array([[0, 1, 0]], np.uint8), value=np.array([[80.0, 81.0, 0.0]]), ratings=np.full((2, 3), 1500.0))
    assert call(3, **good) == 0
    for key, bad in (('kind', np.array([2], np.int32)), ('k', np.array([float('nan')])), ('count', np.array([4], np.uint32)),
                     ('who', np.array([[0, 3, 0]], np.uint8)), ('who', np.array([[1, 1, 0]], np.uint8)),
                     ('value', np.array([[80.0, float('inf'), 0.0]])), ('ratings', np.array([[1500.0, float('nan'), 1500.0]] * 2))):
        args = dict(good, ratings=np.full((2, 3), 1500.0))
        args[key] = bad
        assert call(3, **args) == -1, key
        assert N.lib().mcgp_last_error()
    assert call(33, **good) == -1 and call(0, **good) == -1
