#!/usr/bin/env python3
"""Timing of the Elo season row (f4) on the GPU box: mcgp_elo_season (one launch per season) against the host numpy
path (monte_carlo_gp_amd/elo.py, the reference's per-event methods) and the C oracle.   -> stdout (profiles/r3_elo_season.txt)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_elo_season as T
from monte_carlo_gp_amd.elo import F1EloSystem

with open(os.path.join(ROOT, 'tests', 'golden', 'elo_season.json')) as f:
    fx = json.load(f)
evs = T._events(fx)
drivers = fx['drivers']
n = len(drivers)


def best(fn, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts)


def host():
    e = F1EloSystem()
    for ev in fx['events']:
        e.set_recency_weight(ev['years_ago'], ev['race_index'], ev['total_races'])
        (e.update_quali_ratings if ev['kind'] == 'quali' else e.update_race_ratings)([(d, v) for d, v in ev['results']])
    return e


def device():
    e = F1EloSystem()
    e.update_season(evs)
    return e


arrs = F1EloSystem.season_arrays(evs, drivers, fx['base_k'])


def device_raw():
    import ctypes as C
    from monte_carlo_gp_amd import _native as N
    kind, k, count, who, value = arrs
    r = np.full((2, n), fx['initial'])
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    N.check(N.lib().mcgp_elo_season(n, len(kind), kind.ctypes.data_as(C.POINTER(C.c_int32)), dp(k),
                                    count.ctypes.data_as(C.POINTER(C.c_uint32)), who.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    dp(value), dp(r), None, 0))
    return r


device()                                                      # context creation, library load
t_dev = best(device, 20)
t_raw = best(device_raw, 50)
t_host = best(host, 5)
t_orc = best(lambda: T._oracle_season(n, *arrs, np.full((2, n), fx['initial']), want_after=False), 20)
pairs = sum(len(ev['results']) * (len(ev['results']) - 1) for ev in fx['events'])
print(f'fixture season: {len(evs)} events, {n} drivers, {pairs} pairwise terms')
print(f'  mcgp_elo_season, arrays ready (2 uploads + 1 launch of one 1024-thread block + 1 download): {t_raw * 1e3:.3f} ms')
print(f'  the same through F1EloSystem.update_season (packing the events into arrays in Python): {t_dev * 1e3:.3f} ms')
print(f'  host numpy path, event by event (bit-identical to the reference):                     {t_host * 1e3:.3f} ms')
print(f'  C oracle, one core:                                                                   {t_orc * 1e3:.3f} ms')
assert device().ratings.keys() == host().ratings.keys()
worst = max(abs(device().ratings[d][k] - host().ratings[d][k]) for d in host().ratings for k in ('quali', 'race'))
print(f'  largest |device - host| rating after the season: {worst:.3g}')
