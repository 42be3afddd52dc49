"""Pairwise Elo ratings for qualifying / race results and the softmax pole model.

Host-side helper on the caller side of the hot path ("next" row 4 / 3 of SURVEY.md 8f).
Behaviour follows reference src/elo.py: recency-weighted K (:13-38), expected score with the
exponent clamped to [-10, 10] (:40-43), all-pairs update computed from the ratings BEFORE the
event and scaled by 1/(n-1) (:45-122), softmax with scale 100 for pole probabilities (:124-141).
The O(n^2) loops of the reference are numpy outer products here; accumulation runs in the
reference's opponent order so the ratings agree to the last bit.

`F1EloSystem.update_season` applies a whole sequence of events in one launch of the HIP library
(include/mcgp.h: mcgp_elo_season; the ratings stay in LDS between events): same updates, with the
library's own 10^x instead of libm's, so within a few ulp of the per-event host path instead of
bit-identical to it (and bit-identical to the CPU oracle's restatement).
"""
from __future__ import annotations

import numpy as np


class F1EloSystem:
    def __init__(self, k_factor: float = 32, initial_rating: float = 1500):
        self.base_k = k_factor
        self.k = k_factor
        self.initial = initial_rating
        self.ratings: dict = {}

    def set_recency_weight(self, years_ago: float, race_index: int = 0, total_races: int = 24):
        """K grows with recency: current season 0.75x..1.5x by race index, then 1.0 / 0.7 / 0.5 (:13-38)."""
        if years_ago <= 0:
            self.k = self.base_k * (0.75 + (0.75 * race_index / max(1, total_races - 1)))
        elif years_ago <= 1:
            self.k = self.base_k * 1.0
        elif years_ago <= 2:
            self.k = self.base_k * 0.7
        else:
            self.k = self.base_k * 0.5

    def expected_score(self, rating_a: float, rating_b: float) -> float:
        exponent = max(-10, min(10, (rating_b - rating_a) / 400))
        return 1 / (1 + 10 ** exponent)

    def _update(self, results, key, lower_is_better_values):
        n = len(results)
        if n < 2:
            return
        names = [d for d, _ in results]
        for d in names:
            self.ratings.setdefault(d, {'quali': self.initial, 'race': self.initial})
        r = np.array([self.ratings[d][key] for d in names], np.float64)
        v = np.array(lower_is_better_values, np.float64)
        expo = np.clip((r[None, :] - r[:, None]) / 400, -10, 10)           # [a, b] = (r_b - r_a) / 400
        expected = 1 / (1 + np.array([[10 ** float(e) for e in row] for row in expo]))
        actual = np.where(v[:, None] < v[None, :], 1.0, np.where(v[:, None] > v[None, :], 0.0, 0.5))
        term = self.k * (actual - expected) / (n - 1)
        for a, d in enumerate(names):
            delta = 0
            for b in range(n):                                              # reference accumulation order
                if a != b:
                    delta += float(term[a, b])
            self.ratings[d][key] += delta

    def update_quali_ratings(self, quali_results):
        """quali_results: [(driver, best_lap_time)]; faster beats slower, equal times tie (:45-83)."""
        self._update(quali_results, 'quali', [t for _, t in quali_results])

    def update_race_ratings(self, race_results):
        """race_results: [(driver, finish_position)]; lower position wins (:85-122)."""
        self._update(race_results, 'race', [p for _, p in race_results])

    @staticmethod
    def season_arrays(events, drivers, base_k=32):
        """Dense arrays of mcgp_elo_season for `events`: each a dict with 'kind' ('quali' | 'race'), 'results'
        ([(driver, lap time | finishing position)]) and either 'k' or the arguments of set_recency_weight
        ('years_ago', 'race_index', 'total_races').  Returns (kind i32[E], k f64[E], count u32[E], who u8[E, n],
        value f64[E, n])."""
        idx = {d: i for i, d in enumerate(drivers)}
        n, E = len(drivers), len(events)
        kind = np.zeros(E, np.int32)
        k = np.zeros(E, np.float64)
        count = np.zeros(E, np.uint32)
        who = np.zeros((E, n), np.uint8)
        value = np.zeros((E, n), np.float64)
        scratch = F1EloSystem(k_factor=base_k)
        for e, ev in enumerate(events):
            if ev['kind'] not in ('quali', 'race'):
                raise ValueError(f"event {e}: kind must be 'quali' or 'race'")
            kind[e] = 0 if ev['kind'] == 'quali' else 1
            if 'k' in ev:
                k[e] = ev['k']
            else:
                scratch.set_recency_weight(ev['years_ago'], ev.get('race_index', 0), ev.get('total_races', 24))
                k[e] = scratch.k
            res = ev['results']
            if len(res) > n:
                raise ValueError(f'event {e}: more entries than drivers')
            count[e] = len(res)
            for j, (d, v) in enumerate(res):
                who[e, j] = idx[d]
                value[e, j] = v
        return kind, k, count, who, value

    def update_season(self, events, device: int = 0, snapshots: bool = False):
        """Apply `events` (see season_arrays) in order on the device -- the update_quali_ratings /
        update_race_ratings calls of a season (:45-122) with the K of set_recency_weight (:13-38) per event, one
        kernel launch.  self.ratings is updated in place (drivers appear in it from their first event on, like in
        the reference); with snapshots=True also returns the ratings after every event.  Needs the HIP library
        and a GPU: there is no host fallback behind this method (the per-event methods above ARE the host path).

        Limits the per-event methods do not have (mcgp_elo_season returns MCGP_E_BAD_ARG -> McgpError): at most 32
        distinct drivers over the whole sequence (the device keeps the ratings of a 32-car field in LDS), and no
        driver listed twice in one event -- the reference accepts that (its deltas dict keeps the last entry,
        :62-83); feed such an event through update_quali_ratings / update_race_ratings instead.  The CLI's season
        fixtures use the per-event host path (cli.season_fixtures); this method is the device chain ratings -> grid
        matrix (mcgp_run_from_ratings) -> race kernel."""
        import ctypes as C
        from . import _native as N
        drivers = list(self.ratings)
        for ev in events:
            for d, _ in ev['results']:
                if d not in self.ratings and d not in drivers:
                    drivers.append(d)
        n = len(drivers)
        kind, k, count, who, value = self.season_arrays(events, drivers, self.base_k)
        ratings = np.array([[self.ratings.get(d, {}).get(key, self.initial) for d in drivers] for key in ('quali', 'race')],
                           np.float64)
        after = np.zeros((len(events), 2, n), np.float64) if snapshots else None
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        N.check(N.lib().mcgp_elo_season(n, len(events), kind.ctypes.data_as(C.POINTER(C.c_int32)), dp(k),
                                        count.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        who.ctypes.data_as(C.POINTER(C.c_uint8)), dp(value), dp(ratings),
                                        dp(after) if snapshots else None, device))
        seen = set(self.ratings)
        out = []
        for e, ev in enumerate(events):
            if len(ev['results']) >= 2:
                seen.update(d for d, _ in ev['results'])           # registered at :59-61 / :98-100, after the n < 2 return
            if snapshots:
                out.append({d: {'quali': float(after[e, 0, i]), 'race': float(after[e, 1, i])}
                            for i, d in enumerate(drivers) if d in seen})
        for i, d in enumerate(drivers):
            if d in seen:
                self.ratings[d] = {'quali': float(ratings[0, i]), 'race': float(ratings[1, i])}
        return out if snapshots else None

    def predict_quali_probs(self, drivers):
        """Softmax of quali ratings / 100 with max subtraction (:124-141)."""
        if not drivers:
            return {}
        scaled = {d: self.ratings.get(d, {}).get('quali', self.initial) / 100 for d in drivers}
        top = max(scaled.values())
        e = {d: np.exp(s - top) for d, s in scaled.items()}
        total = sum(e.values())
        n = len(drivers)
        return {d: x / total for d, x in e.items()} if total > 0 else {d: 1.0 / n for d in drivers}

    def get_rating(self, driver: str, rating_type: str = 'quali') -> float:
        return self.ratings.get(driver, {}).get(rating_type, self.initial)
