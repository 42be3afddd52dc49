"""Pairwise Elo ratings for qualifying / race results and the softmax pole model.

Host-side helper on the caller side of the hot path ("next" row 4 / 3 of SURVEY.md 8f).
Behaviour follows reference src/elo.py: recency-weighted K (:13-38), expected score with the
exponent clamped to [-10, 10] (:40-43), all-pairs update computed from the ratings BEFORE the
event and scaled by 1/(n-1) (:45-122), softmax with scale 100 for pole probabilities (:124-141).
The O(n^2) loops of the reference are numpy outer products here; accumulation runs in the
reference's opponent order so the ratings agree to the last bit.
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
