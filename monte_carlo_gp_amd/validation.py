"""Scoring of probabilistic predictions: Brier score and podium accuracy.

Formulas of reference src/validation.py:82-130 ("next" row 2 of SURVEY.md 8f).  The live-data
parts of the reference's validation module (FastF1 schedule / result fetch, :8-79) need the
network and are out of scope; outcomes come from a fixture instead.
"""
from __future__ import annotations

import numpy as np


def brier_score(predictions, actuals) -> float:
    """Mean over races of mean_d (p_d - [d == actual])^2; races without outcome or with
    probabilities outside [0, 1] are skipped; 1.0 when nothing can be scored (:82-106)."""
    scores = []
    for pred, actual in zip(predictions, actuals):
        if actual is None or not pred:
            continue
        if not all(0 <= p <= 1 for p in pred.values()):
            print("Warning: Invalid probabilities detected (not in [0,1])")
            continue
        s = 0.0
        for driver, prob in pred.items():
            s += (prob - (1.0 if driver == actual else 0.0)) ** 2
        scores.append(s / len(pred))
    return np.mean(scores) if scores else 1.0


def podium_accuracy(predictions, actuals) -> float:
    """Share of actual podium finishers among the three highest podium probabilities (:109-130)."""
    correct = total = 0
    for pred, act in zip(predictions, actuals):
        if not act.get('podium'):
            continue
        probs = pred.get('podium_probabilities', {})
        if not probs:
            continue
        top3 = {d for d, _ in sorted(probs.items(), key=lambda kv: kv[1], reverse=True)[:3]}
        correct += len(top3 & set(act['podium']))
        total += 3
    return correct / total if total > 0 else 0.0
