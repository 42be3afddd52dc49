"""Scoring of probabilistic predictions: Brier score, podium accuracy and the calibration curve.

Formulas of reference src/validation.py:82-158 ("next" row 2 of SURVEY.md 8f).  The live-data
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


def calibration_curve(y_true, y_prob, n_bins: int = 5):
    """sklearn.calibration.calibration_curve(y_true, y_prob, n_bins=n_bins) with its defaults (uniform bins, no
    normalisation), which is what reference src/validation.py:153 calls; numpy only (the GPU box has no sklearn
    dependency to lean on).  Bin edges linspace(0, 1, n_bins + 1); a probability joins the bin whose upper edge is the
    first one >= it (searchsorted on the inner edges); per non-empty bin the mean outcome and the mean probability,
    accumulated by np.bincount in input order like sklearn does.  ValueError for probabilities outside [0, 1] or
    outcomes other than 0 / 1."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_prob = np.asarray(y_prob, dtype=np.float64)
    if y_true.shape != y_prob.shape:
        raise ValueError('y_true and y_prob must have the same length')
    if y_prob.size and (y_prob.min() < 0 or y_prob.max() > 1):
        raise ValueError('y_prob has values outside [0, 1].')
    if len(np.setdiff1d(np.unique(y_true), [0.0, 1.0])):
        raise ValueError('Only binary classification is supported.')
    bins = np.linspace(0.0, 1.0, n_bins + 1)
    binids = np.searchsorted(bins[1:-1], y_prob)
    bin_sums = np.bincount(binids, weights=y_prob, minlength=len(bins))
    bin_true = np.bincount(binids, weights=y_true, minlength=len(bins))
    bin_total = np.bincount(binids, minlength=len(bins))
    nonzero = bin_total != 0
    return bin_true[nonzero] / bin_total[nonzero], bin_sums[nonzero] / bin_total[nonzero]


def calibration_analysis(predictions, actuals) -> dict:
    """Calibration of the win probabilities over a sweep (reference :133-158): every (driver, race) pair is a sample
    (probability, won or not); min(10, max(2, samples // 10)) bins; races without a winner or without win
    probabilities are skipped."""
    all_probs, all_outcomes = [], []
    for pred, act in zip(predictions, actuals):
        if not act.get('winner'):
            continue
        win_probs = pred.get('win_probabilities', {})
        if not win_probs:
            continue
        for driver, prob in win_probs.items():
            all_probs.append(prob)
            all_outcomes.append(1 if driver == act['winner'] else 0)
    if not all_probs:
        return {'prob_true': [], 'prob_pred': []}
    n_bins = min(10, max(2, len(all_probs) // 10))
    try:
        prob_true, prob_pred = calibration_curve(all_outcomes, all_probs, n_bins=n_bins)
        return {'prob_true': prob_true.tolist(), 'prob_pred': prob_pred.tolist()}
    except ValueError:
        return {'prob_true': [], 'prob_pred': []}
