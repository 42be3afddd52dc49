"""Offline race-weekend orchestrator: the caller side of the hot path.

Builds every input of RaceSimulator.run_monte_carlo the way the reference's
F1Predictor.predict_weekend does (reference src/predictor.py:186-319), but from a RACE FIXTURE
(a plain dict / JSON file) instead of FastF1 sessions, so it runs without network
("next" rows 1 and 3 of SURVEY.md 8f).  Differences from the reference, on purpose:
`n_simulations` and `seed` are real arguments (the reference hard-codes 10000 and passes no seed,
:283-291; its CLI flag is ignored, main.py:14-15,27-31).

Race fixture keys
    drivers            list of driver codes (order = practice-data order in the reference, :186)
    quali_ratings      {driver: Elo quali rating}            (what the Elo history would have produced)
    quali_features     {driver: {teammate_delta, form_score, circuit_affinity}}   optional
    race_features      {driver: {clutch_factor, dnf_probability, team_trend, wet_performance}}  optional
    practice           {base_pace: {}, tire_deg: {}, tire_compounds: {}}   (outputs of :409-569)
    weather            {rainfall: bool, ...}
"""
from __future__ import annotations

import json

import numpy as np

from . import config as K
from .elo import F1EloSystem
from .simulation import RaceConfig, RaceSimulator

PENALTY_TYPES = {'engine': 10, 'full_pu': 20, 'gearbox': 5, 'pitlane_start': 20}   # reference src/config.py:81-86
UNCERTAINTY = {'fp1': 1.5, 'fp2': 1.2, 'fp3': 1.0, 'quali': 0.9, 'sprint': 0.85}   # reference :241-247
CONFIDENCE = {'fp1': 'low', 'fp2': 'moderate', 'fp3': 'good', 'quali': 'high', 'sprint': 'high'}   # :294-300


def circuit_info(race: str) -> dict:
    """CIRCUITS lookup by exact key, else by substring of the event name, else defaults (:20-43)."""
    if race in K.CIRCUITS:
        return K.CIRCUITS[race]
    low = race.lower()
    for name, info in K.CIRCUITS.items():
        if name.lower() in low:
            return info
    return {'laps': 58, 'pit_loss': 22.0, 'drs_zones': 2, 'overtake_delta': 0.8}


def create_race_config(info: dict, tire_compounds: dict | None = None) -> RaceConfig:
    """RaceConfig with the reference's hard-coded SC / VSC / red-flag / DRS constants (:45-67)."""
    return RaceConfig(
        total_laps=info.get('laps', 58), pit_loss=info.get('pit_loss', 22.0),
        overtake_delta=info.get('overtake_delta', 0.8), sc_probability=K.SC_PROBABILITY,
        vsc_probability=K.VSC_PROBABILITY, red_flag_probability=K.RED_FLAG_PROBABILITY,
        dnf_rates=K.DEFAULT_DNF_RATES, drs_zones=info.get('drs_zones', 2), drs_delta=K.DRS_DELTA,
        tire_compounds=tire_compounds or K.TIRE_COMPOUNDS, driver_teams=K.DRIVER_TEAMS)


def _penalty_value(p):
    return PENALTY_TYPES.get(p, 0) if isinstance(p, str) else p


def apply_grid_penalties(quali_positions: dict, penalties: dict) -> dict:
    """Grid after penalties: sort by (position + penalty, original position) (:69-97)."""
    ranked = sorted(quali_positions.items(), key=lambda kv: kv[1])
    shifted = sorted(((pos + _penalty_value(penalties.get(d, 0)), pos, d) for d, pos in ranked))
    return {d: i + 1 for i, (_, _, d) in enumerate(shifted)}


def predict_quali(elo: F1EloSystem, drivers, features: dict) -> dict:
    """Grid-slot distribution per driver: softmax pole probability, teammate / form / circuit
    adjustments, then a Gaussian bump around (1 - p) n with sigma max(1, n/4) (:321-375; Q22)."""
    if not drivers:
        return {}
    pole = elo.predict_quali_probs(drivers)
    for d in drivers:
        delta = features.get(d, {}).get('teammate_delta', 0)
        if delta != 0 and d in pole:
            pole[d] = pole[d] * max(0.5, min(1.5, 1 + (delta * 0.25)))
    total = sum(pole.values())
    if total > 0:
        pole = {d: p / total for d, p in pole.items()}
    n = len(drivers)
    sigma = max(1.0, n / 4)
    slots = np.arange(n)
    out = {}
    for d in drivers:
        f = features.get(d, {})
        p = pole.get(d, 1 / n) * (1 + f.get('form_score', 0) * 0.15 + f.get('circuit_affinity', 0) * 0.10)
        p = max(0.001, min(0.999, p))
        expected = (1 - p) * n
        bump = [np.exp(-((pos - expected) ** 2) / (2 * sigma ** 2)) for pos in slots.tolist()]
        s = sum(bump)
        out[d] = [b / s for b in bump] if s > 0 else [1.0 / n] * n
    return out


def adjust_for_penalties(quali_probs: dict, penalties: dict) -> dict:
    """Shift each penalised driver's distribution towards the back; mass past the last slot piles
    up there, a penalty >= n puts all mass on the last slot (:377-407)."""
    out = {}
    for d, probs in quali_probs.items():
        pen = _penalty_value(penalties.get(d, 0))
        n = len(probs)
        if pen > 0 and n > 0:
            if pen >= n:
                out[d] = [0.0] * (n - 1) + [1.0]
            else:
                shifted = [0.0] * n
                for i, p in enumerate(probs):
                    shifted[min(i + pen, n - 1)] += p
                out[d] = shifted
        else:
            out[d] = probs
    return out


def actual_grid_probs(drivers, actual_grid: dict) -> dict:
    """One-hot grid distributions from a known grid; unknown / out-of-range drivers on the last slot (:189-205)."""
    n = len(drivers)
    out = {}
    for d in drivers:
        probs = [0.0] * n
        pos = actual_grid[d] - 1 if d in actual_grid else -1
        probs[pos if 0 <= pos < n else -1] = 1.0
        out[d] = probs
    return out


class F1Predictor:
    """predict_weekend over a race fixture; the Monte Carlo step runs on the GPU."""

    def __init__(self, device: int = 0, device_front_end: bool = False):
        """device_front_end: build the grid-probability matrix on the GPU next to the race kernel
        instead of on the host; the matrices agree to ~1e-15 (the device uses its own exp, csrc/frontend_exp.h)."""
        self.elo_system = F1EloSystem()
        self.device = device
        self.device_front_end = device_front_end

    def simulator_inputs(self, fixture: dict, race: str, grid_penalties=None, circuit=None,
                         prediction_point: str = 'fp2', actual_grid=None):
        """Everything predict_weekend computes before the run_monte_carlo call (:186-281)."""
        grid_penalties = grid_penalties or {}
        circuit = circuit or circuit_info(race)
        drivers = list(fixture['drivers'])
        for d, r in fixture.get('quali_ratings', {}).items():
            self.elo_system.ratings.setdefault(d, {'quali': self.elo_system.initial, 'race': self.elo_system.initial})
            self.elo_system.ratings[d]['quali'] = r
        if actual_grid and prediction_point in ('quali', 'sprint'):
            quali_probs = actual_grid_probs(drivers, actual_grid)
        else:
            quali_probs = predict_quali(self.elo_system, drivers, fixture.get('quali_features', {}))
        if grid_penalties:
            quali_probs = adjust_for_penalties(quali_probs, grid_penalties)

        practice = fixture.get('practice', {})
        base_pace = dict(practice.get('base_pace', {}))
        tire_deg = dict(practice.get('tire_deg', {}))
        feats = fixture.get('race_features', {})
        mult = UNCERTAINTY.get(prediction_point, 1.0)
        variance = {}
        for d in drivers:                                            # :235-252
            v = max(0.05, min(0.25, 0.15 * (1 - feats.get(d, {}).get('clutch_factor', 0) * 0.2)))
            variance[d] = min(0.3, v * mult)
        config = create_race_config(circuit, practice.get('tire_compounds'))
        dnf = {d: feats.get(d, {}).get('dnf_probability', 0.05) / config.total_laps for d in drivers}   # :258-261
        weather = fixture.get('weather', {})
        track = 'damp' if weather.get('rainfall', False) else 'dry'                                    # :268
        for d in drivers:                                            # :271-274
            base_pace[d] = base_pace.get(d, 90.0) - (feats.get(d, {}).get('team_trend', 0) * 0.6)
        if track in ('damp', 'wet'):                                 # :277-281
            for d in drivers:
                base_pace[d] = base_pace[d] - (feats.get(d, {}).get('wet_performance', 0) * 0.5)
        return dict(config=config, drivers=drivers, grid_probs=quali_probs, base_pace=base_pace, tire_deg=tire_deg,
                    driver_variance=variance, driver_dnf_rates=dnf, track_condition=track, weather=weather)

    def predict_weekend(self, season: int, race: str, fixture: dict | str, grid_penalties=None, circuit_info=None,
                        prediction_point: str = 'fp2', actual_grid=None, n_simulations: int = 10000,
                        seed: int | None = None) -> dict:
        """Pole / win / podium probabilities for one weekend (:99-319), Monte Carlo on the GPU."""
        if isinstance(fixture, str):
            with open(fixture) as f:
                fixture = json.load(f)
        if not fixture.get('drivers'):
            raise ValueError(f"No practice data available for {season} {race}")       # :183-184
        inp = self.simulator_inputs(fixture, race, grid_penalties, circuit_info, prediction_point, actual_grid)
        sim = RaceSimulator(inp['config'], device=self.device)
        if self.device_front_end and not (actual_grid and prediction_point in ('quali', 'sprint')):
            # same inputs, the matrix built on the device from the ratings (no host matrix crosses PCIe)
            ratings = {d: self.elo_system.ratings.get(d, {}).get('quali', self.elo_system.initial) for d in inp['drivers']}
            race_probs, grid = sim.run_from_ratings(
                n_simulations, inp['drivers'], ratings, fixture.get('quali_features', {}), grid_penalties or {},
                inp['base_pace'], inp['tire_deg'], inp['driver_variance'], inp['driver_dnf_rates'], seed=seed,
                track_condition=inp['track_condition'])
            return pack_result(inp['drivers'], grid, race_probs, inp['weather'], prediction_point, actual_grid)
        race_probs = sim.run_monte_carlo(
            n_simulations=n_simulations, grid_probs=inp['grid_probs'], base_pace=inp['base_pace'],
            tire_deg=inp['tire_deg'], driver_variance=inp['driver_variance'],
            driver_dnf_rates=inp['driver_dnf_rates'], seed=seed, track_condition=inp['track_condition'])
        return pack_result(inp['drivers'], inp['grid_probs'], race_probs, inp['weather'], prediction_point, actual_grid)


def pack_result(drivers, quali_probs, race_probs, weather, prediction_point, actual_grid) -> dict:
    """The result dict of predict_weekend (:302-319)."""
    n = max(1, len(drivers))
    return {
        'pole_probabilities': {d: quali_probs[d][0] if quali_probs.get(d) else 1.0 / n for d in drivers},
        'win_probabilities': {d: race_probs.get(d, {}).get(1, 0) for d in drivers},
        'podium_probabilities': {d: sum(race_probs.get(d, {}).get(p, 0) for p in (1, 2, 3)) for d in drivers},
        'full_distributions': race_probs,
        'weather': weather,
        'prediction_point': prediction_point,
        'confidence': CONFIDENCE.get(prediction_point, 'moderate'),
        'grid_is_actual': actual_grid is not None and prediction_point in ('quali', 'sprint'),
    }
