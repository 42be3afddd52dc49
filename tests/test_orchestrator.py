"""Host-side helpers either side of the hot path vs fixtures produced by the reference
(tests/golden/misc.json = G6, tests/golden/weekend.json = predict_weekend's glue)."""
import json

import numpy as np
import pytest

import oracle_py as O
from monte_carlo_gp_amd import predictor as PR
from monte_carlo_gp_amd.elo import F1EloSystem
from monte_carlo_gp_amd.validation import brier_score, podium_accuracy


@pytest.fixture(scope='module')
def misc():
    with open(O.GOLDEN_DIR + '/misc.json') as f:
        return json.load(f)


@pytest.fixture(scope='module')
def weekend():
    with open(O.GOLDEN_DIR + '/weekend.json') as f:
        return json.load(f)


def _elo(ratings):
    e = F1EloSystem()
    e.ratings = {d: {'quali': r, 'race': r} for d, r in ratings.items()}
    return e


def test_predict_quali_matches_reference(misc):
    drivers = misc['drivers']
    assert _elo(misc['ratings']).predict_quali_probs(drivers) == misc['pole_probs']
    assert PR.predict_quali(_elo(misc['ratings']), drivers, {}) == misc['quali_plain']
    assert PR.predict_quali(_elo(misc['ratings']), drivers, misc['features']) == misc['quali_feat']
    # Q22: every distribution peaks near the back of the grid
    assert max(misc['quali_plain']['VER']) == misc['quali_plain']['VER'][-1] or np.argmax(misc['quali_plain']['VER']) > 10


def test_penalties_match_reference(misc):
    assert PR.adjust_for_penalties(misc['quali_plain'], misc['penalties']) == misc['quali_penalised']
    grid = PR.apply_grid_penalties({d: i + 1 for i, d in enumerate(misc['drivers'])}, misc['penalties'])
    assert grid == misc['apply_grid_penalties']


def test_race_config_and_circuit_lookup(misc):
    c = PR.create_race_config(PR.circuit_info('Bahrain Grand Prix'))
    assert [c.total_laps, c.pit_loss, c.overtake_delta, c.sc_probability, c.vsc_probability,
            c.red_flag_probability, c.drs_delta] == misc['race_config_bahrain']
    f = PR.create_race_config(PR.circuit_info('Nowhere GP'))
    assert [f.total_laps, f.pit_loss, f.overtake_delta, f.drs_zones] == misc['race_config_fallback']


def test_brier_and_podium_match_reference(misc):
    uni = {d: 1.0 / 20 for d in misc['drivers']}
    assert float(brier_score([uni], ['VER'])) == misc['brier_uniform']
    assert abs(misc['brier_uniform'] - 0.0475) < 1e-12          # the reference CLI's quoted baseline
    b = misc['brier_inputs']
    assert float(brier_score(b['preds'], b['actuals'])) == misc['brier_mixed']
    pi = misc['podium_inputs']
    assert podium_accuracy(pi['preds'], pi['actuals']) == misc['podium_accuracy']
    assert brier_score([], []) == 1.0 and podium_accuracy([], []) == 0.0


def test_elo_updates_match_reference(misc):
    drivers = misc['drivers']
    e = F1EloSystem()
    e.set_recency_weight(0, 5, 24)
    e.update_quali_ratings([(d, 80.0 + 0.1 * ((i * 7) % 20)) for i, d in enumerate(drivers)])
    e.set_recency_weight(1)
    e.update_race_ratings([(d, ((i * 3) % 20) + 1) for i, d in enumerate(drivers)])
    for d in drivers:
        for k in ('quali', 'race'):
            assert e.ratings[d][k] == misc['elo_after'][d][k]          # bit-identical (same operations, same order)


@pytest.mark.parametrize('label', ['dry_fp2', 'damp_quali'])
def test_weekend_glue_hands_the_same_arguments_to_the_hot_path(weekend, label):
    w = weekend[label]
    fixture = dict(drivers=w['drivers'], quali_ratings=w['quali_ratings'], quali_features=w['quali_features'],
                   race_features=w['race_features'], practice=w['practice'], weather=w['weather'])
    inp = PR.F1Predictor().simulator_inputs(fixture, w['race'], grid_penalties=w['grid_penalties'],
                                            prediction_point=w['prediction_point'], actual_grid=w['actual_grid'])
    call, cfg = w['captured']['call'], w['captured']['config']
    assert inp['track_condition'] == call['track_condition']
    assert list(inp['grid_probs']) == list(call['grid_probs'])
    for k in ('grid_probs', 'base_pace', 'tire_deg', 'driver_variance', 'driver_dnf_rates'):
        assert inp[k] == call[k], k
    for k, v in cfg.items():
        assert getattr(inp['config'], k) == v, k
    fake = {d: {i + 1: 1.0 if i == j else 0.0 for i in range(3)} for j, d in enumerate(w['drivers'])}
    res = PR.pack_result(inp['drivers'], inp['grid_probs'], fake, inp['weather'], w['prediction_point'], w['actual_grid'])
    for k, v in w['result'].items():
        assert res[k] == v, k


def test_calibration_curve_matches_reference(misc):
    """calibration_analysis (reference src/validation.py:133-158, sklearn underneath) restated in numpy: the fixtures
    hold the reference's own outputs for a season-sized sample, two small ones and the empty one."""
    from monte_carlo_gp_amd.validation import calibration_analysis, calibration_curve
    for name, c in misc['calibration'].items():
        got = calibration_analysis(c['preds'], c['actuals'])
        assert got == c['result'], name                        # same numpy operations in the same order: exact
    with pytest.raises(ValueError):
        calibration_curve([0, 1], [0.2, 1.5], n_bins=2)
    with pytest.raises(ValueError):
        calibration_curve([0, 2], [0.2, 0.5], n_bins=2)
    assert calibration_analysis([{'win_probabilities': {'A': 2.0}}], [{'winner': 'A'}]) == {'prob_true': [], 'prob_pred': []}


def test_known_pairs_update_equals_the_reference_on_a_complete_ordering(misc):
    """cli._update_known_pairs (the sweep's Elo step, from outcome fixtures that only know pole / podium) restricted to
    NO restriction -- every pair known -- is the reference's all-pairs update (src/elo.py:45-122)."""
    from monte_carlo_gp_amd.cli import _update_known_pairs
    from monte_carlo_gp_amd.elo import F1EloSystem
    fx = misc['elo_full_order']
    drivers = misc['drivers']
    e = F1EloSystem()
    for i, d in enumerate(drivers):
        e.ratings[d] = {'quali': 1700.0 - 20.0 * i, 'race': 1650.0 - 15.0 * i}
    e.set_recency_weight(0, fx['race_index'], fx['total'])
    _update_known_pairs(e, 'quali', drivers, fx['order'])
    _update_known_pairs(e, 'race', drivers, fx['order'])
    for d in drivers:
        for k in ('quali', 'race'):
            assert abs(e.ratings[d][k] - fx['after'][d][k]) < 1e-9, (d, k)
