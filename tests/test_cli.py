"""CLI / backtest sweep: host logic on CPU with a stand-in predictor, end-to-end on the GPU."""
import json

import numpy as np
import pytest

from monte_carlo_gp_amd import cli
from monte_carlo_gp_amd import config as K


class _FakePredictor:
    """Deterministic stand-in: favourite = first driver; checks what the sweep hands over."""
    calls = []

    def predict_weekend(self, season, race, fixture, n_simulations=0, seed=None, **kw):
        _FakePredictor.calls.append((season, race, n_simulations, seed))
        d = fixture['drivers']
        win = {x: (0.5 if i == 0 else 0.5 / (len(d) - 1)) for i, x in enumerate(d)}
        return dict(pole_probabilities=win, win_probabilities=win, podium_probabilities=win)


def test_backtest_sweep_scoring_and_sharding():
    _FakePredictor.calls.clear()
    full = cli.backtest([2024], seed=42, n_simulations=123, predictor_factory=_FakePredictor)
    assert full['n_races'] == 24 and len(_FakePredictor.calls) == 24
    assert all(c[2] == 123 for c in _FakePredictor.calls)
    assert len({c[3] for c in _FakePredictor.calls}) == 24          # a distinct seed per race
    # VER (first driver) won 9 of 24: check the Brier arithmetic by hand
    wins = sum(r['actual']['winner'] == 'VER' for r in full['races'])
    n = len(K.DRIVER_TEAMS)
    p0, pr = 0.5, 0.5 / (n - 1)
    hit = ((p0 - 1) ** 2 + (n - 1) * pr ** 2) / n
    miss = (p0 ** 2 + (pr - 1) ** 2 + (n - 2) * pr ** 2) / n
    assert full['win_brier'] == pytest.approx((wins * hit + (24 - wins) * miss) / 24, rel=1e-12)
    # rank sharding covers every race exactly once, with the same per-race seeds
    jobs = cli.backtest_jobs([2024], 42)
    assert [j[2] for j in jobs] == [c[3] for c in _FakePredictor.calls]
    parts = [cli.shard_jobs(jobs, r, 3) for r in range(3)]
    assert sorted(i for part in parts for i, _ in part) == list(range(24))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_results_fixture_is_flagged_and_consistent():
    races = cli.load_results(2024)
    assert len(races) == 24
    for r in races:
        assert r['winner'] == r['podium'][0] and len(r['podium']) == 3
    with open(cli.__file__.replace('cli.py', 'data/results_2024.json')) as f:
        assert 'HAND-ENTERED' in json.load(f)['_note']


def test_predict_requires_offline_or_fixture(capsys):
    assert cli.main(['predict', '--race', 'Bahrain']) == 2
    assert 'offline' in capsys.readouterr().err


@pytest.mark.gpu
def test_cli_predict_end_to_end(require_gpu, tmp_path, capsys):
    out = tmp_path / 'res.json'
    assert cli.main(['predict', '--race', 'Bahrain', '--season', '2024', '--simulations', '10000', '--seed', '42',
                     '--offline', '--json', str(out)]) == 0
    text = capsys.readouterr().out
    assert 'RACE WINNER PROBABILITIES' in text and 'POLE POSITION PROBABILITIES' in text
    res = json.loads(out.read_text())
    assert abs(sum(res['win_probabilities'].values()) - 1.0) < 1e-9
    assert max(res['win_probabilities'], key=res['win_probabilities'].get) == 'VER'
    # --simulations and --seed are honoured: same call, same numbers; the run is the oracle's
    import oracle_py as O
    from monte_carlo_gp_amd.predictor import F1Predictor
    inp = F1Predictor().simulator_inputs(cli.synthetic_fixture(), 'Bahrain')
    case = dict(config=inp['config'].__dict__, grid_probs=inp['grid_probs'], base_pace=inp['base_pace'],
                tire_deg=inp['tire_deg'], driver_variance=inp['driver_variance'],
                driver_dnf_rates=inp['driver_dnf_rates'], track_condition=inp['track_condition'])
    ref = O.Problem(case).run(10000, rng=O.RNG_PHILOX, seed=42)['hist']
    for i, d in enumerate(inp['drivers']):
        assert res['win_probabilities'][d] == ref[i, 0] / 10000


@pytest.mark.gpu
def test_backtest_sweep_on_gpu(require_gpu):
    res = cli.backtest([2024], seed=42, n_simulations=200000)
    assert res['n_races'] == 24
    assert 0.0 < res['win_brier'] < 0.2 and 0.0 <= res['podium_accuracy'] <= 1.0
    again = cli.backtest([2024], seed=42, n_simulations=200000)
    assert again['win_brier'] == res['win_brier'] and again['pole_brier'] == res['pole_brier']
