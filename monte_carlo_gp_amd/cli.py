"""Command line: the reference's main.py / backtest.py flags, working offline.

    python -m monte_carlo_gp_amd.cli predict  --race Bahrain --season 2024 --simulations 10000 --seed 42 --offline
    python -m monte_carlo_gp_amd.cli backtest --seasons 2024 --seed 42 --simulations 10000000 [--fixtures DIR]
    python -m monte_carlo_gp_amd.cli export-fixtures --seasons 2024 --out DIR

Flags kept from the reference: --season, --race, --prediction-point, --simulations (main.py:8-16);
--seasons, --seed (backtest.py:9-14).  Unlike the reference, --simulations and --seed reach the
simulator (the reference parses --simulations and drops it, main.py:14-15 vs predictor.py:284).
--offline / --fixture replace the FastF1 sessions with a race fixture (see predictor.py); without
--fixture a synthetic weekend is used (SURVEY.md 8d canonical inputs) and labelled as such.
Under torch.distributed.run the backtest shards RACES over ranks (independent problems, no collective
on the data path; results are gathered once).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

from . import config as K
from .predictor import F1Predictor, circuit_info
from .validation import brier_score, calibration_analysis, podium_accuracy


def synthetic_fixture(drivers=None) -> dict:
    """Canonical synthetic weekend: Elo 1700 - 20 i, pace 90 + 0.1 i, deg 0.05, empty features."""
    drivers = list(drivers or K.DRIVER_TEAMS)
    return dict(
        drivers=drivers,
        quali_ratings={d: 1700.0 - 20.0 * i for i, d in enumerate(drivers)},
        quali_features={}, race_features={},
        practice=dict(base_pace={d: 90.0 + 0.1 * i for i, d in enumerate(drivers)},
                      tire_deg={d: 0.05 for d in drivers}, tire_compounds=None),
        weather={'rainfall': False},
        synthetic=True,
    )


def _bars(title, probs, top=10):
    print(title)
    print('-' * 40)
    for i, (d, p) in enumerate(sorted(probs.items(), key=lambda kv: kv[1], reverse=True)[:top], 1):
        print(f"{i:2}. {d:4} {p:6.1%} {'#' * int(p * 30)}")


def cmd_predict(args) -> int:
    fixture = synthetic_fixture()
    if args.fixture:
        with open(args.fixture) as f:
            fixture = json.load(f)
    elif not args.offline:
        print('error: live FastF1 data is not available in this build; use --offline or --fixture FILE', file=sys.stderr)
        return 2
    print(f"\n{'=' * 60}\nF1 Race Prediction: {args.season} {args.race}\nPrediction point: {args.prediction_point}")
    print(f"Simulations: {args.simulations}  seed: {args.seed}  data: "
          f"{'synthetic fixture' if fixture.get('synthetic') else args.fixture}\n{'=' * 60}\n")
    t0 = time.perf_counter()
    res = F1Predictor(device=args.device).predict_weekend(
        args.season, args.race, fixture, prediction_point=args.prediction_point,
        n_simulations=args.simulations, seed=args.seed)
    dt = time.perf_counter() - t0
    print(f"Weather: {'Wet' if res['weather'].get('rainfall') else 'Dry'}")
    print(f"Confidence: {res['confidence']}   ({args.simulations / dt:,.0f} simulations/s incl. setup)\n")
    _bars('POLE POSITION PROBABILITIES', res['pole_probabilities'])
    print()
    _bars('RACE WINNER PROBABILITIES', res['win_probabilities'])
    print()
    _bars('PODIUM PROBABILITIES', res['podium_probabilities'])
    if args.json:
        with open(args.json, 'w') as f:
            json.dump({k: v for k, v in res.items() if k != 'full_distributions'}, f)
    return 0


def load_results(season: int) -> list:
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', f'results_{season}.json')
    with open(path) as f:
        return json.load(f)['races']


def _update_known_pairs(elo, key: str, drivers: list, ranked: list) -> None:
    """The reference's all-pairs Elo update (src/elo.py:45-122: deltas from the ratings BEFORE the event,
    K (actual - expected) / (n - 1) per pair) restricted to the pairs whose outcome the fixture knows:
    every driver in `ranked` beat every driver after it in `ranked` and every driver not in it."""
    n = len(drivers)
    before = {d: elo.ratings[d][key] for d in drivers}
    delta = {d: 0.0 for d in drivers}
    for i, a in enumerate(ranked):
        beaten = ranked[i + 1:] + [d for d in drivers if d not in ranked]
        for b in beaten:
            delta[a] += elo.k * (1.0 - elo.expected_score(before[a], before[b])) / (n - 1)
            delta[b] += elo.k * (0.0 - elo.expected_score(before[b], before[a])) / (n - 1)
    for d in drivers:
        elo.ratings[d][key] = before[d] + delta[d]


def season_fixtures(season: int, entries: list) -> list:
    """One race fixture per race of the season: the synthetic weekend with the Elo quali ratings the
    EARLIER races of that season produced, in calendar order.

    Mirrors the loop of reference validation.py:179-205: predict race k, then update the Elo system with
    race k's outcome before predicting race k+1 (K grows with the race index, reference src/elo.py:13-38).
    The outcome fixture only holds pole, winner and podium, so the update covers the pairs it knows
    (_update_known_pairs): the pole sitter out-qualified everybody; the podium finishers beat everybody
    behind them.  (As written, the reference's own update at validation.py:195-199 passes bare driver codes
    where update_*_ratings expects (driver, value) pairs; the ValueError is swallowed by its `except
    Exception: pass`, so its Elo never moves during a backtest.  This sweep applies the update that loop
    intends.)  Deterministic and cheap (24 x O(n^2) on the host): every rank builds the whole list, then
    races shard over ranks.
    """
    from .elo import F1EloSystem
    base = synthetic_fixture()
    drivers = base['drivers']
    elo = F1EloSystem()
    for d in drivers:
        elo.ratings[d] = {'quali': base['quali_ratings'][d], 'race': base['quali_ratings'][d]}
    out = []
    total = len(entries)
    for idx, entry in enumerate(entries):
        out.append(dict(base, quali_ratings={d: elo.ratings[d]['quali'] for d in drivers},
                        race_ratings={d: elo.ratings[d]['race'] for d in drivers}, race_index=idx))
        elo.set_recency_weight(0, idx, total)
        if entry.get('pole') in elo.ratings:
            _update_known_pairs(elo, 'quali', drivers, [entry['pole']])
        podium = [d for d in entry.get('podium', []) if d in elo.ratings]
        if podium:
            _update_known_pairs(elo, 'race', drivers, podium)
    return out


def fixture_file_name(season: int, index: int, race: str) -> str:
    """File name of a per-race fixture in a --fixtures directory: <season>_<NN>_<race with underscores>.json."""
    return f"{season}_{index + 1:02d}_{race.replace(' ', '_')}.json"


def backtest_jobs(seasons, seed, fixtures_dir=None):
    """(season, result entry, per-race seed, race fixture) for every race of the sweep, in calendar order.

    With fixtures_dir, a race whose file (fixture_file_name) exists there is predicted from THAT fixture -- what the
    reference's per-race practice-session extraction (src/predictor.py:409-569, out of scope) would hand over --
    and the others from the synthetic weekend with the evolved Elo ratings (season_fixtures)."""
    rng = random.Random(seed)
    jobs = []
    for season in seasons:
        entries = load_results(season)
        for idx, (entry, fx) in enumerate(zip(entries, season_fixtures(season, entries))):
            if fixtures_dir:
                path = os.path.join(fixtures_dir, fixture_file_name(season, idx, entry['race']))
                if os.path.exists(path):
                    with open(path) as f:
                        fx = dict(json.load(f), fixture_file=path)
            jobs.append((season, entry, rng.getrandbits(63), fx))
    return jobs


def shard_jobs(jobs, rank, world):
    """Round-robin share of the races for one rank: [(global index, job)]."""
    return [(i, j) for i, j in enumerate(jobs) if i % world == rank]


BATCH_MAX_SIMULATIONS = 1_000_000        # per race: up to here the races of a sweep share one launch


def backtest(seasons, seed=42, n_simulations=10000, device=0, rank=0, world=1, predictor_factory=None,
             fixtures_dir=None):
    """Sweep one prediction per race of each season and score it (reference validation.py:161-209).

    A fresh predictor per race, fed that race's fixture (season_fixtures: Elo evolved over the earlier
    races).  Each race gets its own seed drawn from random.Random(seed) (the reference seeds the global
    streams once and lets them run on, :172-174; per-race seeds keep races independent so they can
    shard over GPUs: rank r takes races r, r + world, ...; no collective on the data path, one
    all_gather_object of the per-race rows at the end).  Returns the reference's result dict plus
    per-race rows.
    """
    mine = shard_jobs(backtest_jobs(seasons, seed, fixtures_dir), rank, world)
    factory = predictor_factory or (lambda: F1Predictor(device=device))
    # At the reference's size (10 000 simulations per race, src/predictor.py:284) a launch is as long as one race of
    # one lane: this rank's races then go to the device in ONE launch (run_monte_carlo_batch; every race's histogram is
    # what its own launch would give).  Above BATCH_MAX_SIMULATIONS a race fills the device by itself.
    batched = {}
    if predictor_factory is None and 0 < n_simulations <= BATCH_MAX_SIMULATIONS and mine:
        from .predictor import pack_result
        from .simulation import run_monte_carlo_batch
        todo = [(i, job) for i, job in mine if job[3].get('drivers')]       # (a weekend without data raises below, as ever)
        inputs = [F1Predictor(device=device).simulator_inputs(fixture, entry['race'])
                  for _, (season, entry, race_seed, fixture) in todo]
        outs = run_monte_carlo_batch([dict(inp, seed=job[2]) for inp, (_, job) in zip(inputs, todo)], n_simulations,
                                     device=device)
        for (i, _), inp, (probs, _) in zip(todo, inputs, outs):
            batched[i] = pack_result(inp['drivers'], inp['grid_probs'], probs, inp['weather'], 'fp2', None)
    rows = []
    for i, (season, entry, race_seed, fixture) in mine:
        res = batched[i] if i in batched else factory().predict_weekend(
            season, entry['race'], fixture, n_simulations=n_simulations, seed=race_seed)
        rows.append((i, dict(race=entry['race'], season=season, laps=circuit_info(entry['race'])['laps'],
                             pole=res['pole_probabilities'], win=res['win_probabilities'],
                             podium_probabilities=res['podium_probabilities'], actual=entry, seed=race_seed,
                             fixture='synthetic' if fixture.get('synthetic') and not fixture.get('fixture_file')
                             else fixture.get('fixture_file', 'given'))))
    from .distributed import wants_process_group
    if wants_process_group(world):
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
    rows = [r for _, r in sorted(rows, key=lambda t: t[0])]
    preds = [dict(pole_probabilities=r['pole'], win_probabilities=r['win'],
                  podium_probabilities=r['podium_probabilities']) for r in rows]
    acts = [r['actual'] for r in rows]
    return {
        'pole_brier': float(brier_score([p['pole_probabilities'] for p in preds], [a['pole'] for a in acts])),
        'win_brier': float(brier_score([p['win_probabilities'] for p in preds], [a['winner'] for a in acts])),
        'podium_accuracy': podium_accuracy(preds, acts),
        'calibration_curve': calibration_analysis([dict(win_probabilities=r['win']) for r in rows], acts),
        'n_races': len(rows),
        'races': rows,
        # what these scores are NOT: the reference's backtest reads live FastF1 sessions and results; here the weekends
        # are fixtures and the outcomes a hand-entered file, so the numbers are not comparable with a reference run
        'reference_comparable': False,
    }


def cmd_backtest(args) -> int:
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    device = int(os.environ.get('LOCAL_RANK', str(args.device)))
    # Rehearsal on a one-GPU box (tests): MCGP_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and gathers over
    # gloo (RCCL refuses two ranks on one device).  Never set in production launches.
    share = os.environ.get('MCGP_BENCH_SHARE_GPU') == '1'
    if share:
        device = args.device
    from .distributed import wants_process_group
    grouped = wants_process_group(world)
    if grouped:
        from . import _native
        _native.lib()                     # build / load once before any rank touches the GPU
        import torch
        import torch.distributed as dist
        if share:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(device)
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
    t0 = time.perf_counter()
    res = backtest(args.seasons, args.seed, args.simulations, device, rank, world, fixtures_dir=args.fixtures)
    dt = time.perf_counter() - t0
    if rank == 0:
        given = sum(1 for r in res['races'] if r['fixture'] != 'synthetic')
        print(f"\n{'=' * 60}\nBacktest (offline sweep: {given} race fixture(s) from --fixtures, the rest synthetic weekends with Elo\n"
              f"evolved race by race; hand-entered outcomes -- NOT comparable with the reference's live-data backtest)\n"
              f"Seasons: {args.seasons}   simulations per race: {args.simulations}\n{'=' * 60}\n")
        print(f"Races analyzed: {res['n_races']}   ({res['n_races'] * args.simulations / dt:,.0f} simulations/s overall)\n")
        print('BRIER SCORES (lower = better, 0 = perfect)\n' + '-' * 40)
        print(f"  Pole position: {res['pole_brier']:.4f}\n  Race winner:   {res['win_brier']:.4f}")
        print(f"  (Random baseline: {0.0475:.4f})\n")
        print('PODIUM ACCURACY\n' + '-' * 40 + f"\n  Correct podium picks: {res['podium_accuracy']:.1%}\n")
        cal = res['calibration_curve']
        if cal['prob_pred']:
            print('CALIBRATION (win probability: predicted -> observed)\n' + '-' * 40)
            for pp, pt in zip(cal['prob_pred'], cal['prob_true']):
                print(f"  {pp:6.1%} -> {pt:6.1%}")
            print()
        if args.json:
            with open(args.json, 'w') as f:
                json.dump(res, f)
    if grouped:
        dist.destroy_process_group()
    return 0


def cmd_export_fixtures(args) -> int:
    """One JSON file per race (fixture_file_name): the synthetic weekend with that race's Elo ratings -- a template
    to replace with real practice data (base_pace / tire_deg per driver, features, weather)."""
    os.makedirs(args.out, exist_ok=True)
    n = 0
    for season in args.seasons:
        entries = load_results(season)
        for idx, (entry, fx) in enumerate(zip(entries, season_fixtures(season, entries))):
            with open(os.path.join(args.out, fixture_file_name(season, idx, entry['race'])), 'w') as f:
                json.dump(fx, f, indent=1)
            n += 1
    print(f'{n} race fixtures written to {args.out}')
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog='monte_carlo_gp_amd', description='F1 race prediction on MI355X')
    sub = ap.add_subparsers(dest='cmd', required=True)
    p = sub.add_parser('predict', help='predict one race weekend (main.py of the reference)')
    p.add_argument('--season', type=int, default=2025)
    p.add_argument('--race', type=str, required=True)
    p.add_argument('--prediction-point', type=str, default='fp2', choices=['fp1', 'fp2', 'fp3', 'quali', 'sprint'])
    p.add_argument('--simulations', type=int, default=10000)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--offline', action='store_true', help='use the synthetic weekend fixture')
    p.add_argument('--fixture', type=str, default=None, help='race fixture JSON (see predictor.py)')
    p.add_argument('--device', type=int, default=0)
    p.add_argument('--json', type=str, default=None)
    p.set_defaults(fn=cmd_predict)
    b = sub.add_parser('backtest', help='sweep a season and score it (backtest.py of the reference)')
    b.add_argument('--seasons', type=int, nargs='+', default=[2024])
    b.add_argument('--seed', type=int, default=42)
    b.add_argument('--simulations', type=int, default=10000)
    b.add_argument('--device', type=int, default=0)
    b.add_argument('--json', type=str, default=None)
    b.add_argument('--fixtures', type=str, default=None,
                   help='directory of per-race fixture files (see export-fixtures); races without a file use the synthetic weekend')
    b.set_defaults(fn=cmd_backtest)
    e = sub.add_parser('export-fixtures', help='write the per-race fixtures of the offline sweep as editable JSON files')
    e.add_argument('--seasons', type=int, nargs='+', default=[2024])
    e.add_argument('--out', type=str, required=True)
    e.set_defaults(fn=cmd_export_fixtures)
    args = ap.parse_args(argv)
    return args.fn(args)


if __name__ == '__main__':
    sys.exit(main())
