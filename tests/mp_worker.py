"""Child-process entry for the multi-process GPU tests (tests/test_gpu_multiproc.py).

Each mode exercises a PRODUCT entry point on the HIP kernel in a fresh process and writes its result to
a file; the parent (which never touches the GPU for these tests) compares the files with the CPU oracle.

    python tests/mp_worker.py sharded  <out.npy> <case> <n_total> <seed>     (RANK / WORLD_SIZE / MASTER_* in env)
    python tests/mp_worker.py single   <out.npy> <case> <n_total> <seed>
    python tests/mp_worker.py backtest <out.json> <n_sims> <seed>            (cli.main, RANK / WORLD_SIZE in env)
    python tests/mp_worker.py firstuse <out.npz> <n_threads> <n_sims>

Ranks of one test share GPU 0 and talk over gloo (RCCL refuses two ranks on one device); what runs on
the GPU is exactly what an 8-GPU launch runs per rank.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def _case(name):
    with open(os.path.join(HERE, 'golden', 'cases.json')) as f:
        meta = json.load(f)
    return meta['cases'][name], meta['set_pop']


def _sim(case, set_pop):
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator
    return RaceSimulator(RaceConfig(**case['config']), device=0, set_pop=set_pop)


def mode_sharded(out, name, n_total, seed):
    import torch.distributed as dist
    from monte_carlo_gp_amd.distributed import run_monte_carlo_sharded
    dist.init_process_group('gloo')
    case, set_pop = _case(name)
    sim = _sim(case, set_pop)
    probs = run_monte_carlo_sharded(sim, int(n_total), case['grid_probs'], case['base_pace'], case['tire_deg'],
                                    case['driver_variance'], case['driver_dnf_rates'], seed=int(seed),
                                    track_condition=case['track_condition'])
    # every rank must hold the reduced histogram and the same probabilities
    drivers = list(case['grid_probs'])
    assert abs(sum(probs[drivers[0]].values()) - 1.0) < 1e-12
    np.save(out, sim.last_histogram)
    dist.barrier()
    dist.destroy_process_group()


def mode_single(out, name, n_total, seed):
    case, set_pop = _case(name)
    sim = _sim(case, set_pop)
    sim.run_monte_carlo(int(n_total), case['grid_probs'], case['base_pace'], case['tire_deg'],
                        case['driver_variance'], case['driver_dnf_rates'], seed=int(seed),
                        track_condition=case['track_condition'])
    np.save(out, sim.last_histogram)


def mode_backtest(out, n_sims, seed):
    from monte_carlo_gp_amd import cli
    rc = cli.main(['backtest', '--seasons', '2024', '--seed', str(seed), '--simulations', str(n_sims), '--json', out])
    assert rc == 0


def mode_firstuse(out, n_threads, n_sims):
    """The process's FIRST library calls are n_threads concurrent mcgp_run calls (context initialisation race)."""
    import threading
    from helpers import product_run            # imports only; the library is not loaded yet
    import oracle_py as O
    from monte_carlo_gp_amd import _native
    _native.build()                            # compile if needed, but do not dlopen / initialise anything
    assert _native._lib is None
    names = ['S50', 'WET', 'N10', 'S60']
    cases = [O.load_case(names[i % len(names)]) for i in range(int(n_threads))]
    hists = [None] * int(n_threads)
    errors = []
    gate = threading.Barrier(int(n_threads))

    def work(i):
        try:
            gate.wait()
            hists[i] = product_run(cases[i], int(n_sims), 500 + i)[0]
        except Exception as e:                 # noqa: BLE001
            errors.append(repr(e))
    # load the library object itself once (dlopen is not what the test is about: HIP context creation,
    # buffers and events happen on the first mcgp_run)
    _native.lib()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(int(n_threads))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    np.savez(out, **{f'h{i}': h for i, h in enumerate(hists)})


if __name__ == '__main__':
    mode = sys.argv[1]
    {'sharded': mode_sharded, 'single': mode_single, 'backtest': mode_backtest, 'firstuse': mode_firstuse}[mode](*sys.argv[2:])
