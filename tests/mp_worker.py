"""Child-process entry for the multi-process GPU tests (tests/test_gpu_multiproc.py).

Each mode exercises a PRODUCT entry point on the HIP kernel in a fresh process and writes its result to
a file; the parent (which never touches the GPU for these tests) compares the files with the CPU oracle.

    python tests/mp_worker.py sharded  <out.npy> <case> <n_total> <seed>     (RANK / WORLD_SIZE / MASTER_* in env)
    python tests/mp_worker.py single   <out.npy> <case> <n_total> <seed>
    python tests/mp_worker.py backtest <out.json> <n_sims> <seed>            (cli.main, RANK / WORLD_SIZE in env)
    python tests/mp_worker.py firstuse <out.npz> <n_threads> <n_sims>
    python tests/mp_worker.py runtime  <out.npz> <n_sims> <seed>             (library first, torch.cuda second)
    python tests/mp_worker.py nccl1    <out.npy> <case> <n_total> <seed>     (one-rank nccl group: the RCCL branch)

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


def mode_runtime(out, n_sims, seed):
    """Library FIRST, torch.cuda second (`No HIP GPUs are available` when libmcgp_hip.so had initialised
    the GPU before torch): run_monte_carlo, then torch.cuda's stream and tensors handed to mcgp_run_device, then
    all_reduce_histogram's device branch in a one-rank nccl group.  One HIP runtime must serve all of it."""
    import ctypes as C
    assert 'torch' not in sys.modules
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N
    from monte_carlo_gp_amd.simulation import _Problem, _dptr
    case, set_pop = _case('S60')
    sim = _sim(case, set_pop)
    sim.run_monte_carlo(int(n_sims), case['grid_probs'], case['base_pace'], case['tire_deg'],
                        case['driver_variance'], case['driver_dnf_rates'], seed=int(seed),
                        track_condition=case['track_condition'])          # the library initialises the GPU
    first = sim.last_histogram.copy()
    assert 'torch' not in sys.modules and len(N.hip_runtimes_mapped()) == 1, N.hip_runtimes_mapped()
    import torch                                                          # ... and only now torch
    assert len(N.hip_runtimes_mapped()) == 1, N.hip_runtimes_mapped()
    assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    side = torch.cuda.Stream(dev)
    drivers = list(case['grid_probs'])
    p = _Problem(RaceConfig(**case['config']), drivers, case['base_pace'], case['tire_deg'],
                 case['driver_variance'], case['driver_dnf_rates'], case['track_condition'], set_pop)
    g = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
    hists = []
    for st in (torch.cuda.current_stream(dev), side):
        with torch.cuda.stream(st):
            d = torch.zeros(p.n * p.n, dtype=torch.int64, device=dev)
            N.check(N.lib().mcgp_run_device(C.byref(p.cfg), C.byref(p.drv), _dptr(g), p.n, int(n_sims), 0, int(seed), 0,
                                            C.c_void_p(st.cuda_stream), C.c_void_p(d.data_ptr()), None))
            ms = C.c_float()
            N.check(N.lib().mcgp_stream_kernel_ms(0, C.c_void_p(st.cuda_stream), C.byref(ms)))
            assert ms.value > 0
        st.synchronize()
        hists.append(d.cpu().numpy().reshape(p.n, p.n))
    import torch.distributed as dist
    from monte_carlo_gp_amd.distributed import all_reduce_histogram
    dist.init_process_group('nccl', device_id=dev)
    reduced = all_reduce_histogram(first, device=dev)                      # MCGP_FORCE_PROCESS_GROUP=1: really reduces
    assert dist.get_backend() == 'nccl'
    dist.barrier()
    dist.destroy_process_group()
    np.savez(out, first=first, current=hists[0], side=hists[1], reduced=reduced)


def mode_nccl1(out, name, n_total, seed):
    """run_monte_carlo_sharded through its RCCL branch: a one-rank `nccl` group (RCCL admits one rank per device;
    MCGP_FORCE_PROCESS_GROUP=1 keeps the collective although the world size is 1)."""
    from monte_carlo_gp_amd import _native
    _native.lib()
    import torch
    import torch.distributed as dist
    from monte_carlo_gp_amd.distributed import run_monte_carlo_sharded, wants_process_group
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    assert dist.get_backend() == 'nccl' and wants_process_group(dist.get_world_size())
    case, set_pop = _case(name)
    sim = _sim(case, set_pop)
    calls = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        calls.append((t.device.type, t.dtype, tuple(t.shape)))
        return real(t, *a, **k)
    dist.all_reduce = spy
    run_monte_carlo_sharded(sim, int(n_total), case['grid_probs'], case['base_pace'], case['tire_deg'],
                            case['driver_variance'], case['driver_dnf_rates'], seed=int(seed),
                            track_condition=case['track_condition'])
    dist.all_reduce = real
    assert calls and calls[0][0] == 'cuda', calls          # the histogram went through RCCL on the device
    np.save(out, sim.last_histogram)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    mode = sys.argv[1]
    {'sharded': mode_sharded, 'single': mode_single, 'backtest': mode_backtest, 'firstuse': mode_firstuse,
     'runtime': mode_runtime, 'nccl1': mode_nccl1}[mode](*sys.argv[2:])
