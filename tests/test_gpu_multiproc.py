"""The product's multi-rank entry points on the HIP kernel, in fresh child processes.

What an 8-GPU launch runs per rank -- distributed.run_monte_carlo_sharded, `cli backtest` under
WORLD_SIZE > 1 -- executed here with 2 (and 3) ranks sharing GPU 0 over gloo (RCCL refuses two ranks on
one device), plus the first-use race on the device context: a process whose first library calls are four
concurrent mcgp_run calls.  conftest.py schedules this module before every other GPU test and the parent
never initialises the GPU here: children are started while the parent process is still GPU-free, and
results are compared with the CPU oracle.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'mp_worker.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run_ranks(world, argv_of_rank, timeout=600, extra_env=None):
    """Start `world` children (rank r runs argv_of_rank(r)), wait for all, fail loudly with their output."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), MCGP_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, WORKER] + [str(a) for a in argv_of_rank(r)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f'rank {r} exited {p.returncode}:\n{outs[r][-4000:]}'
    return outs


def test_first_use_from_four_threads(tmp_path):
    """ADVICE r1 / VERDICT item 6: context initialisation under concurrent first calls."""
    out = tmp_path / 'first.npz'
    n_threads, n_sims = 4, 6000
    _run_ranks(1, lambda r: ['firstuse', out, n_threads, n_sims])
    got = np.load(out)
    names = ['S50', 'WET', 'N10', 'S60']
    for i in range(n_threads):
        ref = O.Problem(O.load_case(names[i])).run(n_sims, rng=O.RNG_PHILOX, seed=500 + i)['hist']
        assert np.array_equal(got[f'h{i}'], ref), i


@pytest.mark.parametrize('world', [2, 3])
def test_run_monte_carlo_sharded_on_hip(tmp_path, world):
    """distributed.run_monte_carlo_sharded with the HIP kernel under it: every rank ends with the histogram
    of the single-process run, which equals the oracle's (reference src/simulation.py:83-94: sims independent)."""
    name, n_total, seed = 'S60', 200_001, 42
    _run_ranks(world, lambda r: ['sharded', tmp_path / f'h{r}.npy', name, n_total, seed])
    _run_ranks(1, lambda r: ['single', tmp_path / 'single.npy', name, n_total, seed])
    single = np.load(tmp_path / 'single.npy')
    assert int(single.sum()) == n_total * single.shape[0]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'h{r}.npy'), single), r
    # oracle slice: the first 3000 simulations of the same run, and a piece straddling the 2-rank shard seam
    P = O.Problem(O.load_case(name))
    _run_ranks(1, lambda r: ['single', tmp_path / 'head.npy', name, 3000, seed])
    assert np.array_equal(np.load(tmp_path / 'head.npy'), P.run(3000, rng=O.RNG_PHILOX, seed=seed)['hist'])


def test_cli_backtest_world2_equals_world1(tmp_path):
    """`cli backtest` under WORLD_SIZE=2 (races sharded round-robin over ranks, one all_gather_object) gives the
    Brier scores and per-race rows of the single-process sweep (reference src/validation.py:161-209)."""
    n_sims, seed = 20000, 42
    _run_ranks(2, lambda r: ['backtest', tmp_path / f'bt2_{r}.json', n_sims, seed])
    _run_ranks(1, lambda r: ['backtest', tmp_path / 'bt1.json', n_sims, seed], extra_env={'WORLD_SIZE': '1'})
    with open(tmp_path / 'bt1.json') as f:
        one = json.load(f)
    with open(tmp_path / 'bt2_0.json') as f:
        two = json.load(f)
    assert not os.path.exists(tmp_path / 'bt2_1.json')          # only rank 0 reports
    assert one['n_races'] == two['n_races'] == 24
    for k in ('pole_brier', 'win_brier', 'podium_accuracy'):
        assert one[k] == two[k], k
    assert [r['race'] for r in one['races']] == [r['race'] for r in two['races']]
    for a, b in zip(one['races'], two['races']):
        assert a['win'] == b['win'] and a['pole'] == b['pole'], a['race']
    # races differ by more than circuit constants: the pole distribution moves with the Elo trajectory
    poles = [max(r['pole'], key=r['pole'].get) for r in one['races']]
    assert one['races'][0]['pole'] != one['races'][-1]['pole']
    assert len(set(tuple(sorted(r['pole'].items())) for r in one['races'])) > 12, poles


def test_bench_two_ranks_sharing_the_gpu(tmp_path):
    """bench.py's N > 1 path end to end (rank offsets, per-step histogram reduce, MAX-over-ranks timing, rank-0 JSON),
    launched the way the driver launches it, with the two ranks sharing GPU 0 over gloo (MCGP_BENCH_SHARE_GPU=1:
    RCCL refuses two ranks on one device; the RCCL branch itself needs a multi-GPU node)."""
    env = dict(os.environ, MCGP_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(os.path.dirname(HERE), 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--sims-per-step', '300000']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['config']['sims_per_gpu_per_step'] == 300000 and 'cpu_baseline' not in d
    # the win probabilities come from 2 ranks x 2 steps x 3e5 = 1.2e6 simulations of S60: VER wins ~54 %
    assert abs(d['win_probability_top3']['VER'] - 0.5445) < 0.005


def test_bench_without_a_launcher_launches_itself(tmp_path):
    """`python bench.py --gpus 2` with NO launcher environment: bench.py starts torch.distributed.run as a
    child process and relays rank 0's line (two ranks sharing GPU 0 over gloo here)."""
    env = dict(os.environ, MCGP_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
           '--sims-per-step', '200000', '--no-extras']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert 'starting -m torch.distributed.run' in p.stderr
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['process_group'] == 'gloo' and d['value'] > 0 and len(d['devices']) == 2


# ---------------------------------------------------------------------------------------------------------
# The RCCL branch on the one GPU there is: a world-size-1 `nccl` group (RCCL admits one rank per device), with
# MCGP_FORCE_PROCESS_GROUP=1 keeping the process group and its collectives although nothing needs reducing.
# ---------------------------------------------------------------------------------------------------------
_NCCL1 = {'MCGP_FORCE_PROCESS_GROUP': '1', 'MCGP_BENCH_SHARE_GPU': '0', 'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'}


def test_library_first_then_torch_cuda_share_one_hip_runtime(tmp_path):
    """run_monte_carlo BEFORE torch.cuda is touched, then torch's current stream and a side
    stream into mcgp_run_device, then all_reduce_histogram's device branch -- one HIP runtime throughout."""
    out = tmp_path / 'rt.npz'
    n_sims, seed = 30000, 42
    _run_ranks(1, lambda r: ['runtime', out, n_sims, seed], extra_env=_NCCL1)
    got = np.load(out)
    ref = O.Problem(O.load_case('S60')).run(n_sims, rng=O.RNG_PHILOX, seed=seed)['hist']
    for k in ('first', 'current', 'side', 'reduced'):
        assert np.array_equal(got[k], ref), k


def test_run_monte_carlo_sharded_through_rccl_world1(tmp_path):
    """distributed.run_monte_carlo_sharded with a real `nccl` process group (one rank): the int64 histogram is
    all-reduced ON THE DEVICE by RCCL and the result is the oracle's."""
    name, n_total, seed = 'S78', 50_000, 7
    out = tmp_path / 'h.npy'
    _run_ranks(1, lambda r: ['nccl1', out, name, n_total, seed], extra_env=_NCCL1)
    ref = O.Problem(O.load_case(name)).run(n_total, rng=O.RNG_PHILOX, seed=seed)['hist']
    assert np.array_equal(np.load(out), ref)


def test_cli_backtest_through_rccl_world1(tmp_path):
    """`cli backtest` with the nccl group created (one rank): all_gather_object over RCCL; same scores as without."""
    n_sims, seed = 20000, 42
    _run_ranks(1, lambda r: ['backtest', tmp_path / 'nccl.json', n_sims, seed], extra_env=_NCCL1)
    _run_ranks(1, lambda r: ['backtest', tmp_path / 'plain.json', n_sims, seed], extra_env={'WORLD_SIZE': '1'})
    with open(tmp_path / 'nccl.json') as f:
        a = json.load(f)
    with open(tmp_path / 'plain.json') as f:
        b = json.load(f)
    assert a['n_races'] == b['n_races'] == 24
    for k in ('pole_brier', 'win_brier', 'podium_accuracy'):
        assert a[k] == b[k], k


def test_bench_through_rccl_world1(tmp_path):
    """bench.py's N > 1 code path on RCCL: `nccl` process group, per-step int64 all_reduce on the launch stream,
    barrier + MAX-over-ranks timing -- at world size 1, launched through torch.distributed.run like the driver does."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MCGP_FORCE_PROCESS_GROUP='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MCGP_BENCH_SHARE_GPU'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(os.path.dirname(HERE), 'bench.py'), '--gpus', '1', '--steps', '3',
           '--warmup', '1', '--sims-per-step', '1000000', '--no-cpu-baseline', '--no-extras']
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['process_group'] == 'nccl' and d['value'] > 0
    assert abs(d['win_probability_top3']['VER'] - 0.5445) < 0.005
    # the roofline block names the binding resource and keeps the nominal HBM figure beside it
    roof = d['roofline']
    assert roof['bound'] == 'valu-issue' and roof['kernel'] == 'mcgp::race_kernel_reg<20>' and roof['kernel_ms_avg'] > 0
    assert roof['hbm_nominal']['bound'] == 'hbm' and 0 < roof['hbm_nominal']['frac'] < 1e-3
    assert (roof['frac'] is None) == (roof['counters_note'] is not None)      # counters quoted, or the reason why not
