"""The N > 1 path on CPU: world_size 2 (and 3) over gloo.  The per-rank compute is the oracle's
Philox back-end (the GPU kernel is bit-identical to it, tests/test_gpu_parity.py), so this checks
exactly what multi-GPU adds: the shard arithmetic and the histogram all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as O
from monte_carlo_gp_amd.distributed import run_sharded, shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 1000, 10 ** 9 + 7):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (o1, c1), (o2, _) in zip(parts[:-1], parts[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, seed, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    P = O.Problem(O.load_case('N10'))
    hist = run_sharded(lambda off, cnt: P.run(cnt, rng=O.RNG_PHILOX, seed=seed, sim_offset=off)['hist'],
                       n_total, rank, world)
    np.save(os.path.join(out_dir, f'hist_{rank}.npy'), hist)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_run_equals_single_process(tmp_path, world):
    n_total, seed = 1501, 13
    mp.spawn(_worker, args=(world, _free_port(), n_total, seed, str(tmp_path)), nprocs=world, join=True)
    single = O.Problem(O.load_case('N10')).run(n_total, rng=O.RNG_PHILOX, seed=seed)['hist']
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'hist_{r}.npy'), single)
