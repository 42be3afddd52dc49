"""Multi-GPU sharding of one Monte Carlo run: one process per GPU, simulations split by global
simulation id, and ONE all-reduce of the n x n histogram (RCCL over xGMI; `nccl` backend on ROCm).

The reference has no parallelism; simulations are independent
(reference src/simulation.py:83-94), and with the counter-based RNG the result of simulation i
depends on (seed, i) only, so any partition of [0, N) gives the same total histogram.
The all-reduce carries 8 n^2 bytes (3.2 KB at n = 20): latency-bound, no bucketing or ring
tuning is needed, and it is the path's only exchange.
"""
from __future__ import annotations

import os

import numpy as np


def wants_process_group(world: int) -> bool:
    """A run is distributed when it has more than one rank -- or when MCGP_FORCE_PROCESS_GROUP=1 asks for the
    process group and its collectives even at world size 1 (how the RCCL code path is exercised on a one-GPU
    box: RCCL admits one rank per device)."""
    return world > 1 or os.environ.get('MCGP_FORCE_PROCESS_GROUP') == '1'


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous share of [0, n_total) for `rank`: (offset, count); the remainder goes to the first ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f'bad rank/world {rank}/{world}')
    base, rem = divmod(int(n_total), world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def all_reduce_histogram(hist: np.ndarray, device=None, group=None) -> np.ndarray:
    """Sum an integer histogram over all ranks of the default (or given) process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or not wants_process_group(dist.get_world_size(group)):
        return hist
    t = torch.from_numpy(np.ascontiguousarray(hist.astype(np.int64)))
    if dist.get_backend(group) == 'nccl':
        from . import _native
        _native.assert_single_hip_runtime()       # torch.cuda and the library must share one HIP runtime
        t = t.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def run_sharded(run_shard, n_total: int, rank: int, world: int, device=None, group=None) -> np.ndarray:
    """Run `run_shard(offset, count) -> int histogram` on this rank's share and all-reduce the result."""
    offset, count = shard_range(n_total, rank, world)
    hist = run_shard(offset, count)
    return all_reduce_histogram(np.asarray(hist), device=device, group=group)


def run_monte_carlo_sharded(sim, n_simulations, grid_probs, base_pace, tire_deg, driver_variance,
                            driver_dnf_rates=None, seed=0, track_condition='dry', group=None):
    """RaceSimulator.run_monte_carlo over all ranks of the initialised process group.

    Every rank returns the same {driver: {position: probability}} dict.  `seed` must be the same on
    every rank (seed=None would draw a different seed per process and is rejected).
    """
    import torch.distributed as dist
    from .simulation import histogram_to_probs
    if seed is None:
        raise ValueError('a sharded run needs an explicit seed shared by all ranks')
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1

    def run_shard(offset, count):
        sim.run_monte_carlo(count, grid_probs, base_pace, tire_deg, driver_variance, driver_dnf_rates,
                            seed=seed, track_condition=track_condition, sim_offset=offset)
        return sim.last_histogram

    hist = run_sharded(run_shard, n_simulations, rank, world, group=group)
    sim.last_histogram = hist
    return histogram_to_probs(hist, [str(d) for d in grid_probs.keys()], n_simulations)
