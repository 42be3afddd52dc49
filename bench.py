#!/usr/bin/env python3
"""bench.py -- race-simulations/sec of the HIP hot path on 1..N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path (reference RaceSimulator.run_monte_carlo,
src/simulation.py:59-100) over one batch of synthetic input: 10^7 full-race
simulations per GPU of the S60 workload (20 drivers, 60 laps, Bahrain parameters,
fixed Elo grid, seed 42; BASELINE.json configs[1]), including the
reduction of the driver x position histogram (one RCCL all-reduce of 400 int64
per step when N > 1).  Simulations shard over ranks by global simulation id with
no other exchange ("weak" scaling: per-GPU work is fixed).

The JSON line printed by rank 0 carries, besides the contract fields,
  roofline      algorithmic HBM bytes (20 B per simulation) / measured
                kernel time, against the 8 TB/s HBM peak -- evidence that the path
                is NOT memory bound; the binding resource is VALU issue, reported
                in "valu" as simulated car-laps per second
  cpu_baseline  the CPU oracle (C restatement, Mersenne-Twister back-end = the
                reference-equivalent path) timed on one host core on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402

SIMS_PER_STEP = 10_000_000
ALGORITHMIC_BYTES_PER_SIM = 20          # the n x u8 finishing order, SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s


def load_workload(name):
    with open(os.path.join(ROOT, 'tests', 'golden', 'cases.json')) as f:
        meta = json.load(f)
    return meta['cases'][name], meta['set_pop']


def cpu_baseline(case_name, seconds=12.0):
    """Oracle (kind "port"), MT back-end, one core, bounded sample of the same workload."""
    import oracle_py as O
    P = O.Problem(O.load_case(case_name))
    mt = O.MTState(42)
    P.run(200, rng=O.RNG_MT, mt=mt)            # warm-up / page-in
    done, t0 = 0, time.perf_counter()
    chunk = 2000
    while True:
        P.run(chunk, rng=O.RNG_MT, mt=mt)
        done += chunk
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return dict(value=done / dt, unit='race-simulations/s', cores=1, kind='port',
                sample=f'{done} simulations of {case_name} (MT back-end, reference draw order), {dt:.1f} s, 1 thread')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='S60')
    ap.add_argument('--sims-per-step', type=int, default=SIMS_PER_STEP)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from monte_carlo_gp_amd import RaceConfig, _native as N
    from monte_carlo_gp_amd.simulation import RaceSimulator, _Problem, _dptr

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    # Rehearsal knobs (one-GPU box): MCGP_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and reduces over
    # gloo instead of RCCL (RCCL refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get('MCGP_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    case, set_pop = load_workload(args.workload)
    cfg = RaceConfig(**case['config'])
    drivers = list(case['grid_probs'].keys())
    n = len(drivers)
    L = cfg.total_laps
    prob = _Problem(cfg, drivers, case['base_pace'], case['tire_deg'], case['driver_variance'],
                    case['driver_dnf_rates'], case['track_condition'], set_pop)
    grid = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
    seed = case['seed']
    lib = N.lib()
    per_gpu = args.sims_per_step

    d_hist = torch.zeros(n * n, dtype=torch.int64, device=dev)       # running total (all ranks, all steps)
    d_step = torch.zeros(n * n, dtype=torch.int64, device=dev)       # this step's histogram
    stream = torch.cuda.current_stream(dev)

    def step(index):
        # global simulation ids: [index * world * per_gpu, (index + 1) * world * per_gpu), split by rank
        offset = (index * world + rank) * per_gpu
        d_step.zero_()
        N.check(lib.mcgp_run_device(C.byref(prob.cfg), C.byref(prob.drv), _dptr(grid), n, per_gpu, offset,
                                    seed, local_rank, C.c_void_p(stream.cuda_stream),
                                    C.c_void_p(d_step.data_ptr()), None))
        if world > 1:
            if share:
                h = d_step.cpu()
                dist.all_reduce(h)
                d_step.copy_(h)
            else:
                dist.all_reduce(d_step)        # RCCL over xGMI: 400 x int64, the path's only exchange
        d_hist.add_(d_step)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        step(w)
    sync()
    d_hist.zero_()
    kernel_ms = []
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
        ms = C.c_float()
        # hipEvents recorded on the launch stream around the kernel (read after the step's work is queued;
        # the query synchronises on the stop event only)
        N.check(lib.mcgp_last_kernel_ms(local_rank, C.byref(ms)))
        kernel_ms.append(ms.value)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_sims = per_gpu * world * args.steps
    hist = d_hist.cpu().numpy().reshape(n, n)
    if rank == 0:
        if not os.environ.get('MCGP_BENCH_NOCHECK'):      # diagnostic ablation builds produce wrong results on purpose
            assert int(hist.sum()) == total_sims * n, (int(hist.sum()), total_sims * n)
            assert (hist.sum(axis=1) == total_sims).all() and (hist.sum(axis=0) == total_sims).all()
        kavg_ms = float(np.mean(kernel_ms))
        g, b, lds = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.mcgp_last_launch_info(local_rank, C.byref(g), C.byref(b), C.byref(lds))
        achieved = per_gpu * ALGORITHMIC_BYTES_PER_SIM / (kavg_ms * 1e-3) / 1e9
        kernel_name = lib.mcgp_last_kernel_name(local_rank).decode()
        traffic = None      # HBM bytes per launch from the PMC passes kept under profiles/ (same workload)
        try:
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
                t = json.load(f)
            if t.get('workload') == args.workload and t.get('sims_per_launch') == per_gpu:
                traffic = t['hbm_bytes_per_launch']
        except (OSError, ValueError, KeyError):
            pass
        # VALU issue roofline (the binding resource): wave-level VALU instructions per launch from the PMC pass
        # kept under profiles/ (a property of the workload, not of the box) over the live kernel time;
        # peak = 1024 SIMDs x 2.4 GHz / 2 cycles per 32-bit wave64 instruction (MI355X_MICROARCH.md)
        valu_frac = valu_insts = valu_busy = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'r1_counters.json')) as f:
                pc = json.load(f)
            if args.workload == 'S60' and pc.get('sims_per_launch') == per_gpu:
                valu_insts = pc['counters']['SQ_INSTS_VALU']
                valu_frac = valu_insts / (kavg_ms * 1e-3) / (1024 * 2.4e9 / 2)
                # VALUBusy of the profiled run (gfx9 formula: 4 x SQ_ACTIVE_INST_VALU / SIMDs / busy cycles)
                valu_busy = 4 * pc['counters']['SQ_ACTIVE_INST_VALU'] / 1024 / (pc['counters']['GRBM_GUI_ACTIVE'] / 8)
        except (OSError, ValueError, KeyError):
            pass
        out = {
            'metric': f'race-simulations/sec ({n} drivers, {L} laps)',
            'value': total_sims / elapsed,
            'unit': 'race-simulations/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {n} drivers, {L} laps, {per_gpu} simulations per GPU per step, '
                                   f'fixed Elo grid, seed {seed}',
                       'sims_per_gpu_per_step': per_gpu, 'parallelism': f'sims sharded over {world} GPU(s)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': kernel_name, 'kernel_ms_avg': kavg_ms,
                         'note': 'path is VALU/LDS-issue bound, not HBM bound: 20 algorithmic bytes per simulation'},
            'valu': {'car_laps_per_s': per_gpu * n * L / (kavg_ms * 1e-3),
                     'sims_per_s_kernel_only': per_gpu / (kavg_ms * 1e-3),
                     'valu_insts_per_launch': valu_insts, 'valu_issue_frac_of_peak': valu_frac,
                     'valu_busy_profiled': valu_busy,
                     'launch': {'grid': g.value, 'block': b.value, 'lds_bytes': lds.value}},
            'win_probability_top3': {drivers[i]: float(hist[i, 0]) / total_sims for i in np.argsort(-hist[:, 0])[:3]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
