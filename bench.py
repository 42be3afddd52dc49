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
  roofline       the BINDING resource, VALU instruction issue: wave-level VALU instructions per launch (PMC) / live
                 kernel time (hipEvents on the launch stream) against 1024 SIMDs x 2.4 GHz / 2 cycles (the guide's
                 SIMD-32 rate; `frac`), and against the 4-cycle rate that the datasheet's 78.6 TFLOP/s of vector FP64
                 implies (`frac_at_fp64_rate`), and against the ceiling of the kernel's own instruction mix with issue
                 costs measured per class (`frac_of_mix_ceiling`); the PMC's VALUBusy formula is quoted beside them but is not a
                 utilisation on this hardware (it reads 1.07 for this kernel); `traffic` = HBM bytes per
                 launch from the PMC passes; roofline.hbm_nominal = the 20 algorithmic bytes per simulation
                 against the 8 TB/s HBM peak -- evidence that the path is NOT memory bound.  Counters are quoted only
                 when profiles/r5_counters.json carries the source hash the loaded binary reports (else null, reason in
                 roofline.counters_note)
  workloads.S78  BASELINE configs[2] (78-lap Monaco parameters), three steps in the same run        (N = 1 only)
  orders_mode    the same workload with the 20 B per simulation actually written                    (N = 1 only)
  cpu_baseline   the CPU oracle on this box's host cores, bounded samples: MT back-end on one core (the
                 reference-equivalent path) and Philox back-end on every core of the job's share; nproc, CPU model
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
SWEEP_SIMS_PER_RACE = 10_000_000           # BASELINE configs[4]
ALGORITHMIC_BYTES_PER_SIM = 20          # the n x u8 finishing order, SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TINST = 1024 * 2.4e9 / 2 / 1e12   # 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles at 2.4 GHz
# the rate the datasheet's vector FP64 peak implies: 78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz, i.e. one
# wave64 instruction per 4 cycles.  (The profiler's VALUBusy formula, 4 x SQ_ACTIVE_INST_VALU / SIMDs / cycles, assumes that
# rate for every instruction; with the 2-cycle class in the mix it overshoots 1 -- 1.07 for the S60 kernel -- and is
# quoted as `valu_busy_profiled` for reference only.)
VALU_PEAK_TINST_4CYCLE = 1024 * 2.4e9 / 4 / 1e12
COUNTERS_FILE = 'r5_counters.json'
MIX_FILE = 'r5_valu_mix.json'


def load_workload(name):
    with open(os.path.join(ROOT, 'tests', 'golden', 'cases.json')) as f:
        meta = json.load(f)
    if name not in meta['cases'] and name[:1] == 'N' and name[1:].isdigit():
        # N<k>: a k-car field with S60's parameters (N25 is the profiled one; N22 is a 2026-sized grid)
        base = meta['cases']['S60']
        drivers = [f'D{i:02d}' for i in range(int(name[1:]))]
        teams = list(base['config']['dnf_rates'])
        n = len(drivers)
        sigma = n / 4
        grid = {}
        for i, d in enumerate(drivers):
            w = [np.exp(-((j - i) ** 2) / (2 * sigma ** 2)) for j in range(n)]
            grid[d] = [float(x / sum(w)) for x in w]
        case = dict(base, grid_probs=grid,
                    config=dict(base['config'], driver_teams={d: teams[i % len(teams)] for i, d in enumerate(drivers)}),
                    base_pace={d: 90.0 + 0.1 * i for i, d in enumerate(drivers)}, tire_deg={d: 0.05 for d in drivers},
                    driver_variance={d: 0.18 for d in drivers},
                    driver_dnf_rates={d: 0.05 / base['config']['total_laps'] for d in drivers})
        return case, meta['set_pop']
    return meta['cases'][name], meta['set_pop']


def host_info():
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # CPU share of this job: cgroup v2 quota if one is set, else the affinity mask; a one-GPU box of the pool
    # gives 16 cores to a job whatever nproc says, so the thread count is capped there unless overridden
    threads, source = usable, 'affinity mask'
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            threads, source = max(1, int(int(quota) / int(period))), 'cgroup cpu.max'
    except (OSError, ValueError):
        pass
    env = os.environ.get('MCGP_BENCH_CPU_THREADS')
    if env:
        threads, source = max(1, int(env)), 'MCGP_BENCH_CPU_THREADS'
    elif threads > 16:
        threads, source = 16, f'capped at the 16-core share of a one-GPU job ({source} says {threads})'
    return dict(cpu_model=model, nproc=os.cpu_count() or 1, usable_cores=usable, bench_threads=threads,
                bench_threads_source=source)


def cpu_baseline(case_name, seconds=12.0, all_core_seconds=8.0):
    """The CPU oracle timed on this box's host cores, bounded samples of the same workload:
      * value / cores=1: Mersenne-Twister back-end, one thread -- the reference-equivalent path (kind "port");
      * all_cores: Philox back-end (the arithmetic the GPU kernel runs, draw for draw), one thread per usable
        core over disjoint sim_offset ranges (ctypes releases the GIL)."""
    import oracle_py as O
    from concurrent.futures import ThreadPoolExecutor
    info = host_info()
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
    out = dict(value=done / dt, unit='race-simulations/s', cores=1, kind='port',
               sample=f'{done} simulations of {case_name} (MT back-end, reference draw order), {dt:.1f} s, 1 thread',
               **info,
               reference_python_sims_per_s={'S60': 180, 'S78': 139,
                                            'note': 'the pure-Python reference itself, 1 core of the build container '
                                                    '(SURVEY.md section 6 probe); it cannot travel to the GPU box'})
    # all-core leg: one thread per core of this job's CPU share, each running chunks until a common deadline
    # (bounded wall time whatever the share turns out to be)
    cores = info['bench_threads']
    problems = [O.Problem(O.load_case(case_name)) for _ in range(cores)]
    t0 = time.perf_counter()
    problems[0].run(chunk, rng=O.RNG_PHILOX, seed=42)
    one = chunk / (time.perf_counter() - t0)
    stride = 1 << 32                                   # disjoint simulation-id ranges per thread
    deadline = time.perf_counter() + all_core_seconds

    def work(i):
        n, off = 0, i * stride
        while time.perf_counter() < deadline:
            problems[i].run(chunk, rng=O.RNG_PHILOX, seed=42, sim_offset=off)
            n += chunk
            off += chunk
        return n
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    out['all_cores'] = dict(value=total / dt, unit='race-simulations/s', cores=cores, kind='port',
                            sample=f'{total} simulations of {case_name} (Philox back-end), {dt:.1f} s, {cores} threads '
                                   f"({info['bench_threads_source']})",
                            one_thread_philox=one)
    return out


def profiled_counters(workload, per_gpu, lib_overridden):
    """PMC-derived figures kept under profiles/ (rocprofv3 --pmc passes, tools/summarize_profiles.py), quoted
    ONLY when the file is stamped with the source hash the LOADED binary reports (mcgp_build_hash())."""
    from monte_carlo_gp_amd import _native as N
    path = os.path.join(ROOT, 'profiles', COUNTERS_FILE)
    if lib_overridden:
        return None, 'MCGP_LIB is set: counters under profiles/ belong to the product build'
    try:
        with open(path) as f:
            pc = json.load(f)
    except (OSError, ValueError) as e:
        return None, f'no usable {os.path.relpath(path, ROOT)}: {e}'
    # the identity compiled INTO the mapped binary (mcgp_build_hash()), not a hash of the files on disk: _native.lib()
    # has already refused a binary whose identity is not that of the tree
    if pc.get('source_hash') != N.build_hash():
        return None, (f"profiles/{COUNTERS_FILE} was taken from a binary built from sources {pc.get('source_hash')}, "
                      f'the loaded library reports {N.build_hash()}: re-profile')
    w = pc.get('workloads', {}).get(workload)
    if not w or w.get('sims_per_launch') != per_gpu:
        return None, f'profiles/{COUNTERS_FILE} has no {workload} entry at {per_gpu} simulations per launch'
    return w, None


def relaunch_command(argv, n_gpus, port=None):
    """The torch.distributed.run command line that runs THIS script with one rank per GPU (what the driver itself uses
    for N > 1), given the script's own arguments."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def gpu_initialised():
    """True once this process has touched the GPU through torch (any earlier HIP call of the library is the caller's to know:
    bench.py makes none before this check)."""
    torch = sys.modules.get('torch')
    return bool(torch is not None and torch.cuda.is_initialized())


def self_launch(argv, n_gpus):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: run the launcher as a CHILD process with the same
    arguments and relay its output -- the JSON line of rank 0 -- and its exit code.  A process that has initialised the
    GPU must not do this (and never re-execs itself): on this pool replacing or forking such a process takes the machine
    down, so the request is refused there."""
    import subprocess
    if gpu_initialised():
        raise SystemExit('bench.py: refusing to start torch.distributed.run from a process that has already initialised the GPU; '
                         'start bench.py through the launcher instead')
    from monte_carlo_gp_amd import _native as N
    N.build()                                   # one build before N ranks ask for it (no HIP call: make only)
    cmd = relaunch_command(argv, n_gpus)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    print('bench.py: no launcher environment (WORLD_SIZE unset): starting ' + ' '.join(cmd[1:8]) + ' ... as a child process',
          file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='S60')
    ap.add_argument('--sims-per-step', type=int, default=SIMS_PER_STEP)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the S78 and orders-mode side measurements')
    ap.add_argument('--orders', action='store_true', help='the timed steps also write the per-simulation finishing orders')
    ap.add_argument('--deviates', type=int, default=32, choices=(32, 53),
                    help='53: the timed steps run at the reference\'s deviate width (race_kernel_reg_wide; tools/profile.sh)')
    args = ap.parse_args()

    # started as `python bench.py --gpus N` with no launcher around it: become the launcher's parent (before anything here
    # imports torch or touches the GPU)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from monte_carlo_gp_amd import RaceConfig, _native as N
    from monte_carlo_gp_amd.simulation import RaceSimulator, _Problem, _dptr

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    lib = N.lib()                            # builds once per node (flock) BEFORE anything touches the GPU
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_py                     # ... and so does the CPU oracle of the cpu_baseline leg: a GPU-initialised
        oracle_py.lib()                      # process must not fork + exec make later
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    # Rehearsal knobs (one-GPU box): MCGP_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and reduces over
    # gloo instead of RCCL (RCCL refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get('MCGP_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # MCGP_FORCE_PROCESS_GROUP=1: create the process group and run the per-step all-reduce even at world size 1
    # (exercises the RCCL branch on a one-GPU box).  Never set by the driver.
    from monte_carlo_gp_amd.distributed import wants_process_group
    grouped = wants_process_group(world)
    if grouped:
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    per_gpu = args.sims_per_step
    stream = torch.cuda.current_stream(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run_workload(name, steps, warmup, with_orders=False, deviates=32):
        """`warmup` untimed steps, then exactly `steps` timed steps between barrier + synchronize pairs."""
        case, set_pop = load_workload(name)
        cfg = RaceConfig(**case['config'])
        drivers = list(case['grid_probs'].keys())
        n, L = len(drivers), cfg.total_laps
        prob = _Problem(cfg, drivers, case['base_pace'], case['tire_deg'], case['driver_variance'],
                        case['driver_dnf_rates'], case['track_condition'], set_pop, deviates)
        grid = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
        seed = case['seed']
        d_hist = torch.zeros(n * n, dtype=torch.int64, device=dev)       # running total (all ranks, all steps)
        d_step = torch.zeros(n * n, dtype=torch.int64, device=dev)       # this step's histogram
        d_orders = torch.empty(per_gpu * n, dtype=torch.uint8, device=dev) if with_orders else None

        def step(index):
            # global simulation ids: [index * world * per_gpu, (index + 1) * world * per_gpu), split by rank
            offset = (index * world + rank) * per_gpu
            d_step.zero_()
            N.check(lib.mcgp_run_device(C.byref(prob.cfg), C.byref(prob.drv), _dptr(grid), n, per_gpu, offset,
                                        seed, local_rank, C.c_void_p(stream.cuda_stream),
                                        C.c_void_p(d_step.data_ptr()),
                                        C.c_void_p(d_orders.data_ptr()) if with_orders else None))
            if grouped:
                if share:
                    h = d_step.cpu()
                    dist.all_reduce(h)
                    d_step.copy_(h)
                else:
                    dist.all_reduce(d_step)        # RCCL over xGMI: 400 x int64, the path's only exchange
            d_hist.add_(d_step)

        for w in range(warmup):
            step(w)
        sync()
        d_hist.zero_()
        kernel_ms = []
        sync()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
            ms = C.c_float()
            # hipEvents recorded by the library on the launch stream around the kernel (read after the step's
            # work is queued; the query synchronises on the stop event only)
            N.check(lib.mcgp_last_kernel_ms(local_rank, C.byref(ms)))
            kernel_ms.append(ms.value)
        sync()
        elapsed = time.perf_counter() - t0
        if grouped:
            t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        total = per_gpu * world * steps
        hist = d_hist.cpu().numpy().reshape(n, n)
        if rank == 0 and not os.environ.get('MCGP_BENCH_NOCHECK'):    # diagnostic ablation builds are wrong on purpose
            assert int(hist.sum()) == total * n, (int(hist.sum()), total * n)
            assert (hist.sum(axis=1) == total).all() and (hist.sum(axis=0) == total).all()
        if with_orders and rank == 0:
            o = d_orders[:n * 4096].cpu().numpy().reshape(4096, n)
            assert (np.sort(o, axis=1) == np.arange(n)).all(), 'orders rows are not permutations'
        g, b, lds = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.mcgp_last_launch_info(local_rank, C.byref(g), C.byref(b), C.byref(lds))
        return dict(name=name, n=n, L=L, seed=seed, drivers=drivers, elapsed=elapsed, total=total, hist=hist,
                    kernel_ms=float(np.mean(kernel_ms)), kernel=lib.mcgp_last_kernel_name(local_rank).decode(),
                    launch={'grid': g.value, 'block': b.value, 'lds_bytes': lds.value})

    t_gpu0 = time.perf_counter()
    r = run_workload(args.workload, args.steps, args.warmup, with_orders=args.orders, deviates=args.deviates)
    if rank == 0:
        n, L, kavg_ms, hist = r['n'], r['L'], r['kernel_ms'], r['hist']
        achieved = per_gpu * ALGORITHMIC_BYTES_PER_SIM / (kavg_ms * 1e-3) / 1e9
        pc, why_not = profiled_counters(args.workload, per_gpu, bool(os.environ.get('MCGP_LIB')))
        traffic = None
        roof = {'bound': 'valu-issue', 'achieved': None, 'peak': VALU_PEAK_TINST, 'unit': 'T wave-instructions/s',
                'frac': None, 'traffic': None}
        if pc:
            traffic = pc.get('hbm_bytes_per_launch')
            insts = pc['SQ_INSTS_VALU']
            rate = insts / (kavg_ms * 1e-3) / 1e12
            # VALU issue roofline (the binding resource): wave-level VALU instructions per launch (a property of the
            # workload and the code, stamped with the source hash) over the LIVE kernel time
            roof.update({'achieved': rate, 'frac': rate / VALU_PEAK_TINST, 'traffic': traffic,
                         'peak_at_fp64_rate': VALU_PEAK_TINST_4CYCLE, 'frac_at_fp64_rate': rate / VALU_PEAK_TINST_4CYCLE,
                         'valu_busy_profiled': pc.get('valu_busy'), 'valu_insts_per_launch': insts,
                         'active_lane_ratio': pc.get('active_lane_ratio'), 'waves_per_simd': pc.get('waves_per_simd'),
                         'scratch_bytes_per_lane': int(pc.get('kernel', {}).get('Scratch_Size', 0) or 0),
                         'source_hash': pc.get('source_hash'),
                         'counters': f'profiles/{COUNTERS_FILE} (rocprofv3 --pmc, tools/profile.sh), stamped with the '
                                     'source hash the loaded binary reports (mcgp_build_hash)'})
            # Lane-level view of the same counters.  A wave-instruction issues for 64 lanes whether they
            # are masked or not; weighting by the measured share of active lanes gives the USEFUL lane-operations and
            # their fraction of the lane-level peak (64 lanes x the 2-cycle issue rate).
            alr = pc.get('active_lane_ratio')
            if alr:
                lane_peak = VALU_PEAK_TINST * 1e12 * 64
                roof.update({'lane_ops_per_sim': insts * 64 * alr / per_gpu,
                             'frac_useful_lanes': rate * 1e12 * 64 * alr / lane_peak,
                             'lane_note': 'lane_ops_per_sim = SQ_INSTS_VALU x 64 x active_lane_ratio / simulations (measured); '
                                          "frac_useful_lanes = frac x active_lane_ratio.  SURVEY.md 8(d)'s a-priori figures "
                                          '(5.6e5 lane-ops per simulation, 3.9e13 lane-ops/s peak) are superseded by these '
                                          'measurements: applied to the measured rate they would give a fraction above 1'})
            # the ceiling for THIS kernel's instruction mix: measured issue cost per class (tools/valu_peak.hip: binary64
            # and VOP3 / 64-bit integer instructions ~4.2 cycles per wave64 instruction, VOP1 / VOP2 32-bit ones ~2.25)
            # weighted by the class shares of the lap loop (tools/valu_mix.py), same source hash
            try:
                with open(os.path.join(ROOT, 'profiles', MIX_FILE)) as f:
                    mix = json.load(f)
                if mix.get('source_hash') == pc.get('source_hash'):
                    roof.update({'peak_for_instruction_mix': mix['peak_T_wave_instructions_per_s_for_this_mix'],
                                 'frac_of_mix_ceiling': rate / mix['peak_T_wave_instructions_per_s_for_this_mix'],
                                 'share_four_cycle_class': mix['share_four_cycle'],
                                 'mean_cycles_per_instruction_for_mix': mix['mean_cycles_per_instruction_for_this_mix']})
                    occ = mix.get('at_three_waves_per_simd')
                    if occ and abs(float(roof.get('waves_per_simd') or 0) - 3.0) < 0.2:
                        # the same mix priced at the kernel's own occupancy: at 3 waves per SIMD VOP2 instructions, compares,
                        # 64-bit conversions and v_mad_u64_u32 measure 25-35 % dearer than at 2 or 4 (r3_valu_peak.json)
                        roof.update({'peak_for_instruction_mix_at_occupancy': occ['peak_T_wave_instructions_per_s'],
                                     'frac_of_mix_ceiling_at_occupancy': rate / occ['peak_T_wave_instructions_per_s']})
            except (OSError, ValueError, KeyError):
                pass
        roof.update({
            'kernel': r['kernel'], 'kernel_ms_avg': kavg_ms, 'counters_note': why_not,
            'peak_note': 'peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md: SIMD-32); '
                         'peak_at_fp64_rate = / 4 cycles, the rate behind the 78.6 TFLOP/s vector FP64 peak (the kernel\'s '
                         'instructions are binary64 / VOP3 ones for the most part); peak_for_instruction_mix weights the '
                         'two measured issue costs (tools/valu_peak.hip) by the lap loop\'s instruction classes and is the '
                         'ceiling to read frac_of_mix_ceiling against (costs at 2 / 4 / 8 waves per SIMD; ..._at_occupancy prices the same mix with '
                         'the costs measured at 3 waves per SIMD, this kernel\'s occupancy at 20 cars); valu_busy_profiled is the rocprof VALUBusy formula, '
                         'which overshoots 1 when 2-cycle instructions are in the mix',
            'hbm_nominal': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': achieved / HBM_PEAK_GBS,
                            'note': 'nominal: 20 algorithmic bytes per simulation (the finishing order) over the kernel '
                                    'time; in this histogram-only mode they are not written (3.2 KB per launch is). '
                                    'orders_mode times the run that does write them; `traffic` above is the measured '
                                    'HBM traffic per launch (block prologues, histogram atomics, register spills)'}})
        out = {
            'metric': f'race-simulations/sec ({n} drivers, {L} laps)',
            'value': r['total'] / r['elapsed'],
            'unit': 'race-simulations/s',
            'n_gpus': world,
            'process_group': (dist.get_backend() if grouped else None),
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': r['elapsed'] / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'dtype_note': 'race state and every comparison in IEEE binary64 (as the reference); random deviates carry '
                          '32 bits: uniforms w/2^32, normals from a binary32 piecewise-cubic inverse CDF (|err| <= 4.8e-7); '
                          'measured effect of that substitution on results: profiles/r5_deviate_bias.txt (GPU, 10^9 simulations '
                          'against the library\'s own 53-bit mode, which bench.py prices as `deviates53`); stated tolerance of a '
                          'win probability against the reference: 4 SE of a 2x10^7 / 10^8-simulation pair = 0.05 pp at p = 0.5 '
                          '(tests/test_gpu_scale.py)',
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: {n} drivers, {L} laps, {per_gpu} simulations per GPU per step, '
                                   f'fixed Elo grid, seed {r["seed"]}',
                       'sims_per_gpu_per_step': per_gpu, 'parallelism': f'sims sharded over {world} GPU(s)'},
            'roofline': roof,
            'valu': {'car_laps_per_s': per_gpu * n * L / (kavg_ms * 1e-3),
                     'sims_per_s_kernel_only': per_gpu / (kavg_ms * 1e-3), 'launch': r['launch']},
            'win_probability_top3': {r['drivers'][i]: float(hist[i, 0]) / r['total'] for i in np.argsort(-hist[:, 0])[:3]},
        }
    if world == 1 and not args.no_extras:
        # side measurements in the same run (not part of `value`): BASELINE configs[2] and the orders-writing mode
        w78 = run_workload('S78', 3, 1)
        wo = run_workload(args.workload, 3, 1, with_orders=True)
        out['workloads'] = {'S78': {
            'metric': f"race-simulations/sec ({w78['n']} drivers, {w78['L']} laps; Monaco parameters, BASELINE configs[2])",
            'value': w78['total'] / w78['elapsed'], 'steps': 3, 'kernel_ms_avg': w78['kernel_ms'],
            'car_laps_per_s': per_gpu * w78['n'] * w78['L'] / (w78['kernel_ms'] * 1e-3)}}
        # what the reference's deviate width costs on this device: the same workload with 53-bit uniforms
        # and binary64 normals (mcgp_config.deviates = MCGP_DEVIATES_53; bit-identical to the oracle's PHILOX53 back-end)
        w53 = run_workload(args.workload, 3, 1, deviates=53)
        out['deviates53'] = {'value': w53['total'] / w53['elapsed'], 'unit': 'race-simulations/s', 'steps': 3,
                             'kernel_ms_avg': w53['kernel_ms'], 'kernel': w53['kernel'], 'launch': w53['launch'],
                             'slowdown_vs_32bit_deviates': w53['kernel_ms'] / kavg_ms,
                             'note': 'every draw keeps the 32-bit word as its leading bits and takes 21 more from a companion '
                                     'Philox block -- computed only where the word alone does not decide (normals always; a '
                                     'Bernoulli draw when the word equals the leading word of its threshold, one in 2^32); '
                                     'binary64 normals from a degree-7 table whose rows sit in LDS; bit-identical to the '
                                     "oracle's PHILOX53 back-end; effect on results: profiles/r5_deviate_bias.txt (GPU, 10^9 simulations)"}
        # its own roofline block: the same binding bound, counters of the reference-width kernel
        pcw, why_w = profiled_counters(args.workload + '_wide', per_gpu, bool(os.environ.get('MCGP_LIB')))
        roof53 = {'bound': 'valu-issue', 'achieved': None, 'peak': VALU_PEAK_TINST, 'unit': 'T wave-instructions/s', 'frac': None,
                  'traffic': None, 'kernel': w53['kernel'], 'kernel_ms_avg': w53['kernel_ms'], 'counters_note': why_w}
        if pcw:
            rate53 = pcw['SQ_INSTS_VALU'] / (w53['kernel_ms'] * 1e-3) / 1e12
            roof53.update({'achieved': rate53, 'frac': rate53 / VALU_PEAK_TINST, 'traffic': pcw.get('hbm_bytes_per_launch'),
                           'frac_at_fp64_rate': rate53 / VALU_PEAK_TINST_4CYCLE, 'valu_insts_per_launch': pcw['SQ_INSTS_VALU'],
                           'active_lane_ratio': pcw.get('active_lane_ratio'), 'waves_per_simd': pcw.get('waves_per_simd'),
                           'issue_frac': pcw.get('issue_frac'), 'wait_frac': pcw.get('wait_frac'),
                           'lds_conflict_share': (pcw['SQ_LDS_BANK_CONFLICT'] / pcw['SQ_LDS_IDX_ACTIVE']
                                                  if pcw.get('SQ_LDS_IDX_ACTIVE') else None),
                           'scratch_bytes_per_lane': int(pcw.get('kernel', {}).get('Scratch_Size', 0) or 0),
                           'source_hash': pcw.get('source_hash')})
            if pcw.get('active_lane_ratio'):
                roof53['frac_useful_lanes'] = roof53['frac'] * pcw['active_lane_ratio']
        out['deviates53']['roofline'] = roof53
        out['value_at_reference_width'] = out['deviates53']['value']
        bytes_written = per_gpu * wo['n']
        out['orders_mode'] = {'value': wo['total'] / wo['elapsed'], 'unit': 'race-simulations/s', 'steps': 3,
                              'kernel_ms_avg': wo['kernel_ms'], 'bytes_per_launch': bytes_written,
                              'write_gb_per_s': bytes_written / (wo['kernel_ms'] * 1e-3) / 1e9,
                              'frac_of_hbm_peak': bytes_written / (wo['kernel_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if not args.no_extras:
        # BASELINE configs[4]: the 2024 calendar (24 races) x 10^7 simulations each through the code `cli backtest` runs
        # (reference src/validation.py:161-209): races round-robin over ranks, no collective in the data path, one
        # all_gather_object of the per-race rows at the end.  Wall time between barrier + synchronize pairs, MAX over ranks.
        from monte_carlo_gp_amd.cli import backtest, backtest_jobs
        n_races = len(backtest_jobs([2024], 42))
        sync()
        t0 = time.perf_counter()
        res = backtest([2024], seed=42, n_simulations=SWEEP_SIMS_PER_RACE, device=local_rank, rank=rank, world=world)
        sync()
        dt = time.perf_counter() - t0
        if grouped:
            t = torch.tensor([dt], dtype=torch.float64, device='cpu' if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if rank == 0:
            assert res['n_races'] == n_races
            out.setdefault('workloads', {})['sweep24'] = {
                'metric': f'backtest sweep: {n_races} races of the 2024 calendar x {SWEEP_SIMS_PER_RACE} simulations each '
                          '(BASELINE configs[4]; cli.backtest)',
                'wall_seconds': dt, 'value': n_races * SWEEP_SIMS_PER_RACE / dt, 'unit': 'race-simulations/s',
                'n_races': n_races, 'simulations_per_race': SWEEP_SIMS_PER_RACE, 'kernel_launches': n_races,
                'launches_per_rank': -(-n_races // world), 'n_gpus': world,
                'win_brier': res['win_brier'], 'pole_brier': res['pole_brier'], 'podium_accuracy': res['podium_accuracy'],
                'reference_comparable': res['reference_comparable'],
                'note': 'one launch per race (above 10^6 simulations a race fills the device by itself; at the reference\'s 10^4 the '
                        'races of a rank share one launch); host inputs, Elo evolution and scoring included in the wall time; '
                        'outcomes are a hand-entered results file: scores are not comparable with a live reference run'}
    gpu_seconds = time.perf_counter() - t_gpu0           # every GPU leg of this run: timed steps, warm-up, side lines
    # which device each rank ran on: N ranks on N DISTINCT GPUs is a fact of the run, printed, not an assumption
    props = torch.cuda.get_device_properties(dev)
    me = {'rank': rank, 'local_device': local_rank, 'name': props.name,
          'uuid': str(getattr(props, 'uuid', '')), 'pci_bus_id': getattr(props, 'pci_bus_id', None),
          'pci_device_id': getattr(props, 'pci_device_id', None), 'pci_domain_id': getattr(props, 'pci_domain_id', None)}
    devices = [me]
    if grouped:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    if rank == 0:
        out['devices'] = devices
        out['distinct_devices'] = len({(d['uuid'], d['pci_bus_id'], d['pci_domain_id']) for d in devices})
        out['library'] = {'build_hash': N.build_hash(), 'path': os.environ.get('MCGP_LIB') or N.LIB_PATH}
        out['gpu_seconds'] = gpu_seconds
        if world == 1 and not args.no_cpu_baseline:
            t_cpu0 = time.perf_counter()
            out['cpu_baseline'] = cpu_baseline(args.workload)
            out['cpu_baseline_seconds'] = time.perf_counter() - t_cpu0
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
