"""HIP path vs the CPU oracle (Philox back-end): bit-exact integer results.

Link 2 of the parity chain: same (seed, simulation ids) -> identical
finishing orders and histograms.  Every call goes through the C ABI (ctypes).
"""
import numpy as np
import pytest

import oracle_py as O
from helpers import product_run, product_sim

pytestmark = pytest.mark.gpu

CASES = ['S60', 'S78', 'S50', 'EVT', 'HET', 'DMP', 'WET', 'N10']


@pytest.mark.parametrize('name', CASES)
def test_orders_and_histogram_match_oracle(require_gpu, name):
    case = O.load_case(name)
    P = O.Problem(case)
    n_sims, seed = 3000, 42
    ref = P.run(n_sims, rng=O.RNG_PHILOX, seed=seed, want_orders=True)
    hist, probs, orders = product_run(case, n_sims, seed, orders=True)
    bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
    assert bad.size == 0, f'{name}: {bad.size} of {n_sims} finishing orders differ, first sim {bad[:5]}'
    assert np.array_equal(hist, ref['hist'])
    # reference result shape: only non-zero cells, 1-based positions, probabilities
    for i, d in enumerate(P.drivers):
        assert set(probs[d].keys()) == {int(p) + 1 for p in np.nonzero(ref['hist'][i])[0]}
    assert abs(sum(sum(v.values()) for v in probs.values()) - P.n) < 1e-9


def test_partition_and_offset_invariance(require_gpu):
    """Any split of [0, N) over calls gives the same orders / histogram; 64-bit ids and seeds."""
    case = O.load_case('S50')
    seed = 0xDEADBEEFCAFEF00D
    base = (1 << 40) + 12345
    hist, _, orders = product_run(case, 1500, seed, sim_offset=base, orders=True)
    h1, _, o1 = product_run(case, 333, seed, sim_offset=base, orders=True)
    h2, _, o2 = product_run(case, 1167, seed, sim_offset=base + 333, orders=True)
    assert np.array_equal(np.vstack([o1, o2]), orders)
    assert np.array_equal(h1 + h2, hist)
    ref = O.Problem(case).run(1500, rng=O.RNG_PHILOX, seed=seed, sim_offset=base, want_orders=True)
    assert np.array_equal(orders, ref['orders'])


def test_generic_and_register_kernels_agree(require_gpu, monkeypatch):
    """n = 20 has a register-resident instantiation; MCGP_FORCE_GENERIC routes the same problem to the LDS kernel."""
    case = O.load_case('EVT')
    a = product_run(case, 2000, 9, orders=True)
    monkeypatch.setenv('MCGP_FORCE_GENERIC', '1')
    b = product_run(case, 2000, 9, orders=True)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0])


def _field(n):
    """An n-car field with S60's parameters, 25 laps and an all-zero grid column (Q18 uniform fallback)."""
    rng = np.random.default_rng(n)
    drivers = [f'D{i:02d}' for i in range(n)]
    base = O.load_case('S60')
    case = dict(base)
    case['config'] = dict(base['config'], total_laps=25,
                          driver_teams={d: list(base['config']['dnf_rates'])[i % 10] for i, d in enumerate(drivers)})
    g = rng.random((n, n))
    g[:, n // 2] = 0.0                      # an all-zero column: uniform-over-remaining fallback
    case['grid_probs'] = {d: [float(x) for x in g[i]] for i, d in enumerate(drivers)}
    case['base_pace'] = {d: 90.0 + 0.2 * i for i, d in enumerate(drivers)}
    case['tire_deg'] = {d: 0.05 for d in drivers}
    case['driver_variance'] = {d: 0.2 for d in drivers}
    case['driver_dnf_rates'] = {d: 0.01 for d in drivers}
    return case


@pytest.mark.parametrize('n', [1, 2, 3, 7, 9, 17, 18, 19, 22, 24, 25, 26, 32])
def test_other_field_sizes(require_gpu, n):
    """Field sizes beyond the golden cases (generic LDS kernel and further register instantiations),
    up to the ABI maximum, with an all-zero grid column (Q18 uniform fallback)."""
    case = _field(n)
    hist, probs, orders = product_run(case, 1000, 5, orders=True)
    ref = O.Problem(case).run(1000, rng=O.RNG_PHILOX, seed=5, want_orders=True)
    assert np.array_equal(orders, ref['orders'])
    assert np.array_equal(hist, ref['hist'])


def test_simulate_race_matches_oracle_and_run(require_gpu):
    case = O.load_case('HET')
    sim = product_sim(case)
    P = O.Problem(case)
    rng = np.random.default_rng(0)
    for sim_id in (0, 5, 1 << 33):
        perm = rng.permutation(P.n)
        grid = [P.drivers[i] for i in perm]
        # the product API takes the grid as a driver list; dict inputs are keyed by driver
        res = sim.simulate_race(grid, case['base_pace'], case['tire_deg'], case['driver_variance'],
                                case['driver_dnf_rates'], case['track_condition'], seed=77, sim_id=sim_id)
        assert [p for _, p in res] == list(range(1, P.n + 1))
        # oracle problem with drivers re-indexed in grid order (driver index == grid slot)
        sub = dict(case, grid_probs={d: case['grid_probs'][d] for d in grid})
        order = O.Problem(sub).simulate_race(np.arange(P.n), rng=O.RNG_PHILOX, seed=77, sim_id=sim_id)
        assert [d for d, _ in res] == [grid[int(i)] for i in order]


def test_tail_batches_and_tiny_runs(require_gpu):
    case = O.load_case('N10')
    P = O.Problem(case)
    for n_sims in (1, 63, 64, 65, 257, 1025):
        hist, _, orders = product_run(case, n_sims, 3, orders=True)
        ref = P.run(n_sims, rng=O.RNG_PHILOX, seed=3, want_orders=True)
        assert np.array_equal(orders, ref['orders']), n_sims
        assert np.array_equal(hist, ref['hist']), n_sims


def test_histogram_accumulates_and_orders_optional(require_gpu):
    import ctypes as C
    from monte_carlo_gp_amd import _native as N
    from monte_carlo_gp_amd.simulation import _Problem, _dptr, RaceSimulator
    from monte_carlo_gp_amd import RaceConfig
    case = O.load_case('WET')
    drivers = list(case['grid_probs'])
    p = _Problem(RaceConfig(**case['config']), drivers, case['base_pace'], case['tire_deg'],
                 case['driver_variance'], case['driver_dnf_rates'], case['track_condition'], O.load_cases()['set_pop'])
    g = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
    h = np.zeros((p.n, p.n), np.uint64)
    for off in (0, 500):
        N.check(N.lib().mcgp_run(C.byref(p.cfg), C.byref(p.drv), _dptr(g), p.n, 500, off, 21, 0,
                                 h.ctypes.data_as(C.POINTER(C.c_uint64)), None))
    ref = O.Problem(case).run(1000, rng=O.RNG_PHILOX, seed=21)
    assert np.array_equal(h.astype(np.int64), ref['hist'])


def test_orders_across_the_staging_chunk_boundary(require_gpu):
    """mcgp_run stages per-simulation orders in chunks of 2^22 simulations: check both sides of the seam."""
    case = O.load_case('N10')
    n_sims = (1 << 22) + 1500
    hist, _, orders = product_run(case, n_sims, 8, orders=True)
    P = O.Problem(case)
    for off, cnt in ((0, 1000), ((1 << 22) - 700, 1400), (n_sims - 500, 500)):
        ref = P.run(cnt, rng=O.RNG_PHILOX, seed=8, sim_offset=off, want_orders=True)
        assert np.array_equal(orders[off:off + cnt], ref['orders']), off
    # the histogram is the column-wise count of the orders
    h = np.zeros_like(hist)
    for p in range(P.n):
        h[:, p] = np.bincount(orders[:, p], minlength=P.n)
    assert np.array_equal(h, hist)


def test_concurrent_calls_from_two_threads(require_gpu):
    """The C ABI is blocking and re-entrant (one mutex-guarded context per device)."""
    from concurrent.futures import ThreadPoolExecutor
    cases = [O.load_case('S50'), O.load_case('WET')]

    def work(i):
        return product_run(cases[i % 2], 20000, 100 + i)[0]
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(work, range(8)))
    for i, h in enumerate(got):
        assert np.array_equal(h, O.Problem(cases[i % 2]).run(20000, rng=O.RNG_PHILOX, seed=100 + i)['hist']), i


def test_run_simulations_alias(require_gpu):
    """North-star call surface run_simulations(grid, n_sims, seed): a thin wrapper over run_monte_carlo with the
    per-race inputs given to set_race_inputs(); without them the reference's .get() defaults apply
    (base pace 90.0, degradation 0.05, variance 0.15, team DNF rates; reference src/simulation.py:190-204)."""
    case = O.load_case('S60')
    sim = product_sim(case)
    sim.set_race_inputs(case['base_pace'], case['tire_deg'], case['driver_variance'], case['driver_dnf_rates'],
                        case['track_condition'])
    got = sim.run_simulations(case['grid_probs'], 4000, 42)
    ref = O.Problem(case).run(4000, rng=O.RNG_PHILOX, seed=42)['hist']
    assert np.array_equal(sim.last_histogram, ref)
    direct = product_sim(case).run_monte_carlo(4000, case['grid_probs'], case['base_pace'], case['tire_deg'],
                                               case['driver_variance'], case['driver_dnf_rates'], seed=42,
                                               track_condition=case['track_condition'])
    assert got == direct
    # no inputs set: every dict falls back to the reference defaults, which is the oracle on empty dicts
    bare = product_sim(case)
    bare.run_simulations(case['grid_probs'], 3000, 7)
    dflt = dict(case, base_pace={}, tire_deg={}, driver_variance={}, driver_dnf_rates=None)
    assert np.array_equal(bare.last_histogram, O.Problem(dflt).run(3000, rng=O.RNG_PHILOX, seed=7)['hist'])


def test_long_runs_split_into_launches(require_gpu, monkeypatch):
    """A run longer than the per-launch cap (2^32 - 512 simulations; the block histogram counts in uint32) is
    split into several launches; with the cap lowered to 1000 the seams are exercised: same orders, same histogram."""
    case = O.load_case('N10')
    ref = O.Problem(case).run(3500, rng=O.RNG_PHILOX, seed=11, sim_offset=77, want_orders=True)
    monkeypatch.setenv('MCGP_MAX_SIMS_PER_LAUNCH', '1000')
    hist, _, orders = product_run(case, 3500, 11, sim_offset=77, orders=True)
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


def _device_problem(name, pit_loss_shift=0.0):
    """(case, dense problem, grid matrix); pit_loss_shift makes a problem no earlier test can have left in the
    library's parameter-block cache."""
    import copy
    from monte_carlo_gp_amd.simulation import _Problem, RaceSimulator
    from monte_carlo_gp_amd import RaceConfig
    case = copy.deepcopy(O.load_case(name))
    case['config']['pit_loss'] += pit_loss_shift
    drivers = list(case['grid_probs'])
    p = _Problem(RaceConfig(**case['config']), drivers, case['base_pace'], case['tire_deg'],
                 case['driver_variance'], case['driver_dnf_rates'], case['track_condition'], O.load_cases()['set_pop'])
    return case, p, RaceSimulator._grid_matrix(case['grid_probs'], drivers)


def _run_on_stream(p, g, n_sims, seed, stream, d_hist, offset=0):
    import ctypes as C
    from monte_carlo_gp_amd import _native as N
    from monte_carlo_gp_amd.simulation import _dptr
    N.check(N.lib().mcgp_run_device(C.byref(p.cfg), C.byref(p.drv), _dptr(g), p.n, n_sims, offset, seed, 0,
                                    C.c_void_p(stream.cuda_stream), C.c_void_p(d_hist.data_ptr()), None))


def test_cached_parameter_block_across_streams(require_gpu):
    """mcgp_run_device on torch stream A, then the same problem on stream B and on torch's current stream: the
    cached parameter block is re-used behind an event wait on its upload.  The library was loaded
    BEFORE torch in this process (require_gpu); both use one HIP runtime (_native._bind_hip_runtime)."""
    import torch
    from monte_carlo_gp_amd import _native as N
    assert len(N.hip_runtimes_mapped()) == 1, N.hip_runtimes_mapped()
    dev = torch.device('cuda', 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)]
    for name, seed in (('EVT', 5), ('DMP', 6), ('HET', 7), ('S50', 8), ('N10', 9)):       # more problems than cache slots
        case, p, g = _device_problem(name)
        bufs = [torch.zeros(p.n * p.n, dtype=torch.int64, device=dev) for _ in streams]
        torch.cuda.synchronize(dev)
        for st, d in zip(streams, bufs):
            _run_on_stream(p, g, 5000, seed, st, d)
        torch.cuda.synchronize(dev)
        ref = O.Problem(case).run(5000, rng=O.RNG_PHILOX, seed=seed)['hist']
        for d in bufs:
            assert np.array_equal(d.cpu().numpy().reshape(p.n, p.n), ref), name


def test_parameter_block_eviction_waits_for_every_stream(require_gpu):
    """VERDICT r2 item 6.  Stream A runs a long launch (1e7 simulations) on problem P0; stream B then uses the SAME
    cached block for a short launch, and goes on through four more problems, which evicts P0's block while A's
    kernel is still reading it.  With one completion event per block (re-recorded by B) the eviction would wait for
    B only and overwrite the block under A; with one event per (block, stream) A's result is intact.  Per-stream
    timing: A's and B's kernel times are both readable afterwards and differ by the 1000x in work."""
    import ctypes as C
    import torch
    from monte_carlo_gp_amd import _native as N
    dev = torch.device('cuda', 0)
    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    case0, p0, g0 = _device_problem('S60', 0.125)            # never seen before: takes a fresh cache slot
    big, small = 10_000_000, 10_000
    dA = torch.zeros(p0.n * p0.n, dtype=torch.int64, device=dev)
    dB = torch.zeros(p0.n * p0.n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    _run_on_stream(p0, g0, big, 42, A, dA)                   # ~0.1 s of kernel, asynchronous
    _run_on_stream(p0, g0, small, 42, B, dB)                 # same block, other stream
    others = []
    for name, seed in (('EVT', 5), ('DMP', 6), ('HET', 7), ('S50', 8), ('N10', 9)):       # 5 more new problems > 4 slots
        case, p, g = _device_problem(name, 0.25)
        d = torch.zeros(p.n * p.n, dtype=torch.int64, device=dev)
        _run_on_stream(p, g, 3000, seed, B, d)
        others.append((name, case, p, d, seed))
    msA, msB = C.c_float(), C.c_float()
    N.check(N.lib().mcgp_stream_kernel_ms(0, C.c_void_p(A.cuda_stream), C.byref(msA)))
    N.check(N.lib().mcgp_stream_kernel_ms(0, C.c_void_p(B.cuda_stream), C.byref(msB)))
    torch.cuda.synchronize(dev)
    assert msA.value > 20 * msB.value > 0, (msA.value, msB.value)
    hA = dA.cpu().numpy().reshape(p0.n, p0.n)
    assert (hA.sum(axis=0) == big).all() and (hA.sum(axis=1) == big).all()
    # A's run equals the same run made alone (a block overwritten under it would change the parameters mid-flight)
    alone = product_run(case0, big, 42)[0]
    assert np.array_equal(hA, alone)
    assert np.array_equal(dB.cpu().numpy().reshape(p0.n, p0.n), O.Problem(case0).run(small, rng=O.RNG_PHILOX, seed=42)['hist'])
    for name, case, p, d, seed in others:
        assert np.array_equal(d.cpu().numpy().reshape(p.n, p.n), O.Problem(case).run(3000, rng=O.RNG_PHILOX, seed=seed)['hist']), name


@pytest.mark.parametrize('generic', [False, True])
@pytest.mark.parametrize('sim', [33884187, 666225793])
def test_grid_draw_of_exactly_zero(require_gpu, monkeypatch, sim, generic):
    """S60, seed 42: the two simulations of the first 10^9 whose grid sampling draws the word 0 (u = 0.0): the first
    REMAINING driver with mass is chosen, not a placed one whose cdf entry is 0 (tests/test_kernel_host_build.py).
    Both kernels."""
    if generic:
        monkeypatch.setenv('MCGP_FORCE_GENERIC', '1')
    case = O.load_case('S60')
    ref = O.Problem(case).run(128, rng=O.RNG_PHILOX, seed=42, sim_offset=sim - 64, want_orders=True)
    hist, _, orders = product_run(case, 128, 42, sim_offset=sim - 64, orders=True)
    assert sorted(orders[64].tolist()) == list(range(20))
    assert np.array_equal(orders, ref['orders'])
    assert np.array_equal(hist, ref['hist'])


def test_more_streams_in_flight_than_work_counters(require_gpu):
    """Every stream's launches claim their 64-simulation chunks from that stream's work counter (zeroed on the stream
    before each launch); the library keeps 8 such entries per device and recycles the least recently used one behind
    its last launch.  Eleven streams, all launched before anything is awaited, three rounds: every result must be
    the run made alone -- a counter shared by two live launches would drop or repeat chunks (column sums != n)."""
    import torch
    dev = torch.device('cuda', 0)
    streams = [torch.cuda.Stream(dev) for _ in range(11)]
    case, p, g = _device_problem('S60')
    n_sims = 300_000
    alone = {seed: product_run(case, n_sims, seed)[0] for seed in (1, 2, 3)}
    for seed in (1, 2, 3):
        bufs = [torch.zeros(p.n * p.n, dtype=torch.int64, device=dev) for _ in streams]
        torch.cuda.synchronize(dev)
        for st, d in zip(streams, bufs):
            _run_on_stream(p, g, n_sims, seed, st, d)
        torch.cuda.synchronize(dev)
        for d in bufs:
            h = d.cpu().numpy().reshape(p.n, p.n)
            assert (h.sum(axis=0) == n_sims).all()
            assert np.array_equal(h, alone[seed])
    slice_ref = O.Problem(case).run(4096, rng=O.RNG_PHILOX, seed=1)['hist']
    assert np.array_equal(product_run(case, 4096, 1)[0], slice_ref)


def test_orders_into_an_unaligned_device_buffer(require_gpu):
    """mcgp_run_device with d_orders one byte off dword alignment: the kernel falls back from packed dword stores
    to byte stores (n = 20 and n = 24 are the sizes that otherwise take the packed path)."""
    import ctypes as C
    from monte_carlo_gp_amd import _native as N
    from monte_carlo_gp_amd.simulation import _Problem, _dptr, RaceSimulator
    from monte_carlo_gp_amd import RaceConfig
    N.lib()
    with open('/proc/self/maps') as f:
        paths = sorted({line.split()[-1] for line in f if 'libamdhip64' in line})
    hip = C.CDLL(paths[0])

    def ok(rc):
        assert rc == 0, f'HIP error {rc}'
    ok(hip.hipSetDevice(0))
    case = O.load_case('S60')
    drivers = list(case['grid_probs'])
    p = _Problem(RaceConfig(**case['config']), drivers, case['base_pace'], case['tire_deg'], case['driver_variance'],
                 case['driver_dnf_rates'], case['track_condition'], O.load_cases()['set_pop'])
    g = RaceSimulator._grid_matrix(case['grid_probs'], drivers)
    n_sims = 1000
    ref = O.Problem(case).run(n_sims, rng=O.RNG_PHILOX, seed=4, want_orders=True)
    d_hist, d_orders = C.c_void_p(), C.c_void_p()
    ok(hip.hipMalloc(C.byref(d_hist), C.c_size_t(p.n * p.n * 8)))
    ok(hip.hipMalloc(C.byref(d_orders), C.c_size_t(n_sims * p.n + 16)))
    for shift in (0, 1, 2, 3):
        ok(hip.hipMemset(d_hist, 0, C.c_size_t(p.n * p.n * 8)))
        ok(hip.hipMemset(d_orders, 0xEE, C.c_size_t(n_sims * p.n + 16)))
        N.check(N.lib().mcgp_run_device(C.byref(p.cfg), C.byref(p.drv), _dptr(g), p.n, n_sims, 0, 4, 0, C.c_void_p(0),
                                        d_hist, C.c_void_p(d_orders.value + shift)))
        ok(hip.hipDeviceSynchronize())
        raw = np.zeros(n_sims * p.n + 16, np.uint8)
        ok(hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), d_orders, C.c_size_t(raw.size), 2))
        assert np.array_equal(raw[shift:shift + n_sims * p.n].reshape(n_sims, p.n), ref['orders']), shift
        assert (raw[:shift] == 0xEE).all() and (raw[shift + n_sims * p.n:] == 0xEE).all(), shift     # nothing written outside
    ok(hip.hipFree(d_hist))
    ok(hip.hipFree(d_orders))


def test_whole_node_from_the_drop_in_call_three_shards_on_one_device(require_gpu):
    """RaceSimulator(config, device=[...]): the plain single-process call of the reference's caller (reference
    src/predictor.py:264,283-291) split over several devices by simulation id, one host thread per device, histograms
    added on the host.  With one GPU in the box the list names it three times: three concurrent shards
    (ragged: 5000 = 1667 + 1667 + 1666) must give the single-device result and the oracle's, finishing orders included;
    device='all' must resolve to the visible devices.  More than one physical device: unmeasured on hardware."""
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N
    case = O.load_case('S60')
    n_sims, seed = 5000, 1234
    ref = O.Problem(case).run(n_sims, rng=O.RNG_PHILOX, seed=seed, sim_offset=77, want_orders=True)
    cfg = RaceConfig(**case['config'])
    args = (case['grid_probs'], case['base_pace'], case['tire_deg'], case['driver_variance'], case['driver_dnf_rates'])
    for device in (0, [0, 0, 0], 'all'):
        sim = RaceSimulator(cfg, device=device, set_pop=O.load_cases()['set_pop'])
        probs, orders = sim.run_monte_carlo(n_sims, *args, seed=seed, track_condition=case['track_condition'],
                                            sim_offset=77, return_orders=True)
        assert np.array_equal(sim.last_histogram, ref['hist']), device
        assert np.array_equal(orders, ref['orders']), device
        assert abs(sum(probs[list(probs)[0]].values()) - 1.0) < 1e-12
    assert RaceSimulator(cfg, device='all').devices == list(range(N.lib().mcgp_device_count()))
    with pytest.raises(N.McgpError):
        RaceSimulator(cfg, device=[0, 99]).run_monte_carlo(100, *args, seed=1)      # a shard's error reaches the caller


def test_a_season_of_races_in_one_launch(require_gpu):
    """mcgp_run_batch: 24 races x 10 000 simulations -- the reference's own size (reference
    src/predictor.py:284, backtest loop src/validation.py:179-185) -- in ONE launch, plus races of two other field sizes
    (one launch per size).  Every race's histogram equals the oracle's and what mcgp_run gives for it alone; the GPU time
    of the 24-race launch is recorded (a single race of this size is one ~1.8 ms launch: 24 of them back to back ~44 ms)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from monte_carlo_gp_amd import RaceConfig, run_monte_carlo_batch, _native as N
    names = ['S60', 'S78', 'S50', 'EVT', 'DMP', 'WET']
    cases = {k: O.load_case(k) for k in names + ['HET', 'N10']}
    assert all(len(cases[k]['grid_probs']) == 20 for k in names)
    plan = [(names[i % 6], 1000 + 17 * i, 5 * i) for i in range(24)] + [('HET', 5, 0), ('N10', 6, 3), ('HET', 7, 11)]
    n_sims = 10_000

    def problem(name, seed, off):
        c = cases[name]
        return dict(config=RaceConfig(**c['config']), grid_probs=c['grid_probs'], base_pace=c['base_pace'],
                    tire_deg=c['tire_deg'], driver_variance=c['driver_variance'], driver_dnf_rates=c['driver_dnf_rates'],
                    seed=seed, track_condition=c['track_condition'], sim_offset=off)
    set_pop = O.load_cases()['set_pop']
    out = run_monte_carlo_batch([problem(*p) for p in plan], n_sims, device=0, set_pop=set_pop)
    with ThreadPoolExecutor(8) as ex:
        refs = list(ex.map(lambda p: O.Problem(cases[p[0]]).run(n_sims, rng=O.RNG_PHILOX, seed=p[1], sim_offset=p[2])['hist'], plan))
    for (name, seed, off), (probs, hist), ref in zip(plan, out, refs):
        assert np.array_equal(hist, ref), (name, seed)
        assert (hist.sum(axis=0) == n_sims).all() and (hist.sum(axis=1) == n_sims).all()
    alone = product_run(cases['EVT'], n_sims, plan[3][1], sim_offset=plan[3][2])[0]
    assert np.array_equal(alone, out[3][1])
    # the 24-race launch on its own, timed by the library's events
    run_monte_carlo_batch([problem(*p) for p in plan[:24]], n_sims, device=0, set_pop=set_pop)
    ms = C.c_float()
    N.check(N.lib().mcgp_last_kernel_ms(0, C.byref(ms)))
    name = N.lib().mcgp_last_kernel_name(0).decode()
    print(f'\nbatch of 24 x {n_sims}: kernel {name} {ms.value:.3f} ms')
    assert name == 'mcgp::race_kernel_reg_batch<20>'
    assert ms.value < 4.0            # one full round of wave-chunks and a sparse second one (2.5 ms measured), not 24 launches
    # empty and degenerate batches
    assert run_monte_carlo_batch([], n_sims) == []
    assert run_monte_carlo_batch([problem('S60', 1, 0)], 0)[0][0] == {}


def test_a_batch_gives_every_problem_what_mcgp_run_would(require_gpu, monkeypatch):
    """mcgp_run_batch = mcgp_run semantics (reference src/validation.py:179-185: a sweep is a loop over
    independent predictions).  One call with 22 ordinary 20-car races, one only the generic kernel takes (X_all_attempt:
    overtake_delta = -50) and one at reference width (deviates = 53): each gets the oracle's histogram -- the ordinary
    ones from the shared launch, the other two from their own launches inside the call -- and the call's device time is
    what mcgp_last_kernel_ms reports afterwards, on a context whose first call this may be."""
    import ctypes as C
    import json
    from monte_carlo_gp_amd import RaceConfig, run_monte_carlo_batch, _native as N
    with open(O.GOLDEN_DIR + '/fuzz_cases.json') as f:
        fuzz = json.load(f)
    names = ['S60', 'S78', 'S50', 'EVT', 'DMP', 'WET']
    cases = {k: O.load_case(k) for k in names}
    cases['X'] = fuzz['X_all_attempt']
    plan = [(names[i % 6], 500 + 13 * i, 7 * i, 32) for i in range(22)]
    plan.insert(5, ('X', 77, 3, 32))
    plan.insert(11, ('S60', 78, 9, 53))
    n_sims = 6000

    def problem(name, seed, off, dev):
        c = cases[name]
        return dict(config=RaceConfig(**c['config']), grid_probs=c['grid_probs'], base_pace=c['base_pace'],
                    tire_deg=c['tire_deg'], driver_variance=c['driver_variance'], driver_dnf_rates=c.get('driver_dnf_rates'),
                    seed=seed, track_condition=c['track_condition'], sim_offset=off, deviates=dev)
    set_pop = O.load_cases()['set_pop']
    out = run_monte_carlo_batch([problem(*p) for p in plan], n_sims, device=0, set_pop=set_pop)
    ms = C.c_float()
    N.check(N.lib().mcgp_last_kernel_ms(0, C.byref(ms)))
    assert 0.5 < ms.value < 200.0, ms.value
    for (name, seed, off, dev), (probs, hist) in zip(plan, out):
        ref = O.Problem(cases[name]).run(n_sims, rng=O.RNG_PHILOX53 if dev == 53 else O.RNG_PHILOX, seed=seed, sim_offset=off)['hist']
        assert np.array_equal(hist, ref), (name, seed, dev)
    # MCGP_FORCE_GENERIC=1 is honoured on this path too: every problem through the generic kernel, same results
    monkeypatch.setenv('MCGP_FORCE_GENERIC', '1')
    again = run_monte_carlo_batch([problem(*p) for p in plan[:4]], 1500, device=0, set_pop=set_pop)
    assert N.lib().mcgp_last_kernel_name(0).decode() == 'mcgp::race_kernel'
    monkeypatch.delenv('MCGP_FORCE_GENERIC')
    for (name, seed, off, dev), (probs, hist) in zip(plan[:4], again):
        assert np.array_equal(hist, O.Problem(cases[name]).run(1500, rng=O.RNG_PHILOX, seed=seed, sim_offset=off)['hist'])


def test_a_device_with_less_lds_gets_small_blocks(require_gpu):
    """The 20-car block fills the CU's LDS to the last half kilobyte (163 264 of 163 840 B).  A device or runtime that
    offers less per block gets the same kernel in blocks of 4 waves: same results, the shape
    visible in mcgp_last_launch_info / mcgp_last_kernel_name.  Run in child processes (the limit is read when the
    device context is created): MCGP_LDS_PER_BLOCK = what the device reports minus 4 KB, and 80 KB; at 32 KB not even the
    small block fits and the call fails with an error that says so."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C, numpy as np\n"
        "sys.path.insert(0, 'tests')\n"
        "import oracle_py as O\n"
        "from helpers import product_run\n"
        "from monte_carlo_gp_amd import _native as N\n"
        "for name in ('S60', 'HET', 'N10'):\n"
        "    case = O.load_case(name)\n"
        "    hist, probs, orders = product_run(case, 3000, 42, sim_offset=5, orders=True)\n"
        "    ref = O.Problem(case).run(3000, rng=O.RNG_PHILOX, seed=42, sim_offset=5, want_orders=True)\n"
        "    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist']), name\n"
        "    g, b, l = C.c_uint32(), C.c_uint32(), C.c_uint32()\n"
        "    N.lib().mcgp_last_launch_info(0, C.byref(g), C.byref(b), C.byref(l))\n"
        "    print(name, N.lib().mcgp_last_kernel_name(0).decode(), b.value, l.value)\n")
    import os
    for limit in (163840 - 4096, 81920):
        env = dict(os.environ, MCGP_LDS_PER_BLOCK=str(limit))
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd=O.ROOT, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = dict((ln.split()[0], ln.split()[1:]) for ln in r.stdout.strip().splitlines())
        # default blocks: 20 cars 163 264 B (12 waves), 21 cars 157 168 B (11 waves), 10 cars 69 072 B (8 waves); a block that
        # does not fit the limit is replaced by blocks of 4 waves, which always do
        for name, n, default_bytes in (('S60', 20, 163264), ('HET', 21, 157168), ('N10', 10, 69072)):
            small = default_bytes > limit
            got = ' '.join(lines[name][:-2])
            assert got == (f'mcgp::race_kernel_reg<{n}, 4>' if small else f'mcgp::race_kernel_reg<{n}>'), (limit, lines)
            assert (lines[name][-2] == '256') == small and int(lines[name][-1]) <= limit, (limit, lines)
            if not small:
                assert int(lines[name][-1]) == default_bytes, lines
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, MCGP_LDS_PER_BLOCK='32768'),
                       cwd=O.ROOT, timeout=600)
    assert r.returncode != 0 and 'bytes of LDS' in r.stderr, r.stderr[-2000:]


def test_more_than_eight_attempts_in_a_pass(require_gpu):
    """overtake_delta = 0: every pair with a pace advantage attempts, so most passes have a lane with more than the
    eight attempts the W plane holds and the wave takes the general path, eight at a time (reference :516-524)."""
    import copy
    from monte_carlo_gp_amd import _native as N
    case = copy.deepcopy(O.load_case('S60'))
    case['config']['overtake_delta'] = 0.0
    ref = O.Problem(case).run(3000, rng=O.RNG_PHILOX, seed=5, want_orders=True)
    hist, _, orders = product_run(case, 3000, 5, orders=True)
    assert N.lib().mcgp_last_kernel_name(0).decode() == 'mcgp::race_kernel_reg<20>'
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


@pytest.mark.parametrize('name', ['S60', 'S78', 'S50', 'EVT', 'HET', 'DMP', 'WET', 'N10'])
def test_reference_width_deviates_match_the_oracle(require_gpu, name):
    """deviates = 53 on the device -- 53-bit uniforms and binary64 normals, the reference's width
    (reference src/simulation.py:137,194,302,330,524) -- is bit-identical to the oracle's PHILOX53 back-end on the 8
    golden cases, finishing orders and histograms, ragged offsets included; and it is a different kernel from the
    default one."""
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N
    case = O.load_case(name)
    n_sims, seed, off = 3000, 42, 123_456_789_012
    ref = O.Problem(case).run(n_sims, rng=O.RNG_PHILOX53, seed=seed, sim_offset=off, want_orders=True)
    sim = RaceSimulator(RaceConfig(**case['config']), set_pop=O.load_cases()['set_pop'], deviates=53)
    probs, orders = sim.run_monte_carlo(n_sims, case['grid_probs'], case['base_pace'], case['tire_deg'],
                                        case['driver_variance'], case['driver_dnf_rates'], seed=seed,
                                        track_condition=case['track_condition'], sim_offset=off, return_orders=True)
    assert N.lib().mcgp_last_kernel_name(0).decode().startswith('mcgp::race_kernel_reg_wide<')
    bad = np.nonzero((orders != ref['orders']).any(axis=1))[0]
    assert bad.size == 0, f'{name}: {bad.size} finishing orders differ, first {bad[:5]}'
    assert np.array_equal(sim.last_histogram, ref['hist'])


def test_reference_width_deviates_at_every_field_size(require_gpu):
    """deviates = 53 is built for every field size, and every one of them is run here (19-22-car sessions, SURVEY section 7;
    blocks of 12, 11 and 10 waves at 3 waves per SIMD up to 22 cars, 8 and 7 waves at 2 per SIMD beyond; blocks that hold
    all / some / few of the binary64 table rows in LDS: RegGeo::kNorm53Rows); a problem only the generic kernel takes
    is refused with MCGP_E_BAD_ARG, a width other than 32 / 53 with ValueError."""
    import copy
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator, _native as N
    case = O.load_case('S60')
    for n in range(1, 33):
        c = _field(n) if n != 7 else None
        if c is None:
            keep = list(case['grid_probs'])[:7]
            c = dict(case, **{k: {d: (v[:7] if k == 'grid_probs' else v) for d, v in case[k].items() if d in keep}
                              for k in ('grid_probs', 'base_pace', 'tire_deg', 'driver_variance', 'driver_dnf_rates')})
        ref = O.Problem(c).run(1200, rng=O.RNG_PHILOX53, seed=7, want_orders=True)
        sim = RaceSimulator(RaceConfig(**c['config']), set_pop=O.load_cases()['set_pop'], deviates=53)
        _, orders = sim.run_monte_carlo(1200, c['grid_probs'], c['base_pace'], c['tire_deg'], c['driver_variance'],
                                        c['driver_dnf_rates'], seed=7, track_condition=c['track_condition'], return_orders=True)
        assert N.lib().mcgp_last_kernel_name(0).decode() == f'mcgp::race_kernel_reg_wide<{n}>'
        assert np.array_equal(orders, ref['orders']) and np.array_equal(sim.last_histogram, ref['hist']), n
    slow = copy.deepcopy(case)
    slow['config']['overtake_delta'] = -0.5
    sim = RaceSimulator(RaceConfig(**slow['config']), deviates=53)
    with pytest.raises(N.McgpError) as e:
        sim.run_monte_carlo(100, slow['grid_probs'], slow['base_pace'], slow['tire_deg'], slow['driver_variance'],
                            slow['driver_dnf_rates'], seed=1)
    assert e.value.code == -1 and 'reg_kernel_serves' in str(e.value)
    with pytest.raises(ValueError):
        RaceSimulator(RaceConfig(**case['config']), deviates=64).run_monte_carlo(
            10, case['grid_probs'], case['base_pace'], case['tire_deg'], case['driver_variance'])


def test_lap_times_near_zero_run_on_the_generic_kernel(require_gpu):
    """reg_kernel_serves: the register kernel marks a retirement in the sign of a last-lap time and drops the reference's
    max(0.1, ahead - 0.1) (:528), both of which want lap times safely above zero (reg_time_floor >= 8 s).  A field lapping
    in about 5 s cannot promise that: it runs on the generic kernel, with the oracle's results -- times around 0.1 s, where
    the max() matters, included."""
    import copy
    from monte_carlo_gp_amd import _native as N
    case = copy.deepcopy(O.load_case('S60'))
    case['base_pace'] = {d: 5.0 + 0.1 * i for i, d in enumerate(case['base_pace'])}
    ref = O.Problem(case).run(2000, rng=O.RNG_PHILOX, seed=11, want_orders=True)
    hist, _, orders = product_run(case, 2000, 11, orders=True)
    assert N.lib().mcgp_last_kernel_name(0).decode() == 'mcgp::race_kernel'
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])
    # 20 s laps are fine for the register kernel (floor 9.5 s)
    case['base_pace'] = {d: 20.0 + 0.1 * i for i, d in enumerate(case['base_pace'])}
    ref = O.Problem(case).run(2000, rng=O.RNG_PHILOX, seed=11, want_orders=True)
    hist, _, orders = product_run(case, 2000, 11, orders=True)
    assert N.lib().mcgp_last_kernel_name(0).decode() == 'mcgp::race_kernel_reg<20>'
    assert np.array_equal(orders, ref['orders']) and np.array_equal(hist, ref['hist'])


def test_thousand_lap_race(require_gpu):
    """MCGP_MAX_LAPS: the retirement keys (lap << 5 | driver, 15 bits) and the age field of pk at their largest, in both
    deviate widths."""
    import copy
    from monte_carlo_gp_amd import RaceConfig, RaceSimulator
    case = copy.deepcopy(O.load_case('N10'))
    case['config']['total_laps'] = 1000
    case['driver_dnf_rates'] = {d: 0.002 for d in case['base_pace']}          # most cars out somewhere in 1000 laps
    for deviates, rng in ((32, O.RNG_PHILOX), (53, O.RNG_PHILOX53)):
        ref = O.Problem(case).run(200, rng=rng, seed=3, want_orders=True)
        sim = RaceSimulator(RaceConfig(**case['config']), set_pop=O.load_cases()['set_pop'], deviates=deviates)
        _, orders = sim.run_monte_carlo(200, case['grid_probs'], case['base_pace'], case['tire_deg'], case['driver_variance'],
                                        case['driver_dnf_rates'], seed=3, track_condition=case['track_condition'],
                                        return_orders=True)
        assert np.array_equal(orders, ref['orders']) and np.array_equal(sim.last_histogram, ref['hist']), deviates
